// Shared device helpers for libobe_hip (gfx950 only).
//
// Build flags matter here: the library is compiled with -ffp-contract=off so that the
// parity-critical streams (Bayes update, moments, resample) perform exactly the
// NumPy sequence of correctly-rounded operations, while the flop-bound sweep kernel
// asks for its FMAs explicitly with fma().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/obe_hip.h"

namespace obe {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves: one per SIMD of a CU
constexpr int kMaxBlocks = 2048;   // 256 CUs x 8: grid cap for streaming kernels
constexpr int kFastDims = OBE_FAST_DIMS;   // cloud kernels are templated on the row count up to here, tiled beyond
constexpr double kDblMax = 1.7976931348623157e308;

void set_error(const std::string& msg);
bool defer_host_sync();    // obe_defer_host_sync: host results are copied asynchronously, the caller synchronises
int fail(hipError_t e, const char* what);
int bad_arg(const char* what);

#define OBE_HIP_TRY(expr)                                   \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) return ::obe::fail(e_, #expr); \
    } while (0)

#define OBE_CHECK_LAUNCH(name)                                  \
    do {                                                        \
        hipError_t e_ = hipGetLastError();                      \
        if (e_ != hipSuccess) return ::obe::fail(e_, name);     \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Device-visible address of page-locked host memory (hipHostMalloc / hipHostRegister: what torch's
// pin_memory() gives), or nullptr for pageable memory.  A kernel that delivers a few scalars to the
// host writes them there itself: the blit kernel of a tiny hipMemcpyAsync costs a launch (~5-10 us of
// latency per result on an otherwise short call).
inline void* device_view_of_host(const void* h) {
    if (!h) return nullptr;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, h) != hipSuccess) {
        (void)hipGetLastError();           // pageable memory is "invalid value" to some runtimes: not an error here
        return nullptr;
    }
    return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
}

// Results that a call's LAST kernel writes straight into the caller's page-locked memory are waited for
// by watching that memory, not by hipStreamSynchronize: the host arms one designated word with a bit pattern
// no result can have (a NaN payload / an impossible index), the kernel stores its results, fences at system
// scope and stores that word last, and the host spins until it changes.  On MI355X the round trip of a
// short kernel is 5.9 us this way against 11.1 us through the stream's completion signal
// (tools/microbench_sync.hip) — a cycle of the reference-sized workloads has two of them.  A kernel that
// never delivers (a fault) ends the spin after kHostWaitSpinUs and the stream is synchronised, which
// reports the error; later work on the stream is ordered behind the kernel as usual.
constexpr uint64_t kHostSentinel = 0x7ff8c0dec0dec0deULL;
constexpr double kHostWaitSpinUs = 400.0;        // (longer kernels: the stream's own wait; its 5 us no longer matter)
inline void arm_host_word(void* h_word) { *reinterpret_cast<volatile uint64_t*>(h_word) = kHostSentinel; }
int wait_host_word(const void* h_word, hipStream_t st);      // obe_capi.hip
// Round 4: results of more than one word are waited for WORD BY WORD — every word of the block is armed and
// the host spins until none carries the pattern any more.  An 8-byte store arrives whole, so that needs no
// ordering between the stores at all.  The earlier form (arm one word, store it last behind a system-scope
// fence) was not safe for blocks that span several 128-byte lines: about once in 3000 resamples the host saw
// the watched word while the covariance entries of ANOTHER line still held the previous resample's values
// (per-thread fences, an explicit s_waitcnt after the fence and delivery by a single wave did not close it;
// found by tools/fuzz_parity.py's sweeper recipes, pinned down with tools/diag_host_delivery.py).
inline void arm_host_words(void* h_words, int64_t n) {
    volatile uint64_t* p = reinterpret_cast<volatile uint64_t*>(h_words);
    for (int64_t i = 0; i < n; ++i) p[i] = kHostSentinel;
}
int wait_host_words(const void* h_words, int64_t n, hipStream_t st);      // obe_capi.hip
// device side: everything stored to the host before this call is visible there before what follows
// (the explicit wait restates the one that belongs behind the fence's L2 write-back: ROCm 7.2's compiler drops
// it when its scoreboard says this wave has nothing outstanding, and the flag store that follows can then
// overtake the write-back of the results — cdna_hip_programming.md, Guideline 16 pitfall 12; seen here as a
// host copy of the covariance with entries of the previous resample, about once in 3000 resamples)
__device__ __forceinline__ void host_results_before_flag() {
    __threadfence_system();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- "the last workgroup to arrive folds": arrival counter + write-through partials ----
// Zeroed device words owned by the library, one slot per (device, stream) (obe_capi.hip); nullptr: none
// (allocation failed, or OBE_CONTROL_SLOTS=0).  A full table hands the least recently used slot on — after a
// device synchronisation, so that no kernel of its previous owner is still counting in it.
unsigned* stream_control_words(hipStream_t st);

// A second stream (and three events) that belongs to a caller's stream: two independent kernel chains of one
// call run side by side (obe_resample.hip: the random-number chain next to the CDF / search / covariance
// chain) and are joined again with events before the call's results are consumed on the caller's stream.
// Created on first use per (device, stream), never destroyed.  false: none available — one stream does both.
struct SideStream {
    hipStream_t stream;
    hipEvent_t entry, mid, done;
    hipStream_t stream2;         // a third chain (round 5: the covariance + the (N, D) copy of a resample)
    hipEvent_t done2;
    hipEvent_t pre;              // round 6: the end of a random-number chain enqueued AHEAD of its resample
};
bool side_stream_of(hipStream_t st, SideStream* out);      // obe_capi.hip

// A block partial that another workgroup of the same launch will read: written through to memory
// (a relaxed agent-scope atomic store lowers to `global_store ... sc1`), so publishing needs no release
// fence — an agent-scope release writes back every dirty line of the XCD's L2, i.e. the weights the
// kernel has just written (MI355X_MICROARCH.md, inter-workgroup visibility: publish-large).
__device__ __forceinline__ void store_published(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ... and read back past this CU's L1 (sc1 load): stands in for an agent-scope acquire
__device__ __forceinline__ double load_published_f64(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Call after every thread of the workgroup has issued its store_published() calls.  Returns true in
// EVERY thread of the workgroup that arrived last: by then all workgroups' published values are in
// memory; read them with load_published_f64() (ACQUIRE = false) or, after the agent-scope acquire this
// function then performs, with plain loads.
// Two levels of arrival counters: the workgroups b with b % 8 == g share counter g (its own 128-byte line),
// the last one of each group takes a ticket on the top counter.  256 tickets on ONE word cost ~4.3 us of
// serialised device-scope atomics (measured: the update's normalisation pass took 22.7 / 26.5 / 31.1 us with
// 256 / 512 / 768 workgroups); 8 x 32 on separate lines run side by side: 21.3 us (16 x 16: 21.7 us).  Every counter wraps to zero with
// its last arrival (atomicInc), ready for the next launch on the stream.  `flag` = one int of LDS.
constexpr int kArriveGroups = 8;
constexpr int kArriveStride = 32;                       // words between two counters: one 128-byte line each
constexpr int kControlSlotWords = kArriveStride * (kArriveGroups + 1);
// The resample decision that the enqueue form of the fused update leaves on the device (1: this update is
// followed by a resample) for the kernels of a sweep enqueued behind it (obe_sweep.hip: OBE_SWEEP_SPECULATIVE),
// which then do nothing.  It lives in the CALLER's memory — the last 8-byte word of the workspace both calls
// are given (include/obe_hip.h: OBE_WS_ABORT_WORD) — so the pair is tied by the object that owns the workspace,
// not by the stream: another object's (or thread's) update on the same stream cannot change it.  Both forms
// refuse a workspace without 16 spare bytes behind what their kernels use.
inline unsigned* ws_abort_word(void* d_ws, int64_t ws_bytes) {
    return reinterpret_cast<unsigned*>(static_cast<char*>(d_ws) + (ws_bytes & ~(int64_t)7) - 8);
}
template <bool ACQUIRE = true>
__device__ __forceinline__ bool arrive_last(unsigned* counter, int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & (kArriveGroups - 1);
        const unsigned in_group = (gridDim.x - g + kArriveGroups - 1) / kArriveGroups;
        const unsigned groups = gridDim.x < (unsigned)kArriveGroups ? gridDim.x : (unsigned)kArriveGroups;
        int last = atomicInc(counter + g * kArriveStride, in_group - 1) == in_group - 1;          // device scope
        if (last) last = atomicInc(counter + kArriveGroups * kArriveStride, groups - 1) == groups - 1;
        if (ACQUIRE && last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");    // drop this CU's stale lines
        *flag = last;
    }
    __syncthreads();
    return *flag != 0;
}

inline int stream_blocks(int64_t n, int per_block) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return static_cast<int>(b);
}

// np.nan_to_num defaults: NaN -> 0, +inf -> DBL_MAX, -inf -> -DBL_MAX
__device__ __forceinline__ double nan_to_num(double v) {
    if (v != v) return 0.0;
    if (v > kDblMax) return kDblMax;
    if (v < -kDblMax) return -kDblMax;
    return v;
}

// ---- reductions: wave shuffle, then 4 partials through LDS; fixed order => deterministic
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
    return v;   // lane 0 holds the sum
}

// Sum of v over the block; valid in thread 0.  `red` = kBlock/kWave doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < (int)(blockDim.x / kWave); ++i) s += red[i];
    }
    return s;
}

// Sum over the block, result broadcast to every thread.
__device__ __forceinline__ double block_sum_all(double v, double* red) {
    double s = block_sum(v, red);
    __syncthreads();
    if (threadIdx.x == 0) red[0] = s;
    __syncthreads();
    return red[0];
}

// Deterministic sum of a small device array by one block (every thread gets it).
// (four elements per thread and trip, their loads issued together and added in the order i, i + nt, ... of the
// one-at-a-time loop — the same bits: the 768 partial sums of the update's first pass used to be three dependent
// round trips at the start of EVERY workgroup of its second pass)
// (nt_part: the threads that take part, the first nt_part of the workgroup — a workgroup of more threads than kBlock
// that must reproduce the sum a kBlock-thread workgroup forms, e.g. the total the unfused normalisation divides by:
// the other threads contribute +0.0 behind everybody else's partial sums, which changes nothing)
__device__ __forceinline__ double block_sum_array(const double* a, int n, double* red, int nt_part = 0) {
    double v = 0.0;
    const int nt = nt_part > 0 ? nt_part : (int)blockDim.x;
    const bool part = (int)threadIdx.x < nt;
    for (int base = 0; base < n; base += 4 * nt) {
        double x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = base + r * nt + (int)threadIdx.x;
            x[r] = a[i < n ? i : n - 1];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = base + r * nt + (int)threadIdx.x;
            const double t = v + x[r];
            v = i < n && part ? t : v;
        }
    }
    return block_sum_all(v, red);
}

// Exclusive scan, in place, of a short array (block sums of a reduce-then-scan) by ONE
// workgroup: each thread owns a contiguous segment, the segment totals are scanned through
// LDS in thread order.  Fixed association => deterministic.  Returns the grand total in
// every thread.  `lds` = blockDim.x elements of T.
template <class T>
__device__ __forceinline__ T block_exclusive_scan_inplace(T* __restrict__ data, int64_t n, T* lds) {
    const int t = threadIdx.x, nt = blockDim.x;
    const int64_t per = (n + nt - 1) / nt;
    const int64_t b = (int64_t)t * per, e = b + per < n ? b + per : n;
    T sum = T(0);
    for (int64_t i = b; i < e; ++i) sum = sum + data[i];
    lds[t] = sum;
    __syncthreads();
    if (t == 0) {
        // <= 256 serial adds.  Eight LDS reads are issued together and then added one after the other — the same
        // left-to-right sum: read / wait / add / write per element made this loop ~100 cycles per element, 11 us
        // of a 13 us kernel for the 256 block sums of a 524 288-particle CDF (rocprofv3 trace, round 5).
        T run = T(0);
        int k = 0;
        for (; k + 8 <= nt; k += 8) {
            T s[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = lds[k + j];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lds[k + j] = run;
                run = run + s[j];
            }
        }
        for (; k < nt; ++k) {
            const T s = lds[k];
            lds[k] = run;
            run = run + s;
        }
        lds[nt] = run;
    }
    __syncthreads();
    T run = lds[t];
    for (int64_t i = b; i < e; ++i) {
        const T s = data[i];
        data[i] = run;
        run = run + s;
    }
    return lds[nt];
}

// Strided view of one particle's parameters: th(i) = particles[i*ld + p].
struct ParamRef {
    const double* base;
    int64_t ld;
    __device__ __forceinline__ double operator()(int i) const { return base[(int64_t)i * ld]; }
};

// (value, index) with np.argmax ordering: NaN beats everything, then larger value,
// ties -> lower index.
struct Best {
    double v;
    int64_t i;
};
__device__ __forceinline__ bool better(const Best& a, const Best& b) {
    const bool an = a.v != a.v, bn = b.v != b.v;
    if (an != bn) return an;
    if (an && bn) return a.i < b.i;
    if (a.v != b.v) return a.v > b.v;
    return a.i < b.i;
}

}  // namespace obe
