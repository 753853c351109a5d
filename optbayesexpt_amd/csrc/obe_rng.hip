// Device-side continuation of the caller's NumPy random stream.
//
// resample() needs N uniforms (Generator.choice) and N*D standard normals
// (Generator.multivariate_normal) from the object's `rng`; for seeded runs to reproduce the
// reference they must be the *same numbers* numpy would have produced.  Generating them on
// one host core costs 25 ms at 1M particles (more than the whole utility sweep); here the
// same stream is produced on the GPU:
//
//   pcg64_raw_kernel     PCG64 = 128-bit LCG + XSL-RR output.  Thread g jumps ahead to
//                        stream position g in O(log g) 128-bit multiplies, then strides by
//                        the grid size with a precomputed (A, C) jump: coalesced 8-byte
//                        stores, no serial dependence between threads.
//   uniform              (raw >> 11) * 2^-53, numpy's next_double.
//   normals              numpy's ziggurat consumes a *variable* number of raw values per
//                        normal (1 for 98.8 % of them), so "which raw value starts the k-th
//                        normal" is a serial question.  It is answered in parallel:
//                          classify  every raw position i as if a normal started there:
//                                    value[i] and length[i] (raw values consumed);
//                          starts    position i is a real start iff the chain of jumps
//                                    i -> i + length[i] from the first position hits it;
//                                    each thread finds a nearby position that no jump can
//                                    skip (an anchor) and follows the chain from there;
//                          compact   prefix sum over the start flags = index of the normal;
//                                    scatter value[i] to out[rank[i]].
//   The ziggurat tables (ki, wi, fi) are data supplied by the host
//   (optbayesexpt_amd/data/ziggurat_tables.npz, see tools/make_ziggurat_tables.py).
#include <cstdlib>

#include "obe_common.h"

namespace obe {

struct U128 {
    uint64_t hi, lo;
};

__host__ __device__ __forceinline__ U128 mul128(U128 a, U128 b) {
    U128 r;
#if defined(__HIP_DEVICE_COMPILE__)
    r.hi = __umul64hi(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
#else
    r.hi = (uint64_t)(((unsigned __int128)a.lo * b.lo) >> 64) + a.lo * b.hi + a.hi * b.lo;
#endif
    r.lo = a.lo * b.lo;
    return r;
}

__host__ __device__ __forceinline__ U128 add128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1 : 0);
    return r;
}

// PCG_DEFAULT_MULTIPLIER_128
__host__ __device__ __forceinline__ U128 pcg_mult() { return U128{0x2360ed051fc65da4ULL, 0x4385df649fccf645ULL}; }

// (A, C) with  state_after_delta_steps = A * state + C   (LCG jump-ahead, O(log delta))
__host__ __device__ inline void lcg_jump(U128 inc, uint64_t delta, U128& A, U128& C) {
    U128 acc_mult{0, 1}, acc_plus{0, 0}, cur_mult = pcg_mult(), cur_plus = inc;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult = mul128(acc_mult, cur_mult);
            acc_plus = add128(mul128(acc_plus, cur_mult), cur_plus);
        }
        cur_plus = mul128(add128(cur_mult, U128{0, 1}), cur_plus);
        cur_mult = mul128(cur_mult, cur_mult);
        delta >>= 1;
    }
    A = acc_mult;
    C = acc_plus;
}

// XSL-RR 128 -> 64
__device__ __forceinline__ uint64_t pcg_output(U128 s) {
    const uint64_t x = s.hi ^ s.lo;
    const unsigned rot = (unsigned)(s.hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
}

struct PcgArgs {
    U128 state, inc;     // generator state before the first value of this call
    U128 strideA, strideC;   // jump by gridDim.x * blockDim.x
};

// raw[i] = output(step^(i+1)(state))
__global__ __launch_bounds__(kBlock) void pcg64_raw_kernel(PcgArgs a, int64_t n, uint64_t* __restrict__ raw) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    if (g >= n) return;
    U128 A, C;
    lcg_jump(a.inc, (uint64_t)g + 1, A, C);
    U128 s = add128(mul128(A, a.state), C);
    for (int64_t i = g; i < n; i += stride) {
        raw[i] = pcg_output(s);
        s = add128(mul128(a.strideA, s), a.strideC);
    }
}

__global__ __launch_bounds__(kBlock) void uniform_kernel(const uint64_t* __restrict__ raw, int64_t n,
                                                         double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = (double)(raw[i] >> 11) * (1.0 / 9007199254740992.0);
}

constexpr int kMaxLen = 32;          // longest ziggurat draw handled on the device
constexpr double kZigR = 3.6541528853610087963519472518;      // ziggurat tail start, 256 layers
constexpr double kZigInvR = 0.27366123732975827203338247596;   // 1/r, as published with r

struct ZigTables {
    const uint64_t* ki;
    const double* wi;
    const double* fi;
};

__device__ __forceinline__ double next_double(uint64_t r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }

// numpy/random/src/distributions: random_standard_normal (ziggurat), as if a normal started at raw
// position i.  `next()` hands out raw[i], raw[i + 1], ...; `left` = raw values from i to the end of the
// buffer.  Returns the value; *length = raw values consumed (0 = ran off the buffer / too long).
template <class Next>
__device__ __forceinline__ double zig_classify_one(Next&& next, int64_t left, const ZigTables& t, uint8_t* length) {
    int used = 0;
    double x = 0.0;
    bool done = false, bad = false;
    while (!done) {
        if (used >= left || used >= kMaxLen) { bad = true; break; }
        uint64_t r = next();
        ++used;
        const int idx = (int)(r & 0xff);
        r >>= 8;
        const int sign = (int)(r & 0x1);
        const uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
        x = (double)rabs * t.wi[idx];
        if (sign) x = -x;
        if (rabs < t.ki[idx]) break;                       // 99.3 %: inside the rectangle
        if (idx == 0) {                                    // tail of the base strip
            for (;;) {
                if (used + 1 >= left || used + 2 > kMaxLen) { bad = true; break; }
                const double xx = -kZigInvR * log1p(-next_double(next()));
                const double yy = -log1p(-next_double(next()));
                used += 2;
                if (yy + yy > xx * xx) {
                    x = ((rabs >> 8) & 0x1) ? -(kZigR + xx) : kZigR + xx;
                    done = true;
                    break;
                }
            }
            if (bad) break;
        } else {                                           // wedge
            if (used >= left) { bad = true; break; }
            const double u = next_double(next());
            ++used;
            if ((t.fi[idx - 1] - t.fi[idx]) * u + t.fi[idx] < exp(-0.5 * x * x)) done = true;
        }
    }
    *length = bad ? 0 : (uint8_t)used;
    return x;
}

__global__ __launch_bounds__(kBlock) void zig_classify_kernel(const uint64_t* __restrict__ raw, int64_t n_raw,
                                                              ZigTables t, double* __restrict__ val,
                                                              uint8_t* __restrict__ len, int64_t* __restrict__ result) {
    if (blockIdx.x == 0 && threadIdx.x == 0) result[0] = result[1] = 0;    // (what a memset node did: one launch less)
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_raw; i += (int64_t)gridDim.x * kBlock) {
        int64_t pos = i;
        uint8_t l;
        val[i] = zig_classify_one([&]() { return raw[pos++]; }, n_raw - i, t, &l);
        len[i] = l;
    }
}

// The same without a buffer of raw values (round 4): every thread carries the generator state of its
// position — jump-ahead once, then one 128-bit multiply-add per grid stride, as pcg64_raw_kernel does.
// Positions below n_uniform are the uniforms of Generator.choice, (raw >> 11) * 2^-53; the rest are
// classified, indexed from n_uniform.  One launch instead of three (raw values, uniforms, classify), and 8 B
// per position less written and twice less read.
//
// The classification is split in two: 99.3 % of the draws end inside their ziggurat rectangle after one
// table look-up — this kernel finishes those — and the others (wedge: a second raw value and an exp; tail:
// log1p pairs) are only QUEUED here, {relative position, 128-bit generator state}, in the workgroup's own
// segment of a list, for zig_slow_kernel.  With 1.2 % of the lanes taking those branches, half of all
// wavefront trips used to pay for them (49 us for 5.8 M positions; the fast part alone is bound by its 9 B
// of stores per position).  A segment that is full (128 entries for ~40 expected) makes the lane take the
// branches in place, as before.
// mult^(2^b) and the matching increments, b = 0 .. 39: a thread reaches its stream position with one 128-bit
// multiply-add per SET BIT of the distance applied to the state itself (~11 for 5.8 M positions) instead of
// composing (A, C) with the two squarings per bit of lcg_jump() (~60): the jump-ahead, not the classification,
// was most of the kernel with 22 positions per thread.
constexpr int kJumpBits = 40;
struct JumpTable {
    U128 m[kJumpBits], p[kJumpBits];
};
static JumpTable make_jump_table(U128 inc) {
    JumpTable t;
    U128 cur_mult = pcg_mult(), cur_plus = inc;
    for (int b = 0; b < kJumpBits; ++b) {
        t.m[b] = cur_mult;
        t.p[b] = cur_plus;
        cur_plus = mul128(add128(cur_mult, U128{0, 1}), cur_plus);
        cur_mult = mul128(cur_mult, cur_mult);
    }
    return t;
}
__device__ __forceinline__ U128 jump_state(U128 s, uint64_t delta, const JumpTable& jt) {
    for (int b = 0; delta != 0 && b < kJumpBits; ++b, delta >>= 1)
        if (delta & 1) s = add128(mul128(jt.m[b], s), jt.p[b]);
    return s;
}

struct SlowEntry {
    uint64_t state_hi, state_lo;
    uint32_t rel;
    uint32_t pad;
};
constexpr int kSlowPerBlock = 128;
constexpr int kRngBlocksMax = 4096;

__global__ __launch_bounds__(kBlock) void pcg_uniform_classify_kernel(PcgArgs a, JumpTable jt, U128 step_mult,
                                                                      int64_t n_uniform, int64_t n_rel, ZigTables t,
                                                                      double* __restrict__ uniforms,
                                                                      double* __restrict__ val,
                                                                      uint8_t* __restrict__ len,
                                                                      int64_t* __restrict__ result,
                                                                      SlowEntry* __restrict__ slow,
                                                                      uint32_t* __restrict__ slow_count) {
    __shared__ unsigned queued;
    __shared__ uint64_t s_ki[256];          // the two tables every draw looks up, in LDS: a wave's 64 random
    __shared__ double s_wi[256];            // 8-byte gathers cost ~10 us of the kernel's 28 from the vector L1
    s_ki[threadIdx.x] = t.ki[threadIdx.x];
    s_wi[threadIdx.x] = t.wi[threadIdx.x];
    if (threadIdx.x == 0) queued = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) result[0] = result[1] = 0;
    __syncthreads();
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t n = n_uniform + n_rel;
    if (g < n) {
        U128 s = jump_state(a.state, (uint64_t)g + 1, jt);
        for (int64_t j = g; j < n; j += stride) {
            const uint64_t raw = pcg_output(s);
            if (j < n_uniform) {
                uniforms[j] = next_double(raw);
            } else {
                const int64_t i = j - n_uniform;
                uint64_t r = raw;
                const int idx = (int)(r & 0xff);
                r >>= 8;
                const int sign = (int)(r & 0x1);
                const uint64_t rabs = (r >> 1) & 0x000fffffffffffffULL;
                double x = (double)rabs * s_wi[idx];
                if (sign) x = -x;
                uint8_t l = 1;
                if (!(rabs < s_ki[idx])) {
                    const unsigned slot = atomicAdd(&queued, 1u);             // (LDS)
                    if (slot < (unsigned)kSlowPerBlock) {
                        slow[(int64_t)blockIdx.x * kSlowPerBlock + slot] = SlowEntry{s.hi, s.lo, (uint32_t)i, 0u};
                    } else {                                                   // segment full: here and now
                        U128 c = s;
                        x = zig_classify_one([&]() {
                            const uint64_t v = pcg_output(c);
                            c = add128(mul128(step_mult, c), a.inc);
                            return v;
                        }, n_rel - i, t, &l);
                    }
                }
                val[i] = x;
                len[i] = l;
            }
            s = add128(mul128(a.strideA, s), a.strideC);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) slow_count[blockIdx.x] = queued < (unsigned)kSlowPerBlock ? queued : (unsigned)kSlowPerBlock;
}

// the queued draws: wedge and tail of the ziggurat, from the generator state the entry carries
__global__ __launch_bounds__(kSlowPerBlock) void zig_slow_kernel(U128 inc, U128 step_mult, int64_t n_rel, ZigTables t,
                                                                  const SlowEntry* __restrict__ slow,
                                                                  const uint32_t* __restrict__ slow_count,
                                                                  double* __restrict__ val, uint8_t* __restrict__ len) {
    if (threadIdx.x >= slow_count[blockIdx.x]) return;
    const SlowEntry e = slow[(int64_t)blockIdx.x * kSlowPerBlock + threadIdx.x];
    U128 c{e.state_hi, e.state_lo};
    const int64_t i = e.rel;
    uint8_t l;
    val[i] = zig_classify_one([&]() {
        const uint64_t v = pcg_output(c);
        c = add128(mul128(step_mult, c), inc);
        return v;
    }, n_rel - i, t, &l);
    len[i] = l;
}

// flag[i] = 1 iff a normal really starts at raw position i (i >= first).
// A workgroup stages the lengths of its 2048 positions plus a 32-position halo in LDS.
// Position a is an *anchor* (certainly a start) iff no earlier position reaches past it:
// max_{q<a} (q + len[q]) <= a, and only the 31 predecessors can (len <= kMaxLen).  Each
// thread finds the nearest anchor at or before its first position and then follows the
// chain i -> i + len[i] once across its 8 positions.
constexpr int kStartItems = 8;
constexpr int kStartTile = kBlock * kStartItems;
constexpr int kHalo = 2 * kMaxLen;        // room to step back over a few non-anchors

__global__ __launch_bounds__(kBlock) void zig_starts_kernel(const uint8_t* __restrict__ len, int64_t n_raw,
                                                            int64_t first, uint8_t* __restrict__ flag,
                                                            uint32_t* __restrict__ sums) {
    __shared__ __attribute__((aligned(8))) uint8_t sl[kStartTile + kHalo];
    const int64_t tile0 = (int64_t)blockIdx.x * kStartTile;
    // (staged 8 bytes per thread — the tile starts on a multiple of 2048 and len[] is padded to a multiple of 8:
    // byte-wise staging was 8 dependent trips of 1-byte loads per thread, most of this kernel's 30 us)
    {
        const int64_t g0 = tile0 + (int64_t)threadIdx.x * 8;
        uint64_t v = 0x0101010101010101ULL;
        if (g0 + 8 <= n_raw && g0 >= first) {
            v = *reinterpret_cast<const uint64_t*>(len + g0);
        } else if (g0 < n_raw) {
            v = 0;
            for (int k = 0; k < 8; ++k) {
                const int64_t g = g0 + k;
                const uint64_t b = (g >= first && g < n_raw) ? len[g] : (uint8_t)1;
                v |= b << (8 * k);
            }
        }
        *reinterpret_cast<uint64_t*>(sl + kHalo + threadIdx.x * 8) = v;
        if (threadIdx.x < kHalo) {
            const int64_t g = tile0 - kHalo + threadIdx.x;
            sl[threadIdx.x] = (g >= first && g >= 0 && g < n_raw) ? len[g] : (uint8_t)1;      // before `first`: harmless 1s
        }
    }
    __syncthreads();
    const int base = kHalo + threadIdx.x * kStartItems;              // LDS index of my first position
    const int64_t i0 = tile0 + (int64_t)threadIdx.x * kStartItems;
    // nearest anchor a <= base (LDS index); the global position `first` is always one
    // The test "no predecessor reaches past a" — sl[a - k] <= k for k = 1 .. 31 — on all 32 predecessors at
    // once: five aligned 8-byte LDS reads cover the window [a - 32, a), a funnel shift lines it up, and with
    // byte lengths <= 32 the sum of a byte and 127 - k has its top bit set exactly when the byte exceeds k
    // (no carry between bytes).  (Walking the 31 bytes one dependent LDS read at a time was 16 of the kernel's
    // 27 us.)
    auto is_anchor = [&](int pos) -> bool {
        const int lo = (pos - 32) & ~7, sh = ((pos - 32) & 7) * 8;
        const uint64_t* w8 = reinterpret_cast<const uint64_t*>(sl + lo);
        uint64_t v[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) v[m] = w8[m];
        uint64_t over = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint64_t w = sh ? (v[m] >> sh) | (v[m + 1] << (64 - sh)) : v[m];
            // byte j of word m sits at distance k = 32 - 8 m - j:  constant byte = 127 - k = 95 + 8 m + j
            const uint64_t c = 0x0706050403020100ULL + 0x0101010101010101ULL * (uint64_t)(95 + 8 * m);
            over |= w + c;
        }
        return (over & 0x8080808080808080ULL) == 0;
    };
    int a = base;
    for (;;) {
        const int64_t ga = tile0 - kHalo + a;
        if (ga <= first || a <= kMaxLen) break;
        if (is_anchor(a)) break;
        --a;
    }
    int p = a;
    {   // positions before `first` are not starts; begin the chain at `first` if it is inside my reach
        const int64_t gp = tile0 - kHalo + p;
        if (gp < first) p += (int)(first - gp);
    }
    uint32_t out[kStartItems];
#pragma unroll
    for (int k = 0; k < kStartItems; ++k) out[k] = 0u;
    while (p < base + kStartItems) {
        if (p >= base) out[p - base] = 1u;
        const int l = sl[p];
        if (l == 0) break;                 // unclassifiable (only within kMaxLen of the buffer end)
        p += l;
    }
    // one byte per position: a thread's 8 flags are one 8-byte store (lane-contiguous), and the
    // number of starts in the tile goes to sums[] (the block sums of the scan that follows)
    uint64_t packed = 0;
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < kStartItems; ++k) {
        const int64_t i = i0 + k;
        const uint32_t f = (i < n_raw && i >= first) ? out[k] : 0u;
        packed |= (uint64_t)f << (8 * k);
        mine += f;
    }
    if (i0 < n_raw) *reinterpret_cast<uint64_t*>(flag + i0) = packed;     // flag[] is padded to a multiple of 8
    __shared__ uint32_t wsum[kBlock / kWave];
    for (int o = kWave / 2; o > 0; o >>= 1) mine += __shfl_down(mine, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kBlock / kWave; ++w) t += wsum[w];
        sums[blockIdx.x] = t;
    }
}

// ---- uint32 exclusive scan of the flags (reduce-then-scan, 2048 per block) ----
constexpr int kFlagItems = 8;
constexpr int kFlagTile = kBlock * kFlagItems;

__device__ __forceinline__ uint32_t flag_tile_scan(uint32_t (&v)[kFlagItems], uint32_t* lds) {
#pragma unroll
    for (int k = 1; k < kFlagItems; ++k) v[k] += v[k - 1];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    uint32_t incl = v[kFlagItems - 1];
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += up;
    }
    __syncthreads();
    if (lane == kWave - 1) lds[wid] = incl;
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) {
        if (w < wid) wave_off += lds[w];
        total += lds[w];
    }
    const uint32_t thread_off = wave_off + incl - v[kFlagItems - 1];
#pragma unroll
    for (int k = 0; k < kFlagItems; ++k) v[k] += thread_off;      // inclusive
    return total;
}

// out[rank] = val[i] for the first n starts; result[0] = raw consumed by them (relative to `first`),
// result[1] = number of starts found.
// The tile of candidate values is staged through LDS: a thread needs the flags of 8 CONSECUTIVE
// positions for the scan, but reading val[] and writing out[] that way makes every load / store
// instruction of a wave touch 64 different cache lines; instead the tile is read with lane-contiguous
// loads, compacted inside LDS (98.8 % of the positions are starts) and written out lane-contiguously.
// (host: the device view of the caller's page-locked {consumed, found}, with `counter` the stream's arrival
// counter — then the workgroup that finishes last stores both there, `found` behind a system-scope fence:
// the host watches that word instead of waiting for a copy and a stream synchronisation)
__global__ __launch_bounds__(kBlock) void zig_compact_kernel(const uint8_t* __restrict__ flag,
                                                             const uint32_t* __restrict__ block_off,
                                                             const double* __restrict__ val,
                                                             const uint8_t* __restrict__ len, int64_t n_raw,
                                                             int64_t first, int64_t n, double* __restrict__ out,
                                                             int64_t* result, int64_t* host, unsigned* counter) {
    __shared__ uint32_t lds[kBlock / kWave];
    __shared__ double sval[kFlagTile];
    __shared__ double sout[kFlagTile];
    const int64_t tile0 = (int64_t)blockIdx.x * kFlagTile;
    // This tile's first rank = the number of starts in all tiles before it.  Round 5: every workgroup adds up the
    // tile counts in front of it itself (zig_starts_kernel left them in block_off[]; <= ~2600 values = 10 KB that
    // live in L2, exact integer arithmetic) instead of a one-workgroup exclusive scan in a launch of its own
    // between the two kernels: 8 us + a launch boundary off the longest chain of a resample.  The loads — four
    // per thread and trip, in flight together — are issued FIRST, ahead of the tile's own loads from HBM: as a
    // loop of one load / wait / add behind the tile scan they were up to ten dependent round trips per workgroup.
    uint32_t before = 0;
    for (unsigned base = 0; base < blockIdx.x; base += 4 * kBlock) {
        uint32_t x[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned j = base + r * kBlock + threadIdx.x;
            x[r] = block_off[j < blockIdx.x ? j : 0];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned j = base + r * kBlock + threadIdx.x;
            before += j < blockIdx.x ? x[r] : 0u;
        }
    }
#pragma unroll
    for (int j = 0; j < kFlagItems; ++j) {
        const int64_t i = tile0 + threadIdx.x + j * kBlock;
        sval[threadIdx.x + j * kBlock] = i < n_raw ? val[i] : 0.0;
    }
    uint32_t v[kFlagItems], f[kFlagItems];
    int owns_last_normal = 0;           // this thread found the n-th normal (at most one thread of the grid)
    const int64_t i0 = tile0 + (int64_t)threadIdx.x * kFlagItems;
    const uint64_t packed = i0 < n_raw ? *reinterpret_cast<const uint64_t*>(flag + i0) : 0ull;   // 8 byte flags
#pragma unroll
    for (int k = 0; k < kFlagItems; ++k) f[k] = v[k] = (uint32_t)((packed >> (8 * k)) & 1u);
    const uint32_t total = flag_tile_scan(v, lds);       // (contains the barriers that publish sval)
    for (int o = kWave / 2; o > 0; o >>= 1) before += __shfl_down(before, o, kWave);
    __shared__ uint32_t woff[kBlock / kWave];
    __syncthreads();                                     // (lds[] of the tile scan has been consumed)
    if ((threadIdx.x & (kWave - 1)) == 0) woff[threadIdx.x / kWave] = before;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) off += woff[w];
#pragma unroll
    for (int k = 0; k < kFlagItems; ++k) {
        if (f[k]) {
            sout[v[k] - 1] = sval[threadIdx.x * kFlagItems + k];      // v is inclusive
            const int64_t rank = (int64_t)off + v[k] - 1;
            if (rank == n - 1) {
                const int64_t i = i0 + k;
                __hip_atomic_store(result, i + (int64_t)len[i] - first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                owns_last_normal = 1;
            }
        }
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < total; j += kBlock) {
        const int64_t rank = (int64_t)off + j;
        if (rank < n) out[rank] = sout[j];
    }
    const bool is_last_block = blockIdx.x == gridDim.x - 1;
    const int64_t found = (int64_t)off + total;                  // (in the last workgroup: all starts found)
    if (is_last_block && threadIdx.x == 0)
        __hip_atomic_store(result + 1, found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (host) {
        // Two workgroups hold the result: the one with the n-th normal ({consumed}) and the last one ({found});
        // whichever of the two finishes second delivers (a counter that wraps at two arrivals — a ticket from
        // every one of the ~2600 workgroups cost 19 us of serialised atomics).  No n-th normal (found < n, the
        // caller's check will fail): the last workgroup delivers alone.
        const int owner = __syncthreads_or(owns_last_normal);
        const bool alone = is_last_block && (owner || found < n);
        if (!alone && !owner && !is_last_block) return;
        bool deliver = alone;
        if (!alone) {
            __shared__ int second;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) second = atomicInc(counter, 1u) == 1u;
            __syncthreads();
            deliver = second != 0;
        }
        if (deliver && threadIdx.x == 0) {
            host[0] = __hip_atomic_load(result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            host_results_before_flag();
            host[1] = __hip_atomic_load(result + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace obe

using namespace obe;

extern "C" {

int obe_pcg64_raw(const uint64_t* h_state4, int64_t n_raw, uint64_t* d_raw, void* stream) {
    if (!h_state4 || !d_raw || n_raw <= 0) return bad_arg("obe_pcg64_raw: bad pointer/size");
    PcgArgs a;
    a.state = U128{h_state4[0], h_state4[1]};
    a.inc = U128{h_state4[2], h_state4[3]};
    if ((a.inc.lo & 1) == 0) return bad_arg("obe_pcg64_raw: PCG64 increment must be odd");
    int blocks = static_cast<int>(std::min<int64_t>(512, (n_raw + kBlock - 1) / kBlock));
    lcg_jump(a.inc, (uint64_t)blocks * kBlock, a.strideA, a.strideC);
    pcg64_raw_kernel<<<blocks, kBlock, 0, as_stream(stream)>>>(a, n_raw, d_raw);
    OBE_CHECK_LAUNCH("pcg64_raw_kernel");
    return 0;
}

int obe_pcg64_uniform(const uint64_t* d_raw, int64_t n, double* d_out, void* stream) {
    if (!d_raw || !d_out || n <= 0) return bad_arg("obe_pcg64_uniform: bad pointer/size");
    uniform_kernel<<<stream_blocks(n, kBlock), kBlock, 0, as_stream(stream)>>>(d_raw, n, d_out);
    OBE_CHECK_LAUNCH("uniform_kernel");
    return 0;
}

int64_t obe_ziggurat_workspace_bytes(int64_t n_raw) {
    if (n_raw < 1) n_raw = 1;
    const int64_t nb = (n_raw + kFlagTile - 1) / kFlagTile;
    // candidate values, byte flags (padded to 8), lengths, block sums + the queue of the draws that leave their
    // ziggurat rectangle (a segment per workgroup of the classification)
    return n_raw * (8 + 1 + 1) + nb * 4 + 1024 + (int64_t)kRngBlocksMax * (kSlowPerBlock * (int64_t)sizeof(SlowEntry) + 4) + 64;
}

struct ZigWs {
    int64_t* result;     // [0] consumed [1] starts found
    double* val;         // candidate value per raw position
    uint8_t* flag;       // start flags, one byte per position (padded to 8)
    uint32_t* sums;      // starts per tile of 2048 positions
    uint8_t* len;        // raw values a normal starting here consumes (8-byte aligned: staged 8 at a time)
    SlowEntry* slow;     // kRngBlocksMax segments of kSlowPerBlock queued draws
    uint32_t* slow_count;
    int64_t nb;
};
static ZigWs zig_carve(void* d_ws, int64_t n_raw) {
    ZigWs w;
    w.nb = (n_raw + kFlagTile - 1) / kFlagTile;
    char* base = static_cast<char*>(d_ws);
    w.result = reinterpret_cast<int64_t*>(base);
    w.val = reinterpret_cast<double*>(base + 64);
    const int64_t n_pad = (n_raw + 7) / 8 * 8;
    w.flag = reinterpret_cast<uint8_t*>(base + 64 + n_raw * 8);
    w.sums = reinterpret_cast<uint32_t*>(w.flag + n_pad);
    w.len = reinterpret_cast<uint8_t*>(w.sums + (w.nb + 1) / 2 * 2);
    uintptr_t q = (reinterpret_cast<uintptr_t>(w.len + n_pad) + 15) & ~uintptr_t(15);
    w.slow = reinterpret_cast<SlowEntry*>(q);
    w.slow_count = reinterpret_cast<uint32_t*>(w.slow + (int64_t)kRngBlocksMax * kSlowPerBlock);
    return w;
}
static ZigTables zig_tables(const void* d_tables) {
    ZigTables t;
    t.ki = static_cast<const uint64_t*>(d_tables);
    t.wi = reinterpret_cast<const double*>(t.ki + 256);
    t.fi = t.wi + 256;
    return t;
}

// start flags, their scan and the compaction of the first n normals (the classification has been enqueued)
static int zig_finish(const ZigWs& w, int64_t n_raw, int64_t offset, int64_t n, double* d_out, int64_t* h_consumed,
                      hipStream_t st) {
    static_assert(kStartTile == kFlagTile && kStartItems == kFlagItems, "the start flags and their scan share one tiling");
    zig_starts_kernel<<<(unsigned)w.nb, kBlock, 0, st>>>(w.len, n_raw, offset, w.flag, w.sums);
    OBE_CHECK_LAUNCH("zig_starts_kernel");
    // (no scan launch: zig_compact_kernel's workgroups add up the tile counts in front of them themselves)
    // deferred + page-locked h_consumed: the kernel delivers {consumed, found} itself and the caller watches
    // both words (armed here: no count has that bit pattern)
    int64_t* hv = defer_host_sync() ? static_cast<int64_t*>(device_view_of_host(h_consumed)) : nullptr;
    unsigned* counter = hv ? stream_control_words(st) : nullptr;
    if (hv) arm_host_words(h_consumed, 2);        // (also when the copy node below delivers: it overwrites the words)
    if (!counter) hv = nullptr;
    zig_compact_kernel<<<(unsigned)w.nb, kBlock, 0, st>>>(w.flag, w.sums, w.val, w.len, n_raw, offset, n, d_out,
                                                          w.result, hv, counter);
    OBE_CHECK_LAUNCH("zig_compact_kernel");
    if (defer_host_sync()) {       // {consumed, found} -> h_consumed[0..1]; the caller checks them after its wait
        if (!hv) OBE_HIP_TRY(hipMemcpyAsync(h_consumed, w.result, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        return 0;
    }
    int64_t host[2];
    OBE_HIP_TRY(hipMemcpyAsync(host, w.result, sizeof(host), hipMemcpyDeviceToHost, st));
    OBE_HIP_TRY(hipStreamSynchronize(st));
    if (obe_ziggurat_check(host[0], host[1], n, n_raw, offset)) {
        *h_consumed = -1;
        return 1;      /* OBE_RNG_NEED_MORE */
    }
    *h_consumed = host[0];
    return 0;
}

int obe_ziggurat_normal(const uint64_t* d_raw, int64_t n_raw, int64_t offset, const void* d_tables, int64_t n,
                        double* d_out, int64_t* h_consumed, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_raw || !d_tables || !d_out || !h_consumed || n <= 0 || offset < 0 || offset >= n_raw)
        return bad_arg("obe_ziggurat_normal: bad pointer/size");
    if (!d_ws || ws_bytes < obe_ziggurat_workspace_bytes(n_raw)) return bad_arg("obe_ziggurat_normal: workspace too small");
    if (n_raw >= (int64_t)1 << 31) return bad_arg("obe_ziggurat_normal: n_raw must be < 2^31");
    hipStream_t st = as_stream(stream);
    const ZigWs w = zig_carve(d_ws, n_raw);
    zig_classify_kernel<<<stream_blocks(n_raw, kBlock), kBlock, 0, st>>>(d_raw, n_raw, zig_tables(d_tables), w.val, w.len,
                                                                         w.result);
    OBE_CHECK_LAUNCH("zig_classify_kernel");
    return zig_finish(w, n_raw, offset, n, d_out, h_consumed, st);
}

// Uniforms and normals of one resample straight from the generator state, without a buffer of raw values:
// stage 1 (obe_pcg64_uniforms_classify) = the n_uniform uniforms and the classification of the n_raw_normal
// raw positions behind them in ONE launch; stage 2 (obe_ziggurat_finish) = start flags, scan, compaction of
// the first n normals.  Two calls so that a caller can enqueue what only needs the uniforms in between.
int obe_pcg64_uniforms_classify(const uint64_t* h_state4, int64_t n_uniform, int64_t n_raw_normal, double* d_uniforms,
                                const void* d_tables, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!h_state4 || !d_tables || !d_ws || n_uniform < 0 || n_raw_normal <= 0 || (n_uniform > 0 && !d_uniforms))
        return bad_arg("obe_pcg64_uniforms_classify: bad pointer/size");
    if (ws_bytes < obe_ziggurat_workspace_bytes(n_raw_normal)) return bad_arg("obe_pcg64_uniforms_classify: workspace too small");
    if (n_raw_normal >= (int64_t)1 << 31) return bad_arg("obe_pcg64_uniforms_classify: n_raw_normal must be < 2^31");
    PcgArgs a;
    a.state = U128{h_state4[0], h_state4[1]};
    a.inc = U128{h_state4[2], h_state4[3]};
    if ((a.inc.lo & 1) == 0) return bad_arg("obe_pcg64_uniforms_classify: PCG64 increment must be odd");
    const int64_t total = n_uniform + n_raw_normal;
    static const int forced = getenv("OBE_RNG_BLOCKS") ? atoi(getenv("OBE_RNG_BLOCKS")) : 0;      // tuning aid
    const int cap = forced > 0 && forced <= kRngBlocksMax ? forced : 1024;       // 34 us at 1024 (5.8 M positions), 36 at 2048, 45 at 512
    const int blocks = static_cast<int>(std::min<int64_t>(cap, (total + kBlock - 1) / kBlock));
    lcg_jump(a.inc, (uint64_t)blocks * kBlock, a.strideA, a.strideC);
    const ZigWs w = zig_carve(d_ws, n_raw_normal);
    const ZigTables t = zig_tables(d_tables);
    pcg_uniform_classify_kernel<<<blocks, kBlock, 0, as_stream(stream)>>>(a, make_jump_table(a.inc), pcg_mult(), n_uniform,
                                                                           n_raw_normal, t,
                                                                           d_uniforms, w.val, w.len, w.result, w.slow,
                                                                           w.slow_count);
    OBE_CHECK_LAUNCH("pcg_uniform_classify_kernel");
    zig_slow_kernel<<<blocks, kSlowPerBlock, 0, as_stream(stream)>>>(a.inc, pcg_mult(), n_raw_normal, t, w.slow,
                                                                      w.slow_count, w.val, w.len);
    OBE_CHECK_LAUNCH("zig_slow_kernel");
    return 0;
}

int obe_ziggurat_finish(int64_t n_raw_normal, int64_t n, double* d_out, int64_t* h_consumed, void* d_ws,
                        int64_t ws_bytes, void* stream) {
    if (!d_out || !h_consumed || !d_ws || n <= 0 || n_raw_normal <= 0) return bad_arg("obe_ziggurat_finish: bad pointer/size");
    if (ws_bytes < obe_ziggurat_workspace_bytes(n_raw_normal)) return bad_arg("obe_ziggurat_finish: workspace too small");
    return zig_finish(zig_carve(d_ws, n_raw_normal), n_raw_normal, 0, n, d_out, h_consumed, as_stream(stream));
}

int obe_ziggurat_check(int64_t consumed, int64_t found, int64_t n, int64_t n_raw, int64_t offset) {
    // the n-th normal must exist and end well inside the buffer (positions within
    // kMaxLen of the end may be unclassifiable and would break the chain)
    if (found < n || consumed <= 0 || consumed > n_raw - offset - 2 * kMaxLen) {
        set_error("obe_ziggurat_normal: raw buffer too short for the requested normals");
        return 1;
    }
    return 0;
}

}  // extern "C"
