// K3 building blocks shared by the moment kernels (obe_moments.hip) and the Bayes update that
// produces the first moments of its posterior in the same pass (obe_update.hip): block partial
// sums by wave reduce-scatter, the single-workgroup fold, and the per-particle accumulation and
// the derivation of mean / std — one definition each, so that both routes give the same bits.
#pragma once

#include <algorithm>
#include <cstdlib>

#include "obe_common.h"

namespace obe {

#ifndef OBE_MOM_BLOCKS
#define OBE_MOM_BLOCKS 256
#endif
// grid cap for the moment passes: one workgroup per CU.  The streaming itself is as fast with 1024
// (12-15 us at D = 10, 524 288 particles either way), but every workgroup ends with a block
// reduction of up to 136 values and leaves a row of partials for the single-workgroup fold.
constexpr int kMomBlocks = OBE_MOM_BLOCKS;
// Threads per workgroup of the passes that accumulate FIRST moments (moments_pass1, the update's normalisation pass, the
// constraint mask's second half — one grid and one block shape for all three, so that whichever fills the K3 block
// leaves the same bits).  The grid is one workgroup per CU (the fold of the rows is what the last one pays for), so the
// waves in flight per SIMD are set by the block size: see profiles/r06_update_moments.txt for the measurement.
#ifndef OBE_MOM_THREADS
#define OBE_MOM_THREADS 256
#endif
constexpr int kMomThreads = OBE_MOM_THREADS;
static_assert(kMomThreads % kWave == 0 && kMomThreads >= kBlock && kMomThreads <= 1024, "whole wavefronts, 256..1024 threads");
#ifndef OBE_MOM_UNROLL
#define OBE_MOM_UNROLL 1
#endif

// Wave-wide sums of C values (C a power of two <= 64) by reduce-scatter: at every level of the
// butterfly a lane hands half of its values to its partner and adds the partner's copies of the half
// it keeps, so the tree over the 64 lanes costs C - 1 exchanges (+ one per remaining level once a
// single value is left) instead of 6 C.  The pairing (lane ^ 32, ^ 16, ...) and hence the association
// of every sum is that of wave_sum(): bit-identical totals.  Afterwards v[0] of lane L holds the total
// of value  reduce_scatter_index<C>(L).
template <int C, int O>
__device__ __forceinline__ void wave_reduce_scatter(double* v, int lane) {
    if constexpr (O >= 1) {
        if constexpr (C > 1) {
            const bool upper = (lane & O) != 0;
#pragma unroll
            for (int i = 0; i < C / 2; ++i) {
                const double send = upper ? v[i] : v[i + C / 2];
                const double keep = upper ? v[i + C / 2] : v[i];
                v[i] = keep + __shfl_xor(send, O, kWave);
            }
            wave_reduce_scatter<C / 2, O / 2>(v, lane);
        } else {
            v[0] = v[0] + __shfl_xor(v[0], O, kWave);
            wave_reduce_scatter<1, O / 2>(v, lane);
        }
    }
}
template <int C>
__device__ __forceinline__ int reduce_scatter_index(int lane) {
    int idx = 0, c = C;
    for (int o = kWave / 2; o >= 1 && c > 1; o >>= 1) {
        c >>= 1;
        if (lane & o) idx += c;
    }
    return idx;
}
constexpr int pow2_ceil(int n) { return n <= 1 ? 1 : 2 * pow2_ceil((n + 1) / 2); }

// The NV block sums with ONE barrier: every wavefront reduces its values (in groups of up to 64, by
// reduce-scatter) and parks the totals in LDS, then thread k adds the wave sums of value k in wave
// order — the same arithmetic as NV calls of block_sum (shuffle tree, then the waves in order).
// (NV separate shuffle trees cost ~13 000 cycles per wave at NV = 55: the epilogue, not the
// streaming, set the duration of the covariance pass.)
template <int NV, int G0>
__device__ __forceinline__ void reduce_value_groups(const double (&v)[NV], double* __restrict__ red_row, int lane) {
    if constexpr (G0 < NV) {
        constexpr int CNT = NV - G0 < kWave ? NV - G0 : kWave;      // values in this group
        constexpr int C = pow2_ceil(CNT);
        double t[C];
#pragma unroll
        for (int i = 0; i < C; ++i) t[i] = i < CNT ? v[G0 + (i < CNT ? i : 0)] : 0.0;
        wave_reduce_scatter<C, kWave / 2>(t, lane);
        const int idx = reduce_scatter_index<C>(lane);
        if ((lane & (kWave / C - 1)) == 0 && idx < CNT) red_row[G0 + idx] = t[0];      // one lane per value
        reduce_value_groups<NV, G0 + kWave>(v, red_row, lane);
    }
}

// PUBLISHED: the row is read by another workgroup of the same launch (the last one to arrive folds,
// obe_common.h: arrive_last) and is therefore stored write-through.
template <int NV, bool PUBLISHED = false, int NT = kBlock>
__device__ __forceinline__ void store_block_partials(double (&v)[NV], double* __restrict__ partials) {
    constexpr int NW = NT / kWave;
    __shared__ double red[NW][NV];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    reduce_value_groups<NV, 0>(v, red[wid], lane);
    __syncthreads();
    for (int k = threadIdx.x; k < NV; k += NT) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += red[i][k];
        if constexpr (PUBLISHED) store_published(partials + (int64_t)blockIdx.x * NV + k, s);
        else partials[(int64_t)blockIdx.x * NV + k] = s;
    }
}

// One particle's contribution to  [0] sum w, [1] sum w*w, [2+i] sum x_i*w, [2+D+i] sum (x_i*x_i)*w
template <int D>
__device__ __forceinline__ void accumulate_first_moments(double (&v)[2 + 2 * D], double wp, const double (&xi)[D]) {
    v[0] += wp;
    v[1] += wp * wp;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        v[2 + i] += xi[i] * wp;
        v[2 + D + i] += (xi[i] * xi[i]) * wp;
    }
}

// raw[k] = sum_b partials[b*nv + k]: one wavefront per value (strided partial sums, then the
// shuffle tree: a fixed order).  One workgroup of 16 waves; a wave takes four values at a time and
// issues their loads together — the partials come straight from HBM / another XCD's L2, and a wave
// that walked its values one after the other spent ~1.5 us of latency on each (19 us for the 55
// values of a 10-parameter covariance).  `vals` in LDS.
constexpr int kFoldThreads = 1024;
constexpr int kFoldBatch = 4;
// (NT threads in the folding workgroup: which wave folds a value does not change its sum)
// (PUBLISHED: the rows were stored write-through by other workgroups of THIS launch — store_published —
// and are read past this CU's L1 with sc1 loads, which stands in for an agent-scope acquire)
__device__ __forceinline__ double load_published(const double* p) { return load_published_f64(p); }
// (BATCH values per wave and trip: a 4-wave workgroup that folds inside the producing launch takes 8 at a
// time to keep as many loads in flight as the 16-wave fold kernels do with 4)
// Round 5 (late): the loads of ROWS = 4 row groups x BATCH values are ISSUED TOGETHER, then added in the order
// the one-row-at-a-time loop added them (lane L: rows L, L + 64, L + 128, ...: the same bits).  The old loop's
// `if (k < nv) s += load` compiled to load / s_waitcnt vmcnt(0) / add for every single value: the 256 rows x 23
// values of a 10-parameter update were 24 round trips to another XCD's L2 one behind the other (~10 us of the
// normalisation launch's 23), the 9 values of a 3-parameter one 12.  Rows and values beyond the end are loaded
// from the last valid one and not added.  Callers that know nv at compile time pass BATCH = ceil(nv / waves): one trip, no
// slot that is never valid.
constexpr int kFoldRows = 4;
template <int NT = kFoldThreads, bool PUBLISHED = false, int BATCH = kFoldBatch>
__device__ __forceinline__ void fold_values_block(const double* partials, int nb, int nv, double* vals) {
    constexpr int NW = NT / kWave;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    for (int k0 = wid; k0 < nv; k0 += NW * BATCH) {
        double s[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) s[u] = 0.0;
        for (int base = 0; base < nb; base += kWave * kFoldRows) {
            double v[kFoldRows][BATCH];
#pragma unroll
            for (int r = 0; r < kFoldRows; ++r) {
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    // (clamped, not skipped: every load is a real one whose value the selection below consumes —
                    // an sc1 load the compiler knows to be unused still has to be performed, into a register that
                    // is reused at once, and that costs a full s_waitcnt per row group)
                    const int b = base + r * kWave + lane, k = k0 + u * NW;
                    const double* q = partials + (int64_t)(b < nb ? b : nb - 1) * nv + (k < nv ? k : nv - 1);
                    v[r][u] = PUBLISHED ? load_published(q) : *q;
                }
            }
#pragma unroll
            for (int r = 0; r < kFoldRows; ++r) {
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const int b = base + r * kWave + lane, k = k0 + u * NW;
                    const double t = s[u] + v[r][u];
                    s[u] = b < nb && k < nv ? t : s[u];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) s[u] = wave_sum(s[u]);
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int k = k0 + u * NW;
            if (lane == 0 && k < nv) vals[k] = s[u];
        }
    }
    __syncthreads();
}

// Where a folding workgroup delivers: the K3 block on the device and the device view of the caller's
// page-locked copy (or NULL).  A caller that does not synchronise the stream arms the words it expects and
// waits for each of them (obe_common.h: arm_host_words / wait_host_words).
struct MomentsOut {
    double* out;
    double* host;
};

// cov = S * (1 / (W - W2/W))  (np.cov scales by the reciprocal); raw = the folded upper triangle
__device__ __forceinline__ void derive_covariance(const double* raw, int d, double* __restrict__ out,
                                                  double* __restrict__ host) {
    const double fact = out[0] - out[1] / out[0];
    const double scale = 1.0 / fact;
    double* cov = out + 2 + 4 * d;
    for (int e = threadIdx.x; e < d * d; e += blockDim.x) {
        int i = e / d, j = e % d;
        if (i > j) {
            const int t = i;
            i = j;
            j = t;
        }
        const int k = i * d - i * (i - 1) / 2 + (j - i);     // index of (i, j), i <= j, in the packed upper triangle
        const double c = raw[k] * scale;
        cov[e] = c;
        if (host) host[2 + 4 * d + e] = c;
    }
}

constexpr int kMaxMomentValues = kFastDims * (kFastDims + 1) / 2;     // >= 2 + 2 D (the templated folds' LDS)

// Thread i < d: mean, m1, m2, std of parameter i from the folded sums raw[] (2 + 2 d values), into the
// K3 block `out` ([0]=W [1]=W2 [2..) mean [2+D..) m1 [2+2D..) m2 [2+3D..) std) and, if not NULL, `host`
// (the device view of the caller's page-locked copy).  Thread 0 also stores W and W2.
__device__ __forceinline__ void derive_first_moments(const double* raw, int d, double* __restrict__ out,
                                                     double* __restrict__ host) {
    const int i = threadIdx.x;
    if (i == 0) {
        out[0] = raw[0];
        out[1] = raw[1];
        if (host) {
            host[0] = raw[0];
            host[1] = raw[1];
        }
    }
    if (i < d) {
        const double m1 = raw[2 + i], m2 = raw[2 + d + i];
        const double mean = m1 / raw[0];             // np.average: sum(x w) / sum(w)
        const double sd = sqrt(m2 - m1 * m1);        // particlepdf.py:211-214
        out[2 + i] = mean;
        out[2 + d + i] = m1;
        out[2 + 2 * d + i] = m2;
        out[2 + 3 * d + i] = sd;
        if (host) {
            host[2 + i] = mean;
            host[2 + d + i] = m1;
            host[2 + 2 * d + i] = m2;
            host[2 + 3 * d + i] = sd;
        }
    }
}

// obe_moments() without the final synchronisation (the caller arms and watches the host words it expects,
// or synchronises); *host_written tells whether h_out was page-locked, i.e. written by the kernels themselves
int moments_call(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles, const double* d_weights,
                 int32_t want_cov, double* d_out, double* h_out, void* d_ws, int64_t ws_bytes, hipStream_t st,
                 bool* host_written);

// Grid of the moment passes (and of the update's normalisation pass that shares them): one workgroup per CU
// up to 2 M particles; beyond that a wave per SIMD with only D + 1 loads in flight no longer fills HBM
// (tools/measure_update.py, D = 3, 16.8 M particles: 4.0 TB/s with 256 workgroups, 4.8 with 768; D = 10 is
// as fast with 256), so narrow clouds get up to three per CU.  kMomGridCap bounds it for the workspace.
constexpr int kMomGridCap = 3 * kMomBlocks;
#ifndef OBE_FIRST_MOM_PER_CU_DEFAULT
#define OBE_FIRST_MOM_PER_CU_DEFAULT 0
#endif
static_assert(kMomGridCap <= 1024, "obe_workspace_bytes sizes the moment partials for at most 1024 workgroups");
inline int moment_blocks(int64_t n, int d) {
    static const int forced = getenv("OBE_MOM_PER_CU") ? atoi(getenv("OBE_MOM_PER_CU")) : 0;     // tuning aid (1..3)
    const int per_cu = forced >= 1 && forced <= 3 ? forced : (n < ((int64_t)1 << 21) ? 1 : (d <= 4 ? 3 : (d <= 7 ? 2 : 1)));
    return static_cast<int>(std::min<int64_t>((int64_t)per_cu * kMomBlocks, (n + kBlock - 1) / kBlock));
}

// Grid of the passes that accumulate FIRST moments (moments_pass1, the update's normalisation pass, the
// constraint mask's second half): all three share one grid so that whichever fills the K3 block leaves the
// same bits.  Default: that of the covariance pass above.  OBE_FIRST_MOM_PER_CU=3 makes it the grid of the
// update's likelihood pass (768 workgroups) — what the one-launch update needs (obe_update.hip), since its
// threads keep their particles in registers from the likelihood to the moments.
inline int first_moment_blocks(int64_t n, int d) {
    static const int per_cu = [] {
        const char* e = getenv("OBE_FIRST_MOM_PER_CU");
        const int v = e ? atoi(e) : OBE_FIRST_MOM_PER_CU_DEFAULT;
        return v >= 1 && v <= 3 ? v : 0;
    }();
    if (per_cu == 0) {
        if (kMomThreads == kBlock) return moment_blocks(n, d);
        const int nb = moment_blocks(n, d);           // (the same count of workgroups, of more threads each)
        return static_cast<int>(std::min<int64_t>(nb, (n + kMomThreads - 1) / kMomThreads));
    }
    return static_cast<int>(std::min<int64_t>((int64_t)per_cu * kMomBlocks, (n + kMomThreads - 1) / kMomThreads));
}

}  // namespace obe
