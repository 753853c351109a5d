// K4 — multinomial resampling of the particle cloud (particlepdf.py:260-345).
//
//   cdf    device prefix sum of the weights, normalised by its last element — the CDF
//          that numpy's Generator.choice(p=w) builds (cumsum; cdf /= cdf[-1]).
//          Blocked reduce-then-scan: 8 B read (block sums) + 8 B read + 8 B write per
//          particle.  A strict mode replays np.cumsum's serial rounding on one
//          wavefront for bit-identical CDFs.
//   search searchsorted(cdf, u, side='right') per draw: binary search, ~log2(N)
//          L2-resident probes.
//   gather new[i,p] = old[i, idx[p]] + z[p,:] . F[i,:]  (+ optional contraction to the
//          mean): random 8 B reads, coalesced writes.
#include <cstdlib>

#include "obe_common.h"
#include "obe_moments.h"

namespace obe {

constexpr int kScanItems = 8;                         // contiguous weights per thread
constexpr int kScanTile = kBlock * kScanItems;        // 2048 weights per block

// Inclusive scan of one 2048-element tile held as 8 contiguous items per thread.
// Returns the tile total in every thread.  The same routine is used by the block-sum
// pass and by the final pass, so both see bit-identical values.
__device__ __forceinline__ double tile_scan(double (&v)[kScanItems], double* lds /* kBlock/kWave */) {
#pragma unroll
    for (int k = 1; k < kScanItems; ++k) v[k] = v[k - 1] + v[k];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    // inclusive scan of the thread totals across the wave
    double incl = v[kScanItems - 1];
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const double up = __shfl_up(incl, o, kWave);
        if (lane >= o) incl = up + incl;
    }
    __syncthreads();
    if (lane == kWave - 1) lds[wid] = incl;
    __syncthreads();
    double wave_off = 0.0, total = 0.0;
#pragma unroll
    for (int i = 0; i < kBlock / kWave; ++i) {
        if (i < wid) wave_off += lds[i];
        total += lds[i];
    }
    // exclusive prefix of this thread = inclusive prefix of its left neighbour
    const double prev = __shfl_up(incl, 1, kWave);
    const double thread_off = wave_off + (lane == 0 ? 0.0 : prev);
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] = thread_off + v[k];
    return total;
}

__device__ __forceinline__ void load_tile(const double* __restrict__ w, int64_t n, int64_t base,
                                          double (&v)[kScanItems]) {
    const int64_t i0 = base + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] = (i0 + k < n) ? w[i0 + k] : 0.0;
}

// smallest value of a tile (its padding counts as 0: only "is anything negative" is asked of it)
__device__ __forceinline__ double tile_min(const double (&v)[kScanItems], double* lds /* kBlock/kWave */) {
    double m = v[0];
#pragma unroll
    for (int k = 1; k < kScanItems; ++k) m = fmin(m, v[k]);
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) m = fmin(m, __shfl_down(m, o, kWave));
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    __syncthreads();
    if (lane == 0) lds[wid] = m;
    __syncthreads();
    double r = lds[0];
#pragma unroll
    for (int i = 1; i < kBlock / kWave; ++i) r = fmin(r, lds[i]);
    return r;
}

// What numpy's Generator.choice checks of p, in its order: NaN ("Probabilities contain NaN": the sum is
// NaN), then a negative entry ("Probabilities are not non-negative"), then the sum.  The host gets ONE
// number: the sum, or -inf when some weight is negative (no sum of probabilities is ever -inf otherwise).
__device__ __forceinline__ double total_for_validation(double total, double smallest) {
    return (total == total && smallest < 0.0) ? -INFINITY : total;
}

__global__ __launch_bounds__(kBlock) void scan_block_sums(const double* __restrict__ w, int64_t n,
                                                          double* __restrict__ block_sums,
                                                          double* __restrict__ block_mins) {
    __shared__ double lds[kBlock / kWave];
    double v[kScanItems];
    load_tile(w, n, (int64_t)blockIdx.x * kScanTile, v);
    const double smallest = block_mins ? tile_min(v, lds) : 0.0;
    const double total = tile_scan(v, lds);
    if (threadIdx.x == 0) {
        block_sums[blockIdx.x] = total;
        if (block_mins) block_mins[blockIdx.x] = smallest;
    }
}

// Exclusive scan of the block sums (nb = N/2048: 512 at 1M particles) by one workgroup,
// written in place; scalars[0] = total = offset[last] + sum[last], scalars[1] = the value the host
// validates (total_for_validation).
// (host_total: the device view of the caller's page-locked h_total, or NULL)
__global__ __launch_bounds__(kBlock) void scan_offsets(double* __restrict__ block_sums, int64_t nb,
                                                       const double* __restrict__ block_mins,
                                                       double* __restrict__ scalars, double* __restrict__ host_total) {
    __shared__ double lds[kBlock + 1];
    __shared__ double mins[kBlock];
    double m = 0.0;
    if (block_mins)
        for (int64_t i = threadIdx.x; i < nb; i += kBlock) m = fmin(m, block_mins[i]);
    mins[threadIdx.x] = m;
    const double total = block_exclusive_scan_inplace<double>(block_sums, nb, lds);     // (synchronises)
    // (a tree, not thread 0 walking 255 LDS words one read at a time: ~4 us of this one-workgroup kernel; the minimum
    // does not depend on the order)
    for (int o = kBlock / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) mins[threadIdx.x] = fmin(mins[threadIdx.x], mins[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        m = mins[0];
        scalars[0] = total;
        scalars[1] = total_for_validation(total, m);
        if (host_total) *host_total = scalars[1];
    }
}

__global__ __launch_bounds__(kBlock) void scan_write_cdf(const double* __restrict__ w, int64_t n,
                                                         const double* __restrict__ block_off,
                                                         const double* __restrict__ scalars,
                                                         double* __restrict__ cdf, int normalize) {
    __shared__ double lds[kBlock / kWave];
    double v[kScanItems];
    const int64_t base = (int64_t)blockIdx.x * kScanTile;
    load_tile(w, n, base, v);
    tile_scan(v, lds);
    const double off = block_off[blockIdx.x];
    const double total = scalars[0];
    const int64_t i0 = base + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        const int64_t i = i0 + k;
        if (i < n) {
            if (normalize) cdf[i] = (i == n - 1) ? 1.0 : (off + v[k]) / total;   // cdf[-1]/cdf[-1] == 1
            else cdf[i] = off + v[k];
        }
    }
}

__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// Strict mode: c_i = fl(c_{i-1} + w_i) in index order, exactly np.cumsum.  One
// wavefront: a coalesced 64-wide load, then 64 dependent adds fed by v_readlane.
__global__ __launch_bounds__(kWave) void cdf_strict_kernel(const double* __restrict__ w, int64_t n,
                                                           double* __restrict__ cdf,
                                                           double* __restrict__ scalars, int normalize) {
    const int lane = threadIdx.x;
    double run = 0.0, smallest = 0.0;
    for (int64_t base = 0; base < n; base += kWave) {
        const int64_t i = base + lane;
        const double x = (i < n) ? w[i] : 0.0;
        smallest = fmin(smallest, x);
        double mine = 0.0;
#pragma unroll
        for (int k = 0; k < kWave; ++k) {
            run = run + readlane_f64(x, k);
            if (lane == k) mine = run;
        }
        if (i < n) cdf[i] = mine;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) smallest = fmin(smallest, __shfl_down(smallest, o, kWave));
    if (lane == 0) {
        scalars[0] = run;
        scalars[1] = total_for_validation(run, smallest);
    }
    if (!normalize) return;
    __threadfence_block();
    for (int64_t base = 0; base < n; base += kWave) {
        const int64_t i = base + lane;
        if (i < n) cdf[i] = cdf[i] / run;
    }
}

// The uniforms of a small draw (randdraw's 30 parameter sets, good_setting's single one) travel
// as kernel arguments: no host-to-device copy.
constexpr int kMaxArgDraws = 64;
struct UniformArg {
    double u[kMaxArgDraws];
};

__device__ __forceinline__ int64_t search_right(const double* cdf, int64_t n, double x) {
    int64_t lo = 0, hi = n;                 // first i with cdf[i] > x
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// (total_host: the device view of the caller's page-locked sum(w) slot, or NULL — the last kernel of a small
// draw delivers it itself; a hipMemcpyAsync of 8 bytes would be a blit kernel of its own)
__global__ __launch_bounds__(kWave) void cdf_search_arg_kernel(const double* __restrict__ cdf, int64_t n, UniformArg ua,
                                                               int nd, int64_t* __restrict__ idx,
                                                               const double* __restrict__ total_src,
                                                               double* __restrict__ total_host) {
    // (sum(p) first and fenced: a caller that watches the page-locked index of a single draw — good_setting —
    // finds the sum there as well)
    if (threadIdx.x == 0 && total_host) {
        *total_host = *total_src;
        host_results_before_flag();
    }
    if ((int)threadIdx.x < nd) idx[threadIdx.x] = search_right(cdf, n, ua.u[threadIdx.x]);
}

// CDF + search in one launch for clouds of up to kSmallCloud particles: one workgroup walks the
// 2048-weight tiles with the same tile_scan / tile-offset arithmetic as the three-kernel path
// (identical CDF bits), then up to 64 threads run the searches.  total_out[0] = sum(w).
constexpr int64_t kSmallCloud = 65536;          // what the kernel can hold
// what it is used for: one workgroup walks a 2048-weight tile in ~2 us, the three-kernel scan costs
// ~15 us more in launches — measured crossover at ~15 000 particles (cycle of 200 settings x 30 draws:
// 92.9 vs 101 us at 10 000 particles, 107.8 vs 99.8 us at 20 000, 202 vs 146 us at 50 000)
constexpr int64_t kSmallCloudDefault = 14336;
constexpr int kSmallTiles = static_cast<int>(kSmallCloud / kScanTile);

__global__ __launch_bounds__(kBlock) void draw_small_kernel(const double* __restrict__ w, int64_t n,
                                                            double* __restrict__ cdf, UniformArg ua, int nd,
                                                            int64_t* __restrict__ idx, double* __restrict__ total_out,
                                                            double* __restrict__ total_host) {
    __shared__ double lds[kBlock / kWave];
    __shared__ double toff[kSmallTiles + 1];
    const int nb = static_cast<int>((n + kScanTile - 1) / kScanTile);
    double v[kScanItems];
    double smallest = 0.0;
    for (int b = 0; b < nb; ++b) {
        load_tile(w, n, (int64_t)b * kScanTile, v);
        smallest = fmin(smallest, tile_min(v, lds));
        const double total = tile_scan(v, lds);
        if (threadIdx.x == 0) toff[b] = total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {                 // exclusive scan of the tile totals, in tile order (scan_offsets)
        double run = 0.0;
        for (int b = 0; b < nb; ++b) {
            const double t = toff[b];
            toff[b] = run;
            run = run + t;
        }
        toff[nb] = run;
        total_out[0] = run;
        total_out[1] = total_for_validation(run, smallest);
        if (total_host) {
            *total_host = total_out[1];
            host_results_before_flag();
        }
    }
    __syncthreads();
    const double total = toff[nb];
    for (int b = 0; b < nb; ++b) {
        const int64_t base = (int64_t)b * kScanTile;
        load_tile(w, n, base, v);
        tile_scan(v, lds);
        const double off = toff[b];
        const int64_t i0 = base + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
        for (int k = 0; k < kScanItems; ++k) {
            const int64_t i = i0 + k;
            if (i < n) cdf[i] = (i == n - 1) ? 1.0 : (off + v[k]) / total;
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    if ((int)threadIdx.x < nd) idx[threadIdx.x] = search_right(cdf, n, ua.u[threadIdx.x]);
}

// idx = first i with cdf[i] > u   (searchsorted side='right')
__global__ __launch_bounds__(kBlock) void cdf_search_kernel(const double* __restrict__ cdf, int64_t n,
                                                            const double* __restrict__ u, int64_t nd,
                                                            int64_t* __restrict__ idx) {
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nd; j += (int64_t)gridDim.x * kBlock) {
        const double uj = u[j];
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= uj) lo = mid + 1;
            else hi = mid;
        }
        idx[j] = lo;
    }
}

// Guided search for many draws (a resample: N uniforms into an N-entry CDF).  A plain binary search
// makes ~11 cold, dependent accesses per draw once the shared top levels are cached.  The guide
// table has one entry per bucket [b/K, (b+1)/K), K = N/8 (a bucket spans 8 CDF entries = one cache
// line on average): guide[b] = #{i : cdf[i] < b/K}, built by K + 1 searches whose queries are SORTED
// (neighbouring threads walk the same path: cache-friendly).  A draw u in bucket b then satisfies
// guide[b] <= idx <= guide[b+1]  exactly — every entry below guide[b] is < b/K <= u and every entry
// from guide[b+1] on is >= (b+1)/K > u — so the same upper-bound search runs over that handful of
// entries: identical indices, ~2-3 cold accesses per draw (1 M draws at 1 M entries: 70 -> 30 us).
constexpr int64_t kGuideMinEntries = 32768;
__host__ __device__ __forceinline__ int64_t guide_buckets(int64_t n) { return n >> 3; }

__device__ __forceinline__ double guide_threshold(int64_t b, int64_t k) { return (double)b / (double)k; }

__global__ __launch_bounds__(kBlock) void cdf_guide_kernel(const double* __restrict__ cdf, int64_t n, int64_t k,
                                                           int32_t* __restrict__ guide) {
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b <= k; b += (int64_t)gridDim.x * kBlock) {
        const double t = guide_threshold(b, k);
        int64_t lo = 0, hi = n;                 // first i with cdf[i] >= t
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (cdf[mid] < t) lo = mid + 1;
            else hi = mid;
        }
        guide[b] = static_cast<int32_t>(lo);
    }
}

__global__ __launch_bounds__(kBlock) void cdf_search_guided_kernel(const double* __restrict__ cdf, int64_t n,
                                                                   int64_t k, const int32_t* __restrict__ guide,
                                                                   const double* __restrict__ u, int64_t nd,
                                                                   int64_t* __restrict__ idx) {
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nd; j += (int64_t)gridDim.x * kBlock) {
        const double uj = u[j];
        int64_t lo = 0, hi = n;
        if (uj >= 0.0 && uj < 1.0) {            // (anything else: the plain search over the whole CDF)
            int64_t b = static_cast<int64_t>(uj * (double)k);
            b = b < 0 ? 0 : (b > k - 1 ? k - 1 : b);
            while (b > 0 && guide_threshold(b, k) > uj) --b;            // the product may round across a
            while (b < k - 1 && guide_threshold(b + 1, k) <= uj) ++b;   // bucket edge: at most one step
            lo = guide[b];
            hi = guide[b + 1];
        }
        while (lo < hi) {                       // first i in [lo, hi) with cdf[i] > u
            const int64_t mid = (lo + hi) >> 1;
            if (cdf[mid] <= uj) lo = mid + 1;
            else hi = mid;
        }
        idx[j] = lo;
    }
}

__global__ __launch_bounds__(kBlock) void gather_columns_kernel(const double* __restrict__ x, int64_t ld, int d,
                                                                int64_t n_src, const int64_t* __restrict__ idx,
                                                                int64_t nd, double* __restrict__ out, int64_t ld_out) {
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nd; j += (int64_t)gridDim.x * kBlock) {
        int64_t src = idx[j];
        src = src < 0 ? 0 : (src >= n_src ? n_src - 1 : src);
        for (int i = 0; i < d; ++i) out[(int64_t)i * ld_out + j] = x[(int64_t)i * ld + src];
    }
}

struct NudgeArgs {
    double factor[kFastDims * kFastDims];   // F row-major (D x D)
    double mean[kFastDims];
    double a, one_minus_a, uniform_w;
    int d, scale;
};

// One thread per new particle.  D is a template parameter so that the z row lives in
// registers, the D x D factor is read as wave-uniform kernel arguments and both loops
// unroll (the generic runtime-D loop took 343 us at D = 10, N = 524 288).
// (D, N) SoA -> (N, D) AoS copy of the old cloud, lane-contiguous on both sides (LDS transpose).  The
// gather of resample_kernel reads D random 8-byte values per particle: from the SoA layout every one
// of them pulls its own 64-byte sector (8 x the bytes); from the AoS copy a particle is one or two
// sectors.  Worth the extra pass from D = 2 on.
template <int D>
__global__ __launch_bounds__(kBlock) void soa_to_aos_kernel(const double* __restrict__ x, int64_t ld, int64_t n,
                                                            double* __restrict__ aos) {
    __shared__ double tile[kBlock * D];
    for (int64_t p0 = (int64_t)blockIdx.x * kBlock; p0 < n; p0 += (int64_t)gridDim.x * kBlock) {
        const int64_t p = p0 + threadIdx.x;
        __syncthreads();
        if (p < n) {
#pragma unroll
            for (int i = 0; i < D; ++i) tile[threadIdx.x * D + i] = x[(int64_t)i * ld + p];
        }
        __syncthreads();
        const int64_t run = (n - p0 < kBlock ? n - p0 : kBlock) * D;
        for (int64_t e = threadIdx.x; e < run; e += kBlock) aos[p0 * D + e] = tile[e];
    }
}

// AOS: `old` is the (N, D) copy made by soa_to_aos_kernel (ld_old unused)
// MASK (round 5): the first half of OptBayesExptNoiseParameter.enforce_parameter_constraints
// (obe_noiseparam.py:57-79), which pdf_update() calls right after this resample, done here: a new particle
// whose parameter `row` <= 0 for any row of mask_bits gets weight 0 instead of 1/N, and the workgroup leaves
// the partial sums of the weights and of the zeroed count that mask_kernel (obe_update.hip) would leave —
// the same grid, the same per-thread order, the same block reductions: the same bits — for
// obe_mask_renorm_moments().  One launch and one pass over the noise rows and the weights less.
template <int D, bool AOS, bool MASK = false>
__global__ __launch_bounds__(kBlock) void resample_kernel(NudgeArgs na, const double* __restrict__ old, int64_t ld_old,
                                                          int64_t n, const int64_t* __restrict__ idx,
                                                          const double* __restrict__ z, double* __restrict__ out,
                                                          int64_t ld_new, double* __restrict__ weights,
                                                          unsigned mask_bits = 0u, double* __restrict__ psum = nullptr,
                                                          double* __restrict__ pcount = nullptr) {
    // the (N, D) row-major normals of a workgroup's 256 particles are one contiguous run: read it
    // lane-contiguously into LDS (a thread reading its own row makes every load touch 64 lines)
    __shared__ double zs[kBlock * D];
    __shared__ double red[kBlock / kWave];
    double acc = 0.0, cnt = 0.0;
    for (int64_t p0 = (int64_t)blockIdx.x * kBlock; p0 < n; p0 += (int64_t)gridDim.x * kBlock) {
        const int64_t p = p0 + threadIdx.x;
        const int64_t run = (n - p0 < kBlock ? n - p0 : kBlock) * D;
        __syncthreads();       // the previous trip's rows have been consumed
        if (run == (int64_t)kBlock * D) {
            // a full tile: the D loads of a thread in flight together (as a loop it compiled to load / wait / LDS
            // store, D dependent round trips to HBM in front of every gather — 10 of them for the 10-parameter model)
            double t[D];
#pragma unroll
            for (int j = 0; j < D; ++j) t[j] = z[p0 * D + threadIdx.x + j * kBlock];
#pragma unroll
            for (int j = 0; j < D; ++j) zs[threadIdx.x + j * kBlock] = t[j];
        } else {
            for (int64_t e = threadIdx.x; e < run; e += kBlock) zs[e] = z[p0 * D + e];
        }
        __syncthreads();
        if (p >= n) continue;
        int64_t src = idx[p];
        src = src < 0 ? 0 : (src >= n ? n - 1 : src);
        double zr[D], x0[D];
#pragma unroll
        for (int j = 0; j < D; ++j) zr[j] = zs[threadIdx.x * D + j];
#pragma unroll
        for (int i = 0; i < D; ++i)                                              // D independent gathers in flight
            x0[i] = AOS ? old[src * D + i] : old[(int64_t)i * ld_old + src];
        bool bad = false;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            // (z @ F.T)[p, i]: FMA chain from zero in j order — bit-identical to the
            // dgemm NumPy's multivariate_normal uses (checked against numpy 2.2.6/OpenBLAS)
            double nudge = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) nudge = fma(zr[j], na.factor[i * D + j], nudge);
            double v = x0[i] + nudge;
            if (na.scale) {
                const double va = v * na.a;
                const double mc = na.mean[i] * na.one_minus_a;
                v = va + mc;
            }
            out[(int64_t)i * ld_new + p] = v;
            if constexpr (MASK) bad = bad || (((mask_bits >> i) & 1u) && v <= 0.0);
        }
        if constexpr (MASK) {
            const double w = bad ? 0.0 : na.uniform_w;
            weights[p] = w;
            acc += w;
            if (bad) cnt += 1.0;
        } else {
            weights[p] = na.uniform_w;
        }
    }
    if constexpr (MASK) {
        const double s = block_sum(acc, red);
        __syncthreads();
        const double c = block_sum(cnt, red);
        if (threadIdx.x == 0) {
            psum[blockIdx.x] = s;
            pcount[blockIdx.x] = c;
        }
    }
}

// Wide clouds (D > OBE_FAST_DIMS; any D): the same gather + nudge with run-time loops — the D x D factor and the mean
// in device memory (`fm`: F row-major, then the mean; every lane reads the same element: scalar loads), the gather
// from the (D, N) array, the normals row from (N, D).  The same FMA chain per output as resample_kernel.
__global__ __launch_bounds__(kBlock) void resample_wide_kernel(const double* __restrict__ fm, int d, double a,
                                                               double one_minus_a, double uniform_w, int scale,
                                                               const double* __restrict__ old, int64_t ld_old, int64_t n,
                                                               const int64_t* __restrict__ idx,
                                                               const double* __restrict__ z, double* __restrict__ out,
                                                               int64_t ld_new, double* __restrict__ weights) {
    const double* __restrict__ mean = fm + (int64_t)d * d;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        int64_t src = idx[p];
        src = src < 0 ? 0 : (src >= n ? n - 1 : src);
        const double* __restrict__ zr = z + p * d;
        for (int i = 0; i < d; ++i) {
            double nudge = 0.0;
            for (int j = 0; j < d; ++j) nudge = fma(zr[j], fm[(int64_t)i * d + j], nudge);
            double v = old[(int64_t)i * ld_old + src] + nudge;
            if (scale) {
                const double va = v * a;
                const double mc = mean[i] * one_minus_a;
                v = va + mc;
            }
            out[(int64_t)i * ld_new + p] = v;
        }
        weights[p] = uniform_w;
    }
}

template <int D>
static int launch_soa_to_aos(const double* d_old, int64_t ld_old, int64_t n, double* aos, hipStream_t st) {
    soa_to_aos_kernel<D><<<stream_blocks(n, kBlock), kBlock, 0, st>>>(d_old, ld_old, n, aos);
    OBE_CHECK_LAUNCH("soa_to_aos_kernel");
    return 0;
}

// d_old == nullptr: d_ws already IS the (N, D) copy of the old cloud (obe_resample_particles_aos)
template <int D>
static int launch_resample(const NudgeArgs& na, const double* d_old, int64_t ld_old, int64_t n, const int64_t* d_idx,
                           const double* d_normals, double* d_new, int64_t ld_new, double* d_weights, void* d_ws,
                           int64_t ws_bytes, hipStream_t st, unsigned mask_bits = 0u, double* d_mask_partials = nullptr) {
    const int blocks = stream_blocks(n, kBlock);
    if (!d_old && mask_bits) {
        resample_kernel<D, true, true><<<blocks, kBlock, 0, st>>>(na, static_cast<const double*>(d_ws), 0, n, d_idx,
                                                                  d_normals, d_new, ld_new, d_weights, mask_bits,
                                                                  d_mask_partials, d_mask_partials + kMaxBlocks);
    } else if (!d_old) {
        resample_kernel<D, true><<<blocks, kBlock, 0, st>>>(na, static_cast<const double*>(d_ws), 0, n, d_idx, d_normals,
                                                            d_new, ld_new, d_weights);
    } else if (D >= 2 && n >= 65536 && d_ws && ws_bytes >= (int64_t)sizeof(double) * D * n) {
        double* aos = static_cast<double*>(d_ws);
        soa_to_aos_kernel<D><<<blocks, kBlock, 0, st>>>(d_old, ld_old, n, aos);
        OBE_CHECK_LAUNCH("soa_to_aos_kernel");
        resample_kernel<D, true><<<blocks, kBlock, 0, st>>>(na, aos, 0, n, d_idx, d_normals, d_new, ld_new, d_weights);
    } else {
        resample_kernel<D, false><<<blocks, kBlock, 0, st>>>(na, d_old, ld_old, n, d_idx, d_normals, d_new, ld_new,
                                                             d_weights);
    }
    OBE_CHECK_LAUNCH("resample_kernel");
    return 0;
}

}  // namespace obe

using namespace obe;

extern "C" {

static int scan_common(const char* who, const double* d_x, int64_t n, int32_t strict_order, int normalize,
                       double* d_out, double* h_total, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_x || !d_out || n <= 0) return bad_arg(who);
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    const int64_t need = (2 * nb + 8) * (int64_t)sizeof(double);
    if (!d_ws || ws_bytes < need) return bad_arg("scan: workspace too small");
    double* scalars = static_cast<double*>(d_ws);
    double* block_sums = scalars + 8;
    double* block_mins = normalize ? block_sums + nb : nullptr;       // a CDF of weights: validated like numpy's p
    hipStream_t st = as_stream(stream);
    double* hv = nullptr;
    if (strict_order) {
        cdf_strict_kernel<<<1, kWave, 0, st>>>(d_x, n, d_out, scalars, normalize);
        OBE_CHECK_LAUNCH("cdf_strict_kernel");
    } else {
        scan_block_sums<<<(unsigned)nb, kBlock, 0, st>>>(d_x, n, block_sums, block_mins);
        OBE_CHECK_LAUNCH("scan_block_sums");
        // a deferred, page-locked h_total is stored by the kernel itself (no copy node); the caller reads it
        // after synchronising with anything later on the stream
        hv = defer_host_sync() ? static_cast<double*>(device_view_of_host(h_total)) : nullptr;
        scan_offsets<<<1, kBlock, 0, st>>>(block_sums, nb, block_mins, scalars, hv);
        OBE_CHECK_LAUNCH("scan_offsets");
        scan_write_cdf<<<(unsigned)nb, kBlock, 0, st>>>(d_x, n, block_sums, scalars, d_out, normalize);
        OBE_CHECK_LAUNCH("scan_write_cdf");
    }
    if (h_total && !hv) {
        OBE_HIP_TRY(hipMemcpyAsync(h_total, scalars + 1, sizeof(double), hipMemcpyDeviceToHost, st));
        if (!defer_host_sync()) OBE_HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

int obe_weight_cdf(const double* d_weights, int64_t n_particles, int32_t strict_order, double* d_cdf,
                   double* h_total, void* d_ws, int64_t ws_bytes, void* stream) {
    return scan_common("obe_weight_cdf: bad pointer/size", d_weights, n_particles, strict_order, 1, d_cdf, h_total,
                       d_ws, ws_bytes, stream);
}

int obe_cumsum(const double* d_x, int64_t n, int32_t strict_order, double* d_out, void* d_ws, int64_t ws_bytes,
               void* stream) {
    return scan_common("obe_cumsum: bad pointer/size", d_x, n, strict_order, 0, d_out, nullptr, d_ws, ws_bytes,
                       stream);
}

// Sweeper composition: utility of the sweep start_i..stop_i = difference of the running
// point-utility integral at its two ends over the sweep's cost.
__global__ __launch_bounds__(kBlock) void interval_utility_kernel(const double* __restrict__ cum, int64_t n,
                                                                  const int64_t* __restrict__ pairs, int64_t np,
                                                                  const double* __restrict__ cost,
                                                                  double cost_of_new_sweep,
                                                                  double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < np; i += (int64_t)gridDim.x * kBlock) {
        const longlong2 se = reinterpret_cast<const longlong2*>(pairs)[i];      // one 16 B load
        int64_t a = se.x, b = se.y;
        const double c = cost ? cost[i] : (double)(b - a) + cost_of_new_sweep;   // obe_sweeper.py:119-120
        a = a < 0 ? a + n : a;                                                  // NumPy index wrap
        b = b < 0 ? b + n : b;
        a = a < 0 ? 0 : (a >= n ? n - 1 : a);
        b = b < 0 ? 0 : (b >= n ? n - 1 : b);
        out[i] = (cum[b] - cum[a]) / c;
    }
}

int obe_interval_utility(const double* d_cum, int64_t n_settings, const int64_t* d_pairs, int64_t n_pairs,
                         const double* d_cost, double cost_of_new_sweep, double* d_utility, void* stream) {
    if (!d_cum || !d_pairs || !d_utility || n_settings <= 0 || n_pairs <= 0)
        return bad_arg("obe_interval_utility: bad pointer/size");
    interval_utility_kernel<<<stream_blocks(n_pairs, kBlock), kBlock, 0, as_stream(stream)>>>(
        d_cum, n_settings, d_pairs, n_pairs, d_cost, cost_of_new_sweep, d_utility);
    OBE_CHECK_LAUNCH("interval_utility_kernel");
    return 0;
}

int obe_draw_indices(const double* d_weights, int64_t n_particles, int32_t strict_order, int32_t cdf_is_fresh,
                     double* d_cdf, const double* h_uniforms, int32_t n_draws, int64_t* d_idx,
                     double* h_total_pinned, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_weights || !d_cdf || !h_uniforms || !d_idx || n_particles <= 0 || n_draws <= 0 || n_draws > kMaxArgDraws)
        return bad_arg("obe_draw_indices: bad pointer/size (at most 64 draws)");
    if (!d_ws || ws_bytes < 8 * (int64_t)sizeof(double)) return bad_arg("obe_draw_indices: workspace too small");
    hipStream_t st = as_stream(stream);
    UniformArg ua{};
    for (int i = 0; i < n_draws; ++i) ua.u[i] = h_uniforms[i];
    double* scalars = static_cast<double*>(d_ws);
    if (cdf_is_fresh) {
        cdf_search_arg_kernel<<<1, kWave, 0, st>>>(d_cdf, n_particles, ua, n_draws, d_idx, nullptr, nullptr);
        OBE_CHECK_LAUNCH("cdf_search_arg_kernel");
        return 0;
    }
    double* hv = static_cast<double*>(device_view_of_host(h_total_pinned));      // NULL for pageable memory
    static const int64_t small_limit = getenv("OBE_SMALL_CLOUD") ? atoll(getenv("OBE_SMALL_CLOUD")) : kSmallCloudDefault;   // tuning aid
    if (!strict_order && n_particles <= std::min(small_limit, kSmallCloud)) {
        draw_small_kernel<<<1, kBlock, 0, st>>>(d_weights, n_particles, d_cdf, ua, n_draws, d_idx, scalars, hv);
        OBE_CHECK_LAUNCH("draw_small_kernel");
    } else {
        if (int rc = scan_common("obe_draw_indices: bad pointer/size", d_weights, n_particles, strict_order, 1, d_cdf,
                                 nullptr, d_ws, ws_bytes, stream))
            return rc;
        cdf_search_arg_kernel<<<1, kWave, 0, st>>>(d_cdf, n_particles, ua, n_draws, d_idx, scalars + 1, hv);
        OBE_CHECK_LAUNCH("cdf_search_arg_kernel");
    }
    if (h_total_pinned && !hv)      // pageable memory after all: an asynchronous copy
        OBE_HIP_TRY(hipMemcpyAsync(h_total_pinned, scalars + 1, sizeof(double), hipMemcpyDeviceToHost, st));
    return 0;
}

// Systematic resampling (extension): one uniform u0, draws at (i + u0) / n_draws — sorted, so
// neighbouring threads probe neighbouring CDF entries.
__global__ __launch_bounds__(kBlock) void systematic_search_kernel(const double* __restrict__ cdf, int64_t n,
                                                                   double u0, int64_t nd,
                                                                   int64_t* __restrict__ idx) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nd; i += (int64_t)gridDim.x * kBlock)
        idx[i] = search_right(cdf, n, ((double)i + u0) / (double)nd);
}

int obe_systematic_indices(const double* d_cdf, int64_t n, double u0, int64_t n_draws, int64_t* d_idx_out,
                           void* stream) {
    if (!d_cdf || !d_idx_out || n <= 0 || n_draws <= 0 || !(u0 >= 0.0 && u0 < 1.0))
        return bad_arg("obe_systematic_indices: bad pointer/size or u0 outside [0, 1)");
    systematic_search_kernel<<<stream_blocks(n_draws, kBlock), kBlock, 0, as_stream(stream)>>>(d_cdf, n, u0, n_draws,
                                                                                             d_idx_out);
    OBE_CHECK_LAUNCH("systematic_search_kernel");
    return 0;
}

int obe_cdf_search(const double* d_cdf, int64_t n, const double* d_uniforms, int64_t n_draws, int64_t* d_idx_out,
                   void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_cdf || !d_uniforms || !d_idx_out || n <= 0 || n_draws <= 0) return bad_arg("obe_cdf_search: bad pointer/size");
    hipStream_t st = as_stream(stream);
    // many draws from a large CDF (a resample): build the guide table first
    const int64_t k = guide_buckets(n);
    if (d_ws && n >= kGuideMinEntries && n < ((int64_t)1 << 31) && n_draws >= n / 4 &&
        ws_bytes >= (k + 2) * (int64_t)sizeof(int32_t)) {
        int32_t* guide = static_cast<int32_t*>(d_ws);
        cdf_guide_kernel<<<stream_blocks(k + 1, kBlock), kBlock, 0, st>>>(d_cdf, n, k, guide);
        OBE_CHECK_LAUNCH("cdf_guide_kernel");
        cdf_search_guided_kernel<<<stream_blocks(n_draws, kBlock), kBlock, 0, st>>>(d_cdf, n, k, guide, d_uniforms,
                                                                                    n_draws, d_idx_out);
        OBE_CHECK_LAUNCH("cdf_search_guided_kernel");
        return 0;
    }
    cdf_search_kernel<<<stream_blocks(n_draws, kBlock), kBlock, 0, st>>>(d_cdf, n, d_uniforms, n_draws, d_idx_out);
    OBE_CHECK_LAUNCH("cdf_search_kernel");
    return 0;
}

int obe_gather_columns(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                       const int64_t* d_idx, int64_t n_draws, double* d_out, int64_t ld_out, void* stream) {
    if (!d_particles || !d_idx || !d_out || n_draws <= 0 || n_dims < 1 || n_particles <= 0)
        return bad_arg("obe_gather_columns: bad pointer/size");
    gather_columns_kernel<<<stream_blocks(n_draws, kBlock), kBlock, 0, as_stream(stream)>>>(
        d_particles, ld_p, n_dims, n_particles, d_idx, n_draws, d_out, ld_out);
    OBE_CHECK_LAUNCH("gather_columns_kernel");
    return 0;
}

static int resample_particles(const double* d_old, int64_t ld_old, int32_t n_dims, int64_t n_particles,
                              const int64_t* d_idx, const double* d_normals, const double* h_factor,
                              const double* h_mean, double a_param, int32_t scale, double* d_new, int64_t ld_new,
                              double* d_weights, void* d_ws, int64_t ws_bytes, void* stream, unsigned mask_bits = 0u,
                              double* d_mask_partials = nullptr);

int obe_resample_particles(const double* d_old, int64_t ld_old, int32_t n_dims, int64_t n_particles,
                           const int64_t* d_idx, const double* d_normals, const double* h_factor,
                           const double* h_mean, double a_param, int32_t scale, double* d_new, int64_t ld_new,
                           double* d_weights, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_old) return bad_arg("obe_resample_particles: bad pointer/size");
    if (d_old == d_new) return bad_arg("obe_resample_particles: in-place gather is not supported");
    return resample_particles(d_old, ld_old, n_dims, n_particles, d_idx, d_normals, h_factor, h_mean, a_param, scale,
                              d_new, ld_new, d_weights, d_ws, ws_bytes, stream);
}

int obe_resample_particles_aos(const double* d_old_aos, int32_t n_dims, int64_t n_particles, const int64_t* d_idx,
                               const double* d_normals, const double* h_factor, const double* h_mean, double a_param,
                               int32_t scale, double* d_new, int64_t ld_new, double* d_weights, void* stream) {
    if (!d_old_aos || d_old_aos == d_new) return bad_arg("obe_resample_particles_aos: bad pointer");
    return resample_particles(nullptr, 0, n_dims, n_particles, d_idx, d_normals, h_factor, h_mean, a_param, scale, d_new,
                              ld_new, d_weights, const_cast<double*>(d_old_aos), 0, stream);
}

int obe_resample_particles_aos_masked(const double* d_old_aos, int32_t n_dims, int64_t n_particles,
                                      const int64_t* d_idx, const double* d_normals, const double* h_factor,
                                      const double* h_mean, double a_param, int32_t scale, double* d_new,
                                      int64_t ld_new, double* d_weights, const int32_t* h_rows, int32_t n_rows,
                                      double* d_mask_partials, void* stream) {
    if (!d_old_aos || d_old_aos == d_new || !h_rows || !d_mask_partials || n_rows < 1 || n_rows > kFastDims ||
        n_dims > kFastDims)
        return bad_arg("obe_resample_particles_aos_masked: bad pointer/size");
    unsigned bits = 0u;
    for (int k = 0; k < n_rows; ++k) {
        if (h_rows[k] < 0 || h_rows[k] >= n_dims) return bad_arg("obe_resample_particles_aos_masked: row index out of range");
        bits |= 1u << h_rows[k];
    }
    return resample_particles(nullptr, 0, n_dims, n_particles, d_idx, d_normals, h_factor, h_mean, a_param, scale, d_new,
                              ld_new, d_weights, const_cast<double*>(d_old_aos), 0, stream, bits, d_mask_partials);
}

static int resample_particles(const double* d_old, int64_t ld_old, int32_t n_dims, int64_t n_particles,
                              const int64_t* d_idx, const double* d_normals, const double* h_factor,
                              const double* h_mean, double a_param, int32_t scale, double* d_new, int64_t ld_new,
                              double* d_weights, void* d_ws, int64_t ws_bytes, void* stream, unsigned mask_bits,
                              double* d_mask_partials) {
    if (!d_idx || !d_normals || !h_factor || !h_mean || !d_new || !d_weights || n_particles <= 0)
        return bad_arg("obe_resample_particles: bad pointer/size");
    if (n_dims < 1 || n_dims > OBE_CLOUD_MAX_DIMS) return bad_arg("obe_resample_particles: n_dims must be 1..1024");
    if (n_dims > kFastDims) {
        // wide cloud: only the plain form (the (D, N) cloud, no mask); F and the mean travel through the workspace
        if (!d_old || mask_bits) return bad_arg("obe_resample_particles_aos: more than OBE_FAST_DIMS parameters (use obe_resample_particles)");
        const int64_t fm_doubles = (int64_t)n_dims * n_dims + n_dims;
        if (!d_ws || ws_bytes < fm_doubles * (int64_t)sizeof(double)) return bad_arg("obe_resample_particles: workspace too small");
        hipStream_t ws_st = as_stream(stream);
        double* fm = static_cast<double*>(d_ws);
        OBE_HIP_TRY(hipMemcpyAsync(fm, h_factor, sizeof(double) * n_dims * n_dims, hipMemcpyHostToDevice, ws_st));
        OBE_HIP_TRY(hipMemcpyAsync(fm + (int64_t)n_dims * n_dims, h_mean, sizeof(double) * n_dims, hipMemcpyHostToDevice, ws_st));
        resample_wide_kernel<<<stream_blocks(n_particles, kBlock), kBlock, 0, ws_st>>>(
            fm, n_dims, a_param, 1 - a_param, 1.0 / (double)n_particles, scale, d_old, ld_old, n_particles, d_idx, d_normals,
            d_new, ld_new, d_weights);
        OBE_CHECK_LAUNCH("resample_wide_kernel");
        return 0;
    }
    NudgeArgs na{};
    na.d = n_dims;
    na.scale = scale;
    na.a = a_param;
    na.one_minus_a = 1 - a_param;
    na.uniform_w = 1.0 / (double)n_particles;
    for (int i = 0; i < n_dims * n_dims; ++i) na.factor[i] = h_factor[i];
    for (int i = 0; i < n_dims; ++i) na.mean[i] = h_mean[i];
    hipStream_t st = as_stream(stream);
#define OBE_RS_CASE(DD) \
    case DD: return launch_resample<DD>(na, d_old, ld_old, n_particles, d_idx, d_normals, d_new, ld_new, d_weights, \
                                        d_ws, ws_bytes, st, mask_bits, d_mask_partials);
    switch (n_dims) {
        OBE_RS_CASE(1) OBE_RS_CASE(2) OBE_RS_CASE(3) OBE_RS_CASE(4) OBE_RS_CASE(5) OBE_RS_CASE(6) OBE_RS_CASE(7)
        OBE_RS_CASE(8) OBE_RS_CASE(9) OBE_RS_CASE(10) OBE_RS_CASE(11) OBE_RS_CASE(12) OBE_RS_CASE(13)
        OBE_RS_CASE(14) OBE_RS_CASE(15) OBE_RS_CASE(16)
    }
#undef OBE_RS_CASE
    return bad_arg("obe_resample_particles: n_dims must be 1..16");
}

// The random numbers of a resample, enqueued AHEAD of it (round 6).  The uniforms and normals of resample() depend on
// nothing but the caller's generator state and the cloud's shape (particlepdf.py:272, 296-301), and their chain —
// 92 us for 524 288 x 10 — is what the gather of a resample ends up waiting for.  Enqueued when pdf_update() starts,
// on the library's side stream of `stream`, it runs beside the update's own (latency-bound) kernels and the host round
// trips that follow; obe_resample_begin(h_pcg_state4 = NULL, the same buffers) then launches only the cloud's chains
// and waits for the end of this one.  The caller keeps the numbers for as long as its generator has not moved (a
// cycle that does not resample leaves them for the next one) and compares the state itself: nothing here knows
// whether they will be used.  h_i64[0..1] = {raw values consumed, normals found} are armed here.
// Refused (-1) before anything is launched without the library's side streams or page-locked h_i64.
int obe_resample_randoms_enqueue(const uint64_t* h_pcg_state4, int64_t n_particles, int32_t n_dims, int64_t n_raw,
                                 double* d_uniforms, const void* d_zig_tables, double* d_normals, void* d_zig_ws,
                                 int64_t zig_ws_bytes, int64_t* h_i64, void* stream) {
    if (!h_pcg_state4 || !d_uniforms || !d_zig_tables || !d_normals || !d_zig_ws || !h_i64 || n_particles <= 0 ||
        n_dims < 1 || n_dims > kFastDims)
        return bad_arg("obe_resample_randoms_enqueue: bad pointer/size");
    const int64_t n = n_particles, n_normal = n * n_dims;
    if (n_raw < n + n_normal + 4096) return bad_arg("obe_resample_randoms_enqueue: fewer raw values than the draws need");
    if (!device_view_of_host(h_i64)) return bad_arg("obe_resample_randoms_enqueue: h_i64 must be page-locked");
    hipStream_t st = as_stream(stream);
    SideStream side{};
    if (!side_stream_of(st, &side)) return bad_arg("obe_resample_randoms_enqueue: no side stream for this stream");
    auto ev = [](hipError_t e) { return e == hipSuccess ? 0 : fail(e, "obe_resample_randoms_enqueue: stream / event call"); };
    const int prev = obe_defer_host_sync(1);
    int rc = 0;
    do {
        // (the buffers may still be read by earlier work of the caller's stream: the previous resample's search and gather)
        if ((rc = ev(hipEventRecord(side.entry, st)))) break;
        if ((rc = ev(hipStreamWaitEvent(side.stream, side.entry, 0)))) break;
        if ((rc = obe_pcg64_uniforms_classify(h_pcg_state4, n, n_raw - n, d_uniforms, d_zig_tables, d_zig_ws, zig_ws_bytes,
                                              side.stream)))
            break;
        if ((rc = obe_ziggurat_finish(n_raw - n, n_normal, d_normals, h_i64, d_zig_ws, zig_ws_bytes, side.stream))) break;
        if ((rc = ev(hipEventRecord(side.pre, side.stream)))) break;
    } while (false);
    obe_defer_host_sync(prev);
    return rc;
}

// The device side of ParticlePDF.resample() up to the point where the host must factorise the covariance
// (particlepdf.py:260-301), enqueued by ONE call: the caller's PCG64 stream continued on the device (the N
// uniforms and the classification of the n_raw - N raw positions behind them, in one launch, straight from
// the generator state), the weight CDF (unless the caller's is fresh), the N-draw search, the covariance of
// the pre-resample cloud and the N x D ziggurat normals.  Nothing is waited for.  The host results land in
// page-locked memory, every word of which is ARMED here and watched by the caller
// (obe_host_words_wait: returns when none of the words carries the armed pattern any more):
//   h_f64[0]      sum(w) for numpy's validation of p   (1.0 is stored at once when the CDF was fresh)
//   h_f64[1 ..]   the K3 block; the kernels deliver the covariance (and, unless have_first_moments, the
//                 first moments): wait for h_f64 + 1 + lo .. h_f64 + 1 + obe_moments_len(D), lo = 2 + 4 D or 0
//   h_i64[0..1]   {raw values the normals consumed, normals found}
// The caller factorises, calls obe_resample_particles(), then waits for h_i64 and does its generator
// bookkeeping (obe_ziggurat_check).  Python issued these ~12 launches one library call at a time (5-15 us of
// interpreter between two launches: the GPU idled for most of a resample cycle at 262 144 particles).
int obe_resample_begin(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                       const double* d_weights, const uint64_t* h_pcg_state4, int32_t strict_cdf,
                       int32_t cdf_is_fresh, int32_t have_first_moments, int64_t n_raw, double* d_cdf,
                       double* d_uniforms, int64_t* d_idx, const void* d_zig_tables, double* d_normals,
                       void* d_zig_ws, int64_t zig_ws_bytes, double* d_moments, double* h_f64, int64_t* h_i64,
                       double* d_aos, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_particles || !d_weights || !d_cdf || !d_uniforms || !d_idx || !d_zig_tables ||
        !d_normals || !d_zig_ws || !d_moments || !h_f64 || !h_i64 || n_particles <= 0)
        return bad_arg("obe_resample_begin: bad pointer/size");
    // h_pcg_state4 == NULL: the uniforms and normals of THIS resample were enqueued ahead of it
    // (obe_resample_randoms_enqueue on this stream, into these d_uniforms / d_normals / d_zig_ws / h_i64): nothing of
    // the random chain is launched here, the caller's stream waits for the end of that chain instead
    const bool randoms_ahead = h_pcg_state4 == nullptr;
    if (n_dims < 1 || n_dims > kFastDims) return bad_arg("obe_resample_begin: n_dims must be 1..16 (OBE_FAST_DIMS)");
    const int64_t n = n_particles, n_normal = n * n_dims;
    if (n_raw < n + n_normal + 4096) return bad_arg("obe_resample_begin: fewer raw values than the draws need");
    if (!device_view_of_host(h_f64) || !device_view_of_host(h_i64))
        return bad_arg("obe_resample_begin: the host result buffers must be page-locked");
    hipStream_t st = as_stream(stream);
    const int prev = obe_defer_host_sync(1);
    int rc = 0;
    // Two chains that share nothing but the uniforms: the generator's (raw values -> uniforms, ziggurat normals
    // and their bookkeeping: touches d_uniforms, d_normals, d_zig_ws, h_i64) and the cloud's (CDF, search,
    // covariance: d_ws, d_cdf, d_idx, d_moments, h_f64).  They run side by side on two streams — the random
    // chain on a stream of the library's own — and the host factorises the covariance while the normals are
    // still being made.  The caller's stream waits for the uniforms before the search and for the normals at
    // the end, so whatever the caller enqueues next (the gather + nudge) is ordered behind both, and the stream
    // is not drained before the side stream's host words have been written (wait_host_words relies on that).
    // OBE_RESAMPLE_STREAMS=1: everything on the caller's stream, in the round-3 order (A/B measurements).
    // Only where the chains are long: joining two streams costs two cross-queue waits.  Same box, resample cycle
    // with one / two streams: 5000 particles x 3 (15 000 normals) 0.172 / 0.21 ms; 262 144 x 3 level (0.49 / 0.46,
    // 0.50 / 0.54); 524 288 x 10 (5.2 M normals) 1.518 / 1.494 ms.  OBE_RESAMPLE_STREAMS=2 forces it (tests).
    static const int streams_env = getenv("OBE_RESAMPLE_STREAMS") ? atoi(getenv("OBE_RESAMPLE_STREAMS")) : 0;
    SideStream side{};
    const bool split = (randoms_ahead || streams_env == 2 || (streams_env != 1 && n_normal >= ((int64_t)1 << 21))) &&
                       side_stream_of(st, &side);
    if (randoms_ahead && !split) return bad_arg("obe_resample_begin: randoms enqueued ahead need the library's side streams");
    void* rs = split ? static_cast<void*>(side.stream) : stream;
    const int64_t lo = have_first_moments ? 2 + 4 * (int64_t)n_dims : 0;
    auto ev = [](hipError_t e) { return e == hipSuccess ? 0 : fail(e, "obe_resample_begin: stream / event call"); };
    // d_aos (nullable; 8 n_dims N bytes): the (N, D) copy of the PRE-resample cloud that the gather of
    // obe_resample_particles_aos() reads, made here — it needs nothing but the old cloud, so it runs while the host
    // factorises the covariance instead of in front of the gather (16.5 us at 524 288 x 10).  With it, split mode
    // runs a THIRD chain on a second stream of the library's: the covariance (its partial sums in d_aos, which is
    // written only afterwards — the caller's workspace belongs to the scan and the search, which now run beside
    // it) and the copy.  The host then has the covariance ~20 us earlier, and the caller's stream is scan ->
    // guide -> search only: the random chain (~85 us at that size) is what the gather waits for.
    const int64_t nv_max = std::max<int64_t>(2 + 2 * n_dims, (int64_t)n_dims * (n_dims + 1) / 2);
    const bool third = split && d_aos && n_dims >= 2 && side.stream2 &&
                       (int64_t)n_dims * n >= (int64_t)kMomGridCap * nv_max + nv_max;
    const bool copy_here = d_aos && n_dims >= 2;
    auto make_aos = [&](hipStream_t on) -> int {
#define OBE_AOS_CASE(DD) \
    case DD: return launch_soa_to_aos<DD>(d_particles, ld_p, n, d_aos, on);
        switch (n_dims) {
            OBE_AOS_CASE(2) OBE_AOS_CASE(3) OBE_AOS_CASE(4) OBE_AOS_CASE(5) OBE_AOS_CASE(6) OBE_AOS_CASE(7) OBE_AOS_CASE(8)
            OBE_AOS_CASE(9) OBE_AOS_CASE(10) OBE_AOS_CASE(11) OBE_AOS_CASE(12) OBE_AOS_CASE(13) OBE_AOS_CASE(14)
            OBE_AOS_CASE(15) OBE_AOS_CASE(16)
        }
#undef OBE_AOS_CASE
        return 0;
    };
    do {
        if (third) {
            // Three chains.  The host enqueues the LONGEST first — the random numbers (uniforms + classification,
            // then four ziggurat kernels: ~90 us at 524 288 x 10, what the gather ends up waiting for) —, then the
            // covariance + (N, D) copy (the host's factorisation waits for the covariance), then CDF -> guide ->
            // search on the caller's stream: every launch costs the host 2-4 us, and with the cloud's chains first
            // the random chain used to start 20-45 us after the others.
            if ((rc = ev(hipEventRecord(side.entry, st)))) break;
            if (!randoms_ahead) {
                if ((rc = ev(hipStreamWaitEvent(side.stream, side.entry, 0)))) break;
                if ((rc = obe_pcg64_uniforms_classify(h_pcg_state4, n, n_raw - n, d_uniforms, d_zig_tables, d_zig_ws,
                                                      zig_ws_bytes, rs)))
                    break;
                if ((rc = ev(hipEventRecord(side.mid, side.stream)))) break;
                if ((rc = obe_ziggurat_finish(n_raw - n, n_normal, d_normals, h_i64, d_zig_ws, zig_ws_bytes, rs))) break;
                if ((rc = ev(hipEventRecord(side.done, side.stream)))) break;
            }
            if ((rc = ev(hipStreamWaitEvent(side.stream2, side.entry, 0)))) break;
            arm_host_words(h_f64 + 1 + lo, obe_moments_len(n_dims) - lo);
            bool host_written = false;
            if ((rc = moments_call(d_particles, ld_p, n_dims, n, d_weights, have_first_moments ? 2 : 1, d_moments,
                                   h_f64 + 1, d_aos, (int64_t)sizeof(double) * n_dims * n, side.stream2, &host_written)))
                break;
            if ((rc = make_aos(side.stream2))) break;
            if ((rc = ev(hipEventRecord(side.done2, side.stream2)))) break;
            if (!cdf_is_fresh) {
                arm_host_word(h_f64);
                if ((rc = obe_weight_cdf(d_weights, n, strict_cdf, d_cdf, h_f64, d_ws, ws_bytes, stream))) break;
            } else {
                h_f64[0] = 1.0;
            }
            if ((rc = ev(hipStreamWaitEvent(st, randoms_ahead ? side.pre : side.mid, 0)))) break;
            if ((rc = obe_cdf_search(d_cdf, n, d_uniforms, n, d_idx, d_ws, ws_bytes, stream))) break;
            if (!randoms_ahead && (rc = ev(hipStreamWaitEvent(st, side.done, 0)))) break;
            if ((rc = ev(hipStreamWaitEvent(st, side.done2, 0)))) break;
            break;
        }
        // (split: the cloud's chain is enqueued first — the host waits for the covariance, and every launch costs
        // it a few microseconds; the kernels of the random chain arrive while the scan is already running)
        if (split && (rc = ev(hipEventRecord(side.entry, st)))) break;
        auto cdf_and_covariance = [&]() -> int {
            if (!cdf_is_fresh) {
                arm_host_word(h_f64);
                if (int e = obe_weight_cdf(d_weights, n, strict_cdf, d_cdf, h_f64, d_ws, ws_bytes, stream)) return e;
            } else {
                h_f64[0] = 1.0;
            }
            if (!split) return 0;       // (one stream: the round-3 order, covariance after the search)
            arm_host_words(h_f64 + 1 + lo, obe_moments_len(n_dims) - lo);
            bool host_written = false;
            return moments_call(d_particles, ld_p, n_dims, n, d_weights, have_first_moments ? 2 : 1, d_moments, h_f64 + 1,
                                d_ws, ws_bytes, st, &host_written);
        };
        if (split && (rc = cdf_and_covariance())) break;
        // the buffers of the random chain may still be read by earlier work of the caller's stream
        if (randoms_ahead) {             // (always split: the chain that ran ahead ends in side.pre)
            if ((rc = ev(hipStreamWaitEvent(st, side.pre, 0)))) break;
            if ((rc = obe_cdf_search(d_cdf, n, d_uniforms, n, d_idx, d_ws, ws_bytes, stream))) break;
            if (copy_here && (rc = make_aos(st))) break;
            break;
        }
        if (split && (rc = ev(hipStreamWaitEvent(side.stream, side.entry, 0)))) break;
        // uniforms + the classification of every raw position behind them, straight from the generator state
        if ((rc = obe_pcg64_uniforms_classify(h_pcg_state4, n, n_raw - n, d_uniforms, d_zig_tables, d_zig_ws,
                                              zig_ws_bytes, rs)))
            break;
        if (split) {
            if ((rc = ev(hipEventRecord(side.mid, side.stream)))) break;
            if ((rc = obe_ziggurat_finish(n_raw - n, n_normal, d_normals, h_i64, d_zig_ws, zig_ws_bytes, rs))) break;
            if ((rc = ev(hipEventRecord(side.done, side.stream)))) break;
            if ((rc = ev(hipStreamWaitEvent(st, side.mid, 0)))) break;
        } else if ((rc = cdf_and_covariance())) {
            break;
        }
        if ((rc = obe_cdf_search(d_cdf, n, d_uniforms, n, d_idx, d_ws, ws_bytes, stream))) break;
        if (copy_here && (rc = make_aos(st))) break;
        if (split) {
            if ((rc = ev(hipStreamWaitEvent(st, side.done, 0)))) break;
        } else {
            arm_host_words(h_f64 + 1 + lo, obe_moments_len(n_dims) - lo);
            bool host_written = false;
            if ((rc = moments_call(d_particles, ld_p, n_dims, n, d_weights, have_first_moments ? 2 : 1, d_moments,
                                   h_f64 + 1, d_ws, ws_bytes, st, &host_written)))
                break;
            if ((rc = obe_ziggurat_finish(n_raw - n, n_normal, d_normals, h_i64, d_zig_ws, zig_ws_bytes, stream))) break;
        }
    } while (false);
    obe_defer_host_sync(prev);
    return rc;
}

}  // extern "C"
