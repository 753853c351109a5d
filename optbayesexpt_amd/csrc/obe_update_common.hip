// K2, model-independent half — see obe_update.h.  Pass B (normalisation) alone and fused with K3's first moments, the
// folds, the np.sum-ordered sums of small clouds, the y / likelihood forms of the update, the constraint masks of
// OptBayesExptNoiseParameter, good_setting()'s power normalisation.  All HBM-bound streams with fixed-order
// reductions (no float atomics): run-to-run identical.
#include "obe_update.h"

namespace obe {

// pass A, model output supplied (C, N) with ld_y between channels
__global__ __launch_bounds__(kBlock) void update_y_kernel(
    LikArgs la, int n_channels, const double* __restrict__ yv, int64_t ld_y,
    const double* __restrict__ particles, int64_t ld, int64_t n, double* __restrict__ weights,
    double* __restrict__ partials) {
    __shared__ double red[kBlock / kWave];
    double acc = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        double y[OBE_MAX_CHANNELS];
        for (int c = 0; c < n_channels; ++c) y[c] = yv[(int64_t)c * ld_y + p];
        const double t = nan_to_num(weights[p] * likelihood_of(y, la, particles, ld, p));
        weights[p] = t;
        acc += t;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// pass A, likelihood supplied
__global__ __launch_bounds__(kBlock) void update_lik_kernel(const double* __restrict__ lik, int64_t n,
                                                            double* __restrict__ weights,
                                                            double* __restrict__ partials) {
    __shared__ double red[kBlock / kWave];
    double acc = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const double t = nan_to_num(weights[p] * lik[p]);
        weights[p] = t;
        acc += t;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// pass B: normalise by the (deterministically re-folded) total; partial sums of w'^2
__global__ __launch_bounds__(kBlock) void normalize_kernel(const double* __restrict__ partials_in,
                                                           int n_partials, int64_t n,
                                                           double* __restrict__ weights,
                                                           double* __restrict__ partials_out,
                                                           const double* __restrict__ stop) {
    __shared__ double red[kBlock / kWave];
    if (stop && stop[0] != 0.0) return;
    const double total = block_sum_array(partials_in, n_partials, red);
    double acc = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const double w = nan_to_num(weights[p] / total);
        weights[p] = w;
        acc += nan_to_num(w * w);
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials_out[blockIdx.x] = s;
}

template <int D, bool FOLD>
__global__ __launch_bounds__(kMomThreads) void normalize_moments_kernel(const double* __restrict__ partials_in,
                                                                   int n_partials, const double* __restrict__ x,
                                                                   int64_t ld, int64_t n, double* __restrict__ weights,
                                                                   double* partials_w2, double* partials_mom,
                                                                   UpdateFold fold) {
    __shared__ double red[kMomThreads / kWave];
    const double total = block_sum_array(partials_in, n_partials, red, kBlock);      // (normalize_kernel's total, bit for bit)
    // FOLD: sum nan_to_num(w'^2) travels as one more column of the moment rows (one round of loads in the fold)
    constexpr int NV = 2 + 2 * D + (FOLD ? 1 : 0);
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
    double acc = 0.0;
    // U particles per trip: the loads of all of them are issued before the first is used (a wave of the one-per-CU
    // grid otherwise has D + 1 loads in flight and ~50 dependent FP64 instructions between two round trips to
    // HBM); they are accumulated in the order p, p + stride, ... of the one-at-a-time loop: the same bits.
    constexpr int U = OBE_NORM_UNROLL;
    const int64_t stride = (int64_t)gridDim.x * kMomThreads;
    for (int64_t p = (int64_t)blockIdx.x * kMomThreads + threadIdx.x; p < n; p += U * stride) {
        double xi[U][D], t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = p + u * stride;
            const int64_t qq = q < n ? q : p;
#pragma unroll
            for (int i = 0; i < D; ++i) xi[u][i] = x[(int64_t)i * ld + qq];
            t[u] = weights[qq];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = p + u * stride;
            if (q < n) {
                const double w = nan_to_num(t[u] / total);
                weights[q] = w;
                acc += nan_to_num(w * w);
                accumulate_first_moments<D>(reinterpret_cast<double(&)[2 + 2 * D]>(v), w, xi[u]);
            }
        }
    }
    if constexpr (!FOLD) {
        const double s = block_sum(acc, red);
        if (threadIdx.x == 0) partials_w2[blockIdx.x] = s;
        store_block_partials<NV, false, kMomThreads>(v, partials_mom);
    } else {
        v[NV - 1] = acc;
        publish_and_fold_update<D, kMomThreads>(reinterpret_cast<double(&)[3 + 2 * D]>(v), total, partials_mom, fold);
    }
}

// ... and its fold: {sum t, sum w'^2} + the K3 block (mean, m1, m2, std), to the device copies and,
// when the caller's h_out is page-locked, straight to the host: [0] sum t, [1] sum w'^2, [2..) K3 block
__global__ __launch_bounds__(kFoldThreads) void fold_update_moments_kernel(
    const double* __restrict__ pa, int n_pa, const double* __restrict__ pb, int n_pb,
    const double* __restrict__ partials_mom, int d, double* __restrict__ scalars, double* __restrict__ mom_out,
    double* __restrict__ host_out) {
    __shared__ double raw[kMaxMomentValues];
    __shared__ double red[kFoldThreads / kWave];
    fold_values_block(partials_mom, n_pb, 2 + 2 * d, raw);
    derive_first_moments(raw, d, mom_out, host_out ? host_out + 2 : nullptr);
    if (host_out && (int)threadIdx.x < d) host_results_before_flag();      // (the threads that stored a moment to the host;
                                                                             // a fence by all 1024 costs 5 us)
    const double a = block_sum_array(pa, n_pa, red);
    __syncthreads();
    const double b = block_sum_array(pb, n_pb, red);     // (its barriers order the moment stores before thread 0)
    if (threadIdx.x == 0) {
        scalars[0] = a;
        scalars[1] = b;
        if (host_out) {
            host_out[0] = a;
            host_results_before_flag();
            host_out[1] = b;                             // the word the host watches (wait_host_word)
        }
    }
}

// pass C: scalars[0] = sum t, scalars[1] = sum w'^2
// (host_out: the device view of the caller's page-locked h_out, or NULL)
__global__ __launch_bounds__(kBlock) void fold2_kernel(const double* __restrict__ pa, const double* __restrict__ pb,
                                                       int n_partials, double* __restrict__ scalars,
                                                       double* __restrict__ host_out) {
    __shared__ double red[kBlock / kWave];
    const double a = block_sum_array(pa, n_partials, red);
    __syncthreads();
    const double b = block_sum_array(pb, n_partials, red);
    if (threadIdx.x == 0) {
        scalars[0] = a;
        scalars[1] = b;
        if (host_out) {
            host_out[0] = a;
            host_results_before_flag();
            host_out[1] = b;                             // the word the host watches (wait_host_word)
        }
    }
}

// fold + delivery of {sum t, sum w'^2} to h_out: written by the kernel itself when h_out is page-locked
static int fold2_to_host(const double* pa, const double* pb, int nb, double* scalars, double* h_out, hipStream_t st) {
    double* hv = static_cast<double*>(device_view_of_host(h_out));
    if (hv) arm_host_words(h_out, 2);
    fold2_kernel<<<1, kBlock, 0, st>>>(pa, pb, nb, scalars, hv);
    OBE_CHECK_LAUNCH("fold2_kernel");
    if (h_out) {
        if (hv) return wait_host_words(h_out, 2, st);
        OBE_HIP_TRY(hipMemcpyAsync(h_out, scalars, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        OBE_HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

// last launch of a sweep batch: fold the final point's partials and test it (unless an earlier
// point already stopped the batch).
__global__ __launch_bounds__(kBlock) void fold2_stop_kernel(const double* __restrict__ pa,
                                                            const double* __restrict__ pb, int n_partials,
                                                            double* __restrict__ scalars, double n_particles,
                                                            int auto_resample, double resample_threshold,
                                                            int n_points) {
    __shared__ double red[kBlock / kWave];
    if (scalars[2] != 0.0) return;
    const double a = block_sum_array(pa, n_partials, red);
    __syncthreads();
    const double b = block_sum_array(pb, n_partials, red);
    if (threadIdx.x == 0) {
        scalars[0] = a;
        scalars[1] = b;
        scalars[3] = (double)n_points;
        if (auto_resample && resample_due(b, n_particles, resample_threshold)) scalars[2] = 1.0;
    }
}

__global__ void sweep_state_reset_kernel(double* __restrict__ scalars) {
    scalars[2] = 0.0;
    scalars[3] = 0.0;
}

__global__ __launch_bounds__(kBlock) void weight_sums_kernel(const double* __restrict__ weights, int64_t n,
                                                             double* __restrict__ pa, double* __restrict__ pb) {
    __shared__ double red[kBlock / kWave];
    double a = 0.0, b = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const double w = weights[p];
        a += nan_to_num(w * w);
        b += w;
    }
    const double sa = block_sum(a, red);
    __syncthreads();
    const double sb = block_sum(b, red);
    if (threadIdx.x == 0) {
        pa[blockIdx.x] = sa;
        pb[blockIdx.x] = sb;
    }
}

// (accumulate: this launch handles one group of <= OBE_MAX_CHANNELS channels of a wider record and multiplies its
// product onto the groups before it — lky *= ... in channel order, obe_base.py:453-456)
__global__ __launch_bounds__(kBlock) void likelihood_y_kernel(LikArgs la, int n_channels,
                                                              const double* __restrict__ yv, int64_t ld_y,
                                                              const double* __restrict__ particles, int64_t ld,
                                                              int64_t n, double* __restrict__ out, int accumulate = 0) {
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        double y[OBE_MAX_CHANNELS];
        for (int c = 0; c < n_channels; ++c) y[c] = yv[(int64_t)c * ld_y + p];
        const double lk = likelihood_of(y, la, particles, ld, p);
        out[p] = accumulate ? out[p] * lk : lk;
    }
}

// np.power(lky, choke) over the finished product of a record wider than one group (obe_base.py:458-459)
__global__ __launch_bounds__(kBlock) void choke_kernel(double* __restrict__ lk, int64_t n, double choke) {
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock)
        lk[p] = pow(lk[p], choke);
}

// ---- OptBayesExptNoiseParameter.enforce_parameter_constraints (obe_noiseparam.py:57-79)
struct RowsArg {
    int n;
    int rows[OBE_MAX_DIMS];
};

__global__ __launch_bounds__(kBlock) void mask_kernel(RowsArg ra, const double* __restrict__ particles, int64_t ld,
                                                      int64_t n, double* __restrict__ weights,
                                                      double* __restrict__ psum, double* __restrict__ pcount) {
    __shared__ double red[kBlock / kWave];
    double acc = 0.0, cnt = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        bool bad = false;
        for (int k = 0; k < ra.n; ++k) bad = bad || (particles[(int64_t)ra.rows[k] * ld + p] <= 0.0);
        double w = weights[p];
        if (bad) {
            w = 0.0;
            weights[p] = 0.0;
            cnt += 1.0;
        }
        acc += w;
    }
    const double s = block_sum(acc, red);
    __syncthreads();
    const double c = block_sum(cnt, red);
    if (threadIdx.x == 0) {
        psum[blockIdx.x] = s;
        pcount[blockIdx.x] = c;
    }
}

// renormalise only if anything was zeroed (scalars[1] = count)
// (host_changed: the device view of the caller's page-locked count, or NULL)
__global__ __launch_bounds__(kBlock) void mask_renorm_kernel(const double* __restrict__ scalars, int64_t n,
                                                             double* __restrict__ weights,
                                                             int64_t* __restrict__ host_changed) {
    if (host_changed && blockIdx.x == 0 && threadIdx.x == 0) *host_changed = (int64_t)scalars[1];
    if (scalars[1] == 0.0) return;
    const double total = scalars[0];
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock)
        weights[p] = weights[p] / total;
}

// The constraint mask's second half and the first moments of the constrained cloud in ONE pass (round 4):
// renormalise by the folded total if anything was zeroed — mask_renorm_kernel's arithmetic — and accumulate
// sum w, sum w^2, sum w x, sum w x^2 with moments_pass1's grid and order, folded by the last workgroup to
// arrive: what obe_mask_nonpositive() + obe_moments() leave behind, bit for bit, in two launches instead of
// five and without a host round trip in between (the count arrives with the moments).
struct MaskFold {
    unsigned* counter;
    double* mom_out;          // K3 block on the device
    double* host_mom;         // device view of the caller's page-locked copy, or NULL
    int64_t* host_changed;    // device view of the caller's page-locked count (stored last), or NULL
};

template <int D>
__global__ __launch_bounds__(kMomThreads) void mask_renorm_moments_kernel(const double* __restrict__ psum,
                                                                     const double* __restrict__ pcount, int nb_in,
                                                                     const double* __restrict__ x, int64_t ld,
                                                                     int64_t n, double* __restrict__ weights,
                                                                     double* partials_mom, MaskFold mf) {
    __shared__ double red[kMomThreads / kWave];
    const double total = block_sum_array(psum, nb_in, red, kBlock);
    __syncthreads();
    const double count = block_sum_array(pcount, nb_in, red, kBlock);
    const bool renorm = count != 0.0;
    double v[2 + 2 * D];
#pragma unroll
    for (int k = 0; k < 2 + 2 * D; ++k) v[k] = 0.0;
    constexpr int U = OBE_NORM_UNROLL;       // (as in normalize_moments_kernel)
    const int64_t stride = (int64_t)gridDim.x * kMomThreads;
    for (int64_t p = (int64_t)blockIdx.x * kMomThreads + threadIdx.x; p < n; p += U * stride) {
        double xi[U][D], t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = p + u * stride;
            const int64_t qq = q < n ? q : p;
#pragma unroll
            for (int i = 0; i < D; ++i) xi[u][i] = x[(int64_t)i * ld + qq];
            t[u] = weights[qq];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = p + u * stride;
            if (q < n) {
                double w = t[u];
                if (renorm) {
                    w = w / total;
                    weights[q] = w;
                }
                accumulate_first_moments<D>(v, w, xi[u]);
            }
        }
    }
    store_block_partials<2 + 2 * D, true, kMomThreads>(v, partials_mom);
    __shared__ int last;
    if (!arrive_last<false>(mf.counter, &last)) return;
    __shared__ double raw[kMaxMomentValues];
    fold_values_block<kMomThreads, true, (2 + 2 * D + kMomThreads / kWave - 1) / (kMomThreads / kWave)>(partials_mom, gridDim.x, 2 + 2 * D, raw);
    if (threadIdx.x < kWave) {
        derive_first_moments(raw, D, mf.mom_out, mf.host_mom);
        if (mf.host_changed) {
            host_results_before_flag();
            if (threadIdx.x == 0) *mf.host_changed = (int64_t)count;
        }
    }
}

// good_setting(): p = nan_to_num(u ** pickiness)  (obe_base.py:781-783)
__global__ __launch_bounds__(kBlock) void power_kernel(const double* __restrict__ u, int64_t n, double k,
                                                       double* __restrict__ p, double* __restrict__ partials) {
    __shared__ double red[kBlock / kWave];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double v = nan_to_num(pow(u[i], k));
        p[i] = v;
        acc += v;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// p /= sum(p)   (obe_base.py:784)
__global__ __launch_bounds__(kBlock) void divide_by_total_kernel(const double* __restrict__ partials, int n_partials,
                                                                 int64_t n, double* __restrict__ p) {
    __shared__ double red[kBlock / kWave];
    const double total = block_sum_array(partials, n_partials, red);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        p[i] = p[i] / total;
}

// np.average(sigma**2, weights=w) per channel from the K3 block: m2[row] / sum w
__global__ void noise_var_kernel(const double* __restrict__ moments, int d, RowsArg ra, double* __restrict__ out) {
    const int c = threadIdx.x;
    if (c < ra.n) out[c] = moments[2 + 2 * d + ra.rows[c]] / moments[0];
}

int fill_lik_args(LikArgs& la, const double* h_y_meas, const double* h_sigma,
                         const int32_t* h_noise_rows, int32_t n_lik_channels, double choke, int n_rows) {
    if (n_lik_channels < 0 || n_lik_channels > OBE_MAX_CHANNELS) return bad_arg("n_lik_channels out of range");
    if (!h_y_meas) return bad_arg("h_y_meas is NULL");
    if (!h_sigma && !h_noise_rows) return bad_arg("need h_sigma or h_noise_rows");
    la.n_ch = n_lik_channels;
    la.use_rows = h_noise_rows != nullptr;
    la.use_choke = !(choke != choke);
    la.choke = la.use_choke ? choke : 1.0;
    for (int c = 0; c < OBE_MAX_CHANNELS; ++c) {
        la.y_meas[c] = c < n_lik_channels ? h_y_meas[c] : 0.0;
        la.sigma[c] = (c < n_lik_channels && h_sigma) ? h_sigma[c] : 1.0;
        la.noise_rows[c] = (c < n_lik_channels && h_noise_rows) ? h_noise_rows[c] : 0;
        if (la.use_rows && c < n_lik_channels && (la.noise_rows[c] < 0 || la.noise_rows[c] >= n_rows))
            return bad_arg("noise row index out of range");
    }
    return 0;
}

int64_t update_ws_bytes(int moments_dims) {
    const int64_t mom = moments_dims > 0 ? (int64_t)kMomGridCap * (3 + 2 * moments_dims) : 0;   // (+1: sum w'^2 rides along)
    return (2 * (int64_t)kMaxBlocks + 8 + mom) * sizeof(double);
}
int carve_update_ws(void* d_ws, int64_t ws_bytes, UpdateWs& w, int moments_dims) {
    const int64_t need = update_ws_bytes(moments_dims);
    if (!d_ws || ws_bytes < need) return bad_arg("workspace too small");
    w.pa = static_cast<double*>(d_ws);
    w.pb = w.pa + kMaxBlocks;
    w.scalars = w.pb + kMaxBlocks;
    w.mom = w.scalars + 8;
    return 0;
}

// Grid of the update passes: 768 workgroups (3 per CU) measured best at 1M particles
// (18.3 us per update vs 21.3 us at 2048: fewer partials for pass B / C to fold).
int update_blocks(int64_t n) {
    static const int forced = getenv("OBE_UPDATE_BLOCKS") ? atoi(getenv("OBE_UPDATE_BLOCKS")) : 0;   // tuning aid
    const int cap = forced > 0 ? forced : 768;
    const int nb = stream_blocks(n, kBlock);
    return nb > cap ? cap : nb;
}

// ---- strict sums (obe_strict_sums; tuning_parameters['strict_sums']): np.sum's ORDER of additions ----------------
// particlepdf.py:138 divides by np.sum(tmp) and :243 tests 1 / np.sum(wsquared).  NumPy adds a contiguous float64
// vector in a fixed order (numpy/_core/src/umath/loops_utils.h.src, pairwise_sum; restated and pinned against
// np.sum itself in oracle/obe_oracle.py: numpy_pairwise_sum): the vector in pieces of 8192 elements, res = 0.0,
// res += pairwise(piece); pairwise(n < 8) = the elements one after the other from 0.0; pairwise(n <= 128) = eight
// interleaved running sums combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the last n % 8 elements added one by
// one; longer runs split at n/2 rounded down to a multiple of 8 and the halves are added.  One workgroup does
// exactly that: a (sub)tree of at most 4104 elements is staged in LDS, its leaves (runs of <= 128) are summed by one
// thread each, thread 0 adds them up the tree.  The reference's own tests compare updated weights with
// assert_array_equal (tests/test_optbayesexpt.py:58-69) — with these sums the device holds that, bit for bit.
constexpr int kNpPiece = 8192, kNpLeaf = 128, kNpStage = 4104;

template <bool SQUARE>
__device__ __forceinline__ double np_element(double x) { return SQUARE ? nan_to_num(x * x) : x; }

// pairwise_sum of a run of <= 128 staged elements
__device__ __forceinline__ double np_leaf_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + a[i];
    return res;
}

struct NpRun {
    int lo, n;
};

// the leaves of pairwise_sum's recursion over n elements, left to right: f(leaf index, first element, length)
template <class F>
__device__ __forceinline__ int np_for_each_leaf(int n, F f) {
    NpRun stack[20];
    int sp = 0, idx = 0;
    stack[sp++] = NpRun{0, n};
    while (sp > 0) {
        const NpRun nd = stack[--sp];
        if (nd.n <= kNpLeaf) {
            f(idx++, nd.lo, nd.n);
            continue;
        }
        int n2 = nd.n / 2;
        n2 -= n2 % 8;
        stack[sp++] = NpRun{nd.lo + n2, nd.n - n2};      // (popped second: the right half)
        stack[sp++] = NpRun{nd.lo, n2};
    }
    return idx;
}

// the leaf sums added up the recursion tree, left + right at every node (thread 0; leaves in left-to-right order)
__device__ __forceinline__ double np_combine(const double* leaf, int n) {
    // post-order without recursion: a node is expanded once (its halves pushed), then found again with its halves'
    // sums on the value stack
    NpRun stack[20];
    bool expanded[20];
    double vals[20];
    int sp = 0, vp = 0, idx = 0;
    stack[sp] = NpRun{0, n};
    expanded[sp++] = false;
    while (sp > 0) {
        const NpRun nd = stack[sp - 1];
        if (nd.n <= kNpLeaf) {
            vals[vp++] = leaf[idx++];
            --sp;
        } else if (!expanded[sp - 1]) {
            expanded[sp - 1] = true;
            int n2 = nd.n / 2;
            n2 -= n2 % 8;
            stack[sp] = NpRun{nd.lo + n2, nd.n - n2};
            expanded[sp++] = false;
            stack[sp] = NpRun{nd.lo, n2};
            expanded[sp++] = false;
        } else {
            const double right = vals[--vp], left = vals[--vp];
            vals[vp++] = left + right;
            --sp;
        }
    }
    return vals[0];
}

// pairwise_sum of a run of <= kNpStage elements of a[] (global), element transform applied while staging; valid in thread 0
template <bool SQUARE>
__device__ __forceinline__ double np_staged_sum(const double* __restrict__ a, int n, double* stage, double* leaf) {
    __syncthreads();                                     // (the previous run's stage and leaves have been consumed)
    for (int i = threadIdx.x; i < n; i += blockDim.x) stage[i] = np_element<SQUARE>(a[i]);
    __syncthreads();
    np_for_each_leaf(n, [&](int idx, int lo, int len) {
        if (idx % (int)blockDim.x == (int)threadIdx.x) leaf[idx] = np_leaf_sum(stage + lo, len);
    });
    __syncthreads();
    return threadIdx.x == 0 ? np_combine(leaf, n) : 0.0;
}

// (stop: a sweep batch whose earlier point asked for a resample — every later launch of the batch returns at once)
template <bool SQUARE>
__global__ __launch_bounds__(kBlock) void numpy_order_sum_kernel(const double* __restrict__ a, int64_t n,
                                                                 double* __restrict__ out,
                                                                 const double* __restrict__ stop = nullptr) {
    __shared__ double stage[kNpStage];
    __shared__ double leaf[kNpStage / 64 + 8];           // (a leaf holds at least 64 elements unless it is the whole run)
    if (stop && stop[0] != 0.0) return;
    double res = 0.0;
    for (int64_t lo = 0; lo < n; lo += kNpPiece) {
        const int m = (int)(n - lo < kNpPiece ? n - lo : kNpPiece);
        double piece;
        if (m <= kNpStage) {
            piece = np_staged_sum<SQUARE>(a + lo, m, stage, leaf);
        } else {                                         // the first split of pairwise_sum, its halves staged in turn
            int n2 = m / 2;
            n2 -= n2 % 8;
            const double left = np_staged_sum<SQUARE>(a + lo, n2, stage, leaf);
            const double right = np_staged_sum<SQUARE>(a + lo + n2, m - n2, stage, leaf);
            piece = left + right;
        }
        res = res + piece;
    }
    if (threadIdx.x == 0) out[0] = res;
}

// whether the calling thread's unfused updates sum in np.sum's order (obe_strict_sums)
static thread_local int g_strict_sums = 0;
bool strict_sums_on() { return g_strict_sums != 0; }

int finish_update(const UpdateWs& w, int nb, int64_t n, double* d_weights, double* h_out, hipStream_t st) {
    if (g_strict_sums) {
        // sum t in np.sum's order -> pa[0]; the normalisation divides by exactly that; sum nan_to_num(w'^2) likewise
        numpy_order_sum_kernel<false><<<1, kBlock, 0, st>>>(d_weights, n, w.pa);
        OBE_CHECK_LAUNCH("numpy_order_sum_kernel");
        normalize_kernel<<<nb, kBlock, 0, st>>>(w.pa, 1, n, d_weights, w.pb, nullptr);
        OBE_CHECK_LAUNCH("normalize_kernel");
        numpy_order_sum_kernel<true><<<1, kBlock, 0, st>>>(d_weights, n, w.pb);
        OBE_CHECK_LAUNCH("numpy_order_sum_kernel");
        return fold2_to_host(w.pa, w.pb, 1, w.scalars, h_out, st);
    }
    normalize_kernel<<<nb, kBlock, 0, st>>>(w.pa, nb, n, d_weights, w.pb, nullptr);
    OBE_CHECK_LAUNCH("normalize_kernel");
    return fold2_to_host(w.pa, w.pb, nb, w.scalars, h_out, st);
}


int launch_normalize_moments(int d, const UpdateWs& w, int nb, int nm, const double* d_particles, int64_t ld_p,
                             int64_t n_particles, double* d_weights, const UpdateFold& fold, double* d_moments, double* hv,
                             hipStream_t st) {
    const bool counter = fold.counter != nullptr;
#define OBE_UPD_MOM_CASE(DD)                                                                                           \
    case DD:                                                                                                           \
        if (counter)                                                                                                   \
            normalize_moments_kernel<DD, true><<<nm, kMomThreads, 0, st>>>(w.pa, nb, d_particles, ld_p, n_particles,       \
                                                                      d_weights, w.pb, w.mom, fold);                   \
        else                                                                                                           \
            normalize_moments_kernel<DD, false><<<nm, kMomThreads, 0, st>>>(w.pa, nb, d_particles, ld_p, n_particles,      \
                                                                       d_weights, w.pb, w.mom, fold);                  \
        break;
    switch (d) {
        OBE_UPD_MOM_CASE(1) OBE_UPD_MOM_CASE(2) OBE_UPD_MOM_CASE(3) OBE_UPD_MOM_CASE(4) OBE_UPD_MOM_CASE(5)
        OBE_UPD_MOM_CASE(6) OBE_UPD_MOM_CASE(7) OBE_UPD_MOM_CASE(8) OBE_UPD_MOM_CASE(9) OBE_UPD_MOM_CASE(10)
        OBE_UPD_MOM_CASE(11) OBE_UPD_MOM_CASE(12) OBE_UPD_MOM_CASE(13) OBE_UPD_MOM_CASE(14) OBE_UPD_MOM_CASE(15)
        OBE_UPD_MOM_CASE(16)
    }
#undef OBE_UPD_MOM_CASE
    OBE_CHECK_LAUNCH("normalize_moments_kernel");
    if (!counter) {
        fold_update_moments_kernel<<<1, kFoldThreads, 0, st>>>(w.pa, nb, w.pb, nm, w.mom, d, w.scalars, d_moments, hv);
        OBE_CHECK_LAUNCH("fold_update_moments_kernel");
    }
    return 0;
}

int launch_sweep_reset(const UpdateWs& w, hipStream_t st) {
    sweep_state_reset_kernel<<<1, 1, 0, st>>>(w.scalars);
    OBE_CHECK_LAUNCH("sweep_state_reset_kernel");
    return 0;
}

int launch_sweep_point_tail(const UpdateWs& w, int nb, int nfold, int64_t n_particles, double* d_weights, bool strict,
                            hipStream_t st) {
    const double* stop = w.scalars + 2;
    if (strict) {
        numpy_order_sum_kernel<false><<<1, kBlock, 0, st>>>(d_weights, n_particles, w.pa, stop);
        OBE_CHECK_LAUNCH("numpy_order_sum_kernel");
    }
    normalize_kernel<<<nb, kBlock, 0, st>>>(w.pa, nfold, n_particles, d_weights, w.pb, stop);
    OBE_CHECK_LAUNCH("normalize_kernel");
    if (strict) {
        numpy_order_sum_kernel<true><<<1, kBlock, 0, st>>>(d_weights, n_particles, w.pb, stop);
        OBE_CHECK_LAUNCH("numpy_order_sum_kernel");
    }
    return 0;
}

int launch_sweep_end(const UpdateWs& w, int nfold, int64_t n_particles, int auto_resample, double resample_threshold,
                     int n_points, hipStream_t st) {
    fold2_stop_kernel<<<1, kBlock, 0, st>>>(w.pa, w.pb, nfold, w.scalars, (double)n_particles, auto_resample,
                                            resample_threshold, n_points);
    OBE_CHECK_LAUNCH("fold2_stop_kernel");
    return 0;
}

}  // namespace obe

using namespace obe;

extern "C" {

int obe_strict_sums(int32_t on) {
    const int prev = g_strict_sums;
    if (on >= 0) g_strict_sums = on != 0;
    return prev;
}

int obe_bayes_update_y(const double* d_y, int64_t ld_y, int32_t n_channels, const double* d_particles,
                       int64_t ld_p, int64_t n_particles, double* d_weights, const double* h_y_meas,
                       const double* h_sigma, const int32_t* h_noise_rows, int32_t n_lik_channels, double choke,
                       void* d_ws, int64_t ws_bytes, double* h_out, void* stream) {
    if (!d_y || !d_weights || n_particles <= 0) return bad_arg("obe_bayes_update_y: bad pointer/size");
    if (n_channels < 1 || n_channels > OBE_MAX_CHANNELS || n_lik_channels > n_channels) return bad_arg("bad channel count");
    if (h_noise_rows && !d_particles) return bad_arg("noise rows need d_particles");
    LikArgs la;
    if (int rc = fill_lik_args(la, h_y_meas, h_sigma, h_noise_rows, n_lik_channels, choke, OBE_CLOUD_MAX_DIMS)) return rc;
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w)) return rc;
    hipStream_t st = as_stream(stream);
    const int nb = update_blocks(n_particles);
    update_y_kernel<<<nb, kBlock, 0, st>>>(la, n_channels, d_y, ld_y, d_particles, ld_p, n_particles, d_weights, w.pa);
    OBE_CHECK_LAUNCH("update_y_kernel");
    return finish_update(w, nb, n_particles, d_weights, h_out, st);
}

int obe_bayes_update_lik(const double* d_lik, int64_t n_particles, double* d_weights, void* d_ws,
                         int64_t ws_bytes, double* h_out, void* stream) {
    if (!d_lik || !d_weights || n_particles <= 0) return bad_arg("obe_bayes_update_lik: bad pointer/size");
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w)) return rc;
    hipStream_t st = as_stream(stream);
    const int nb = update_blocks(n_particles);
    update_lik_kernel<<<nb, kBlock, 0, st>>>(d_lik, n_particles, d_weights, w.pa);
    OBE_CHECK_LAUNCH("update_lik_kernel");
    return finish_update(w, nb, n_particles, d_weights, h_out, st);
}

int obe_likelihood_y(const double* d_y, int64_t ld_y, int32_t n_channels, const double* d_particles,
                     int64_t ld_p, int64_t n_particles, const double* h_y_meas, const double* h_sigma,
                     const int32_t* h_noise_rows, int32_t n_lik_channels, double choke, double* d_lik_out,
                     void* stream) {
    if (!d_y || !d_lik_out || n_particles <= 0) return bad_arg("obe_likelihood_y: bad pointer/size");
    if (n_channels < 1 || n_lik_channels > n_channels || n_lik_channels < 0) return bad_arg("bad channel count");
    if (h_noise_rows && !d_particles) return bad_arg("noise rows need d_particles");
    hipStream_t st = as_stream(stream);
    const int nb = stream_blocks(n_particles, kBlock);
    if (n_lik_channels <= OBE_MAX_CHANNELS) {
        LikArgs la;
        if (int rc = fill_lik_args(la, h_y_meas, h_sigma, h_noise_rows, n_lik_channels, choke, OBE_CLOUD_MAX_DIMS)) return rc;
        likelihood_y_kernel<<<nb, kBlock, 0, st>>>(la, n_lik_channels, d_y, ld_y, d_particles, ld_p, n_particles, d_lik_out);
        OBE_CHECK_LAUNCH("likelihood_y_kernel");
        return 0;
    }
    // a record of more channels than one launch takes (the reference has no limit: obe_base.py:807-824): groups of
    // OBE_MAX_CHANNELS in channel order, each multiplied onto the product so far; the choke at the end
    for (int c0 = 0; c0 < n_lik_channels; c0 += OBE_MAX_CHANNELS) {
        const int nc = n_lik_channels - c0 < OBE_MAX_CHANNELS ? n_lik_channels - c0 : OBE_MAX_CHANNELS;
        LikArgs la;
        if (int rc = fill_lik_args(la, h_y_meas + c0, h_sigma ? h_sigma + c0 : nullptr,
                                   h_noise_rows ? h_noise_rows + c0 : nullptr, nc, NAN, OBE_CLOUD_MAX_DIMS))
            return rc;
        likelihood_y_kernel<<<nb, kBlock, 0, st>>>(la, nc, d_y + (int64_t)c0 * ld_y, ld_y, d_particles, ld_p, n_particles,
                                                   d_lik_out, c0 > 0);
        OBE_CHECK_LAUNCH("likelihood_y_kernel");
    }
    if (!(choke != choke)) {
        choke_kernel<<<nb, kBlock, 0, st>>>(d_lik_out, n_particles, choke);
        OBE_CHECK_LAUNCH("choke_kernel");
    }
    return 0;
}

int obe_weight_sums(const double* d_weights, int64_t n_particles, void* d_ws, int64_t ws_bytes, double* h_out,
                    void* stream) {
    if (!d_weights || n_particles <= 0 || !h_out) return bad_arg("obe_weight_sums: bad pointer/size");
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w)) return rc;
    hipStream_t st = as_stream(stream);
    const int nb = stream_blocks(n_particles, kBlock);
    weight_sums_kernel<<<nb, kBlock, 0, st>>>(d_weights, n_particles, w.pa, w.pb);
    OBE_CHECK_LAUNCH("weight_sums_kernel");
    return fold2_to_host(w.pa, w.pb, nb, w.scalars, h_out, st);
}

int obe_mask_nonpositive(const double* d_particles, int64_t ld_p, int64_t n_particles, const int32_t* h_rows,
                         int32_t n_rows, double* d_weights, int64_t* h_changed, void* d_ws, int64_t ws_bytes,
                         void* stream) {
    if (!d_particles || !d_weights || !h_rows || n_rows < 1 || n_rows > OBE_MAX_DIMS || n_particles <= 0)
        return bad_arg("obe_mask_nonpositive: bad pointer/size");
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w)) return rc;
    RowsArg ra{};
    ra.n = n_rows;
    for (int k = 0; k < n_rows; ++k) ra.rows[k] = h_rows[k];
    hipStream_t st = as_stream(stream);
    const int nb = stream_blocks(n_particles, kBlock);
    mask_kernel<<<nb, kBlock, 0, st>>>(ra, d_particles, ld_p, n_particles, d_weights, w.pa, w.pb);
    OBE_CHECK_LAUNCH("mask_kernel");
    fold2_kernel<<<1, kBlock, 0, st>>>(w.pa, w.pb, nb, w.scalars, nullptr);
    OBE_CHECK_LAUNCH("fold2_kernel");
    int64_t* hv = static_cast<int64_t*>(device_view_of_host(h_changed));
    if (hv) arm_host_word(h_changed);
    mask_renorm_kernel<<<nb, kBlock, 0, st>>>(w.scalars, n_particles, d_weights, hv);
    OBE_CHECK_LAUNCH("mask_renorm_kernel");
    if (h_changed) {
        if (hv) {
            // (the count is the kernel's first store; the renormalisation that may still be running is ordered
            // before everything the caller enqueues next)
            if (int rc = wait_host_word(h_changed, st)) return rc;
        } else {
            double sc[2];
            OBE_HIP_TRY(hipMemcpyAsync(sc, w.scalars, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            OBE_HIP_TRY(hipStreamSynchronize(st));
            *h_changed = (int64_t)sc[1];
        }
    }
    return 0;
}

static int mask_renorm_moments(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                               const double* psum, const double* pcount, int nb, double* d_weights, double* d_moments,
                               double* h_moments, double* hm, int64_t* h_changed, int64_t* hc, unsigned* counter,
                               double* partials_mom, hipStream_t st);

int obe_mask_renorm_moments(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                            const double* d_mask_partials, double* d_weights, double* d_moments, double* h_moments,
                            int64_t* h_changed, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_particles || !d_weights || !d_mask_partials || !d_moments || n_particles <= 0)
        return bad_arg("obe_mask_renorm_moments: bad pointer/size");
    if (n_dims < 1 || n_dims > kFastDims) return bad_arg("obe_mask_renorm_moments: n_dims must be 1..16 (OBE_FAST_DIMS)");
    hipStream_t st = as_stream(stream);
    unsigned* counter = stream_control_words(st);
    int64_t* hc = static_cast<int64_t*>(device_view_of_host(h_changed));
    double* hm = static_cast<double*>(device_view_of_host(h_moments));
    // (refused before any launch: the caller then runs obe_mask_nonpositive_moments(), which on weights the gather
    // has already masked finds the same particles and leaves the same bits)
    if (!counter || (h_changed && !hc) || (h_moments && !hm))
        return bad_arg("obe_mask_renorm_moments: needs an arrival counter for the stream and page-locked host outputs");
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w, n_dims)) return rc;
    return mask_renorm_moments(d_particles, ld_p, n_dims, n_particles, d_mask_partials, d_mask_partials + kMaxBlocks,
                               stream_blocks(n_particles, kBlock), d_weights, d_moments, h_moments, hm, h_changed, hc,
                               counter, w.mom, st);
}

int obe_mask_nonpositive_moments(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                                 const int32_t* h_rows, int32_t n_rows, double* d_weights, double* d_moments,
                                 double* h_moments, int64_t* h_changed, void* d_ws, int64_t ws_bytes, void* stream) {
    if (!d_particles || !d_weights || !h_rows || !d_moments || n_rows < 1 || n_rows > OBE_MAX_DIMS || n_particles <= 0)
        return bad_arg("obe_mask_nonpositive_moments: bad pointer/size");
    if (n_dims < 1 || n_dims > OBE_CLOUD_MAX_DIMS) return bad_arg("obe_mask_nonpositive_moments: n_dims must be 1..1024");
    hipStream_t st = as_stream(stream);
    unsigned* counter = stream_control_words(st);
    int64_t* hc = static_cast<int64_t*>(device_view_of_host(h_changed));
    double* hm = static_cast<double*>(device_view_of_host(h_moments));
    if (!counter || (h_changed && !hc) || (h_moments && !hm) || n_dims > kFastDims) {
        // no arrival counter for this stream / pageable host buffers / a cloud wider than the fused kernels are
        // compiled for: the two separate calls (synchronous)
        if (int rc = obe_mask_nonpositive(d_particles, ld_p, n_particles, h_rows, n_rows, d_weights, h_changed, d_ws,
                                          ws_bytes, stream))
            return rc;
        return obe_moments(d_particles, ld_p, n_dims, n_particles, d_weights, 0, d_moments, h_moments, d_ws, ws_bytes,
                           stream);
    }
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w, n_dims)) return rc;
    RowsArg ra{};
    ra.n = n_rows;
    for (int k = 0; k < n_rows; ++k) {
        if (h_rows[k] < 0 || h_rows[k] >= n_dims) return bad_arg("obe_mask_nonpositive_moments: row index out of range");
        ra.rows[k] = h_rows[k];
    }
    const int nb = stream_blocks(n_particles, kBlock);
    mask_kernel<<<nb, kBlock, 0, st>>>(ra, d_particles, ld_p, n_particles, d_weights, w.pa, w.pb);
    OBE_CHECK_LAUNCH("mask_kernel");
    return mask_renorm_moments(d_particles, ld_p, n_dims, n_particles, w.pa, w.pb, nb, d_weights, d_moments, h_moments,
                               hm, h_changed, hc, counter, w.mom, st);
}

// the second half: renormalise if anything was zeroed + the first moments of the constrained cloud, from the partial
// sums {sum w, count} that mask_kernel — or the masked gather of a resample (obe_resample_particles_aos_masked) — left
static int mask_renorm_moments(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                               const double* psum, const double* pcount, int nb, double* d_weights, double* d_moments,
                               double* h_moments, double* hm, int64_t* h_changed, int64_t* hc, unsigned* counter,
                               double* partials_mom, hipStream_t st) {
    if (hc) arm_host_word(h_changed);
    if (hm) arm_host_words(h_moments, 2 + 4 * (int64_t)n_dims);
    const MaskFold mf{counter, d_moments, hm, hc};
    const int nm = first_moment_blocks(n_particles, n_dims);
#define OBE_MASK_MOM_CASE(DD)                                                                                       \
    case DD:                                                                                                        \
        mask_renorm_moments_kernel<DD><<<nm, kMomThreads, 0, st>>>(psum, pcount, nb, d_particles, ld_p, n_particles,    \
                                                              d_weights, partials_mom, mf);                         \
        break;
    switch (n_dims) {
        OBE_MASK_MOM_CASE(1) OBE_MASK_MOM_CASE(2) OBE_MASK_MOM_CASE(3) OBE_MASK_MOM_CASE(4) OBE_MASK_MOM_CASE(5)
        OBE_MASK_MOM_CASE(6) OBE_MASK_MOM_CASE(7) OBE_MASK_MOM_CASE(8) OBE_MASK_MOM_CASE(9) OBE_MASK_MOM_CASE(10)
        OBE_MASK_MOM_CASE(11) OBE_MASK_MOM_CASE(12) OBE_MASK_MOM_CASE(13) OBE_MASK_MOM_CASE(14)
        OBE_MASK_MOM_CASE(15) OBE_MASK_MOM_CASE(16)
    }
#undef OBE_MASK_MOM_CASE
    OBE_CHECK_LAUNCH("mask_renorm_moments_kernel");
    return 0;
}

int obe_power_normalize(const double* d_u, int64_t n, double exponent, double* d_p_out, void* d_ws,
                        int64_t ws_bytes, void* stream) {
    if (!d_u || !d_p_out || n <= 0) return bad_arg("obe_power_normalize: bad pointer/size");
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w)) return rc;
    hipStream_t st = as_stream(stream);
    const int nb = stream_blocks(n, kBlock);
    power_kernel<<<nb, kBlock, 0, st>>>(d_u, n, exponent, d_p_out, w.pa);
    OBE_CHECK_LAUNCH("power_kernel");
    divide_by_total_kernel<<<nb, kBlock, 0, st>>>(w.pa, nb, n, d_p_out);
    OBE_CHECK_LAUNCH("divide_by_total_kernel");
    return 0;
}

int obe_noise_var_from_moments(const double* d_moments, int32_t n_dims, const int32_t* h_rows, int32_t n_rows,
                               double* d_out, void* stream) {
    if (!d_moments || !h_rows || !d_out || n_rows < 1 || n_rows > OBE_MAX_DIMS)
        return bad_arg("obe_noise_var_from_moments: bad pointer/size");
    RowsArg ra{};
    ra.n = n_rows;
    for (int k = 0; k < n_rows; ++k) {
        if (h_rows[k] < 0 || h_rows[k] >= n_dims) return bad_arg("noise row index out of range");
        ra.rows[k] = h_rows[k];
    }
    noise_var_kernel<<<1, kWave, 0, as_stream(stream)>>>(d_moments, n_dims, ra, d_out);
    OBE_CHECK_LAUNCH("noise_var_kernel");
    return 0;
}

}  // extern "C"
