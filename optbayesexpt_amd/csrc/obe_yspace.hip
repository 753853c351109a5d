// The reference's non-default utilities work on the explicit "y-space"
// utility_y_space (N_draws, C, N_s) — the model evaluated for every drawn parameter set
// over all settings (obe_base.py:480-484, 508-512, 526-530, 713-715) — and reduce it along
// the draw axis per (channel, setting):
//   max_min       span^2 = (max - min)^2                       obe_base.py:520-535
//   pseudo        exp(2 H) / (2 pi e), H = differential entropy  obe_base.py:491-518
//   full_kld      exp(H(y + noise) - H(noise)) - 1              obe_base.py:688-720
// H is scipy.stats.differential_entropy(method='auto'): sort the draws, then a spacing
// estimator chosen by sample size — van Es (n <= 10), Ebrahimi et al. (n <= 1000),
// Vasicek (n > 1000) — with window m = floor(sqrt(n) + 0.5).
// N_draws is small (default 30): one thread per (channel, setting) sorts its column in
// registers/scratch; loads are coalesced across settings.  HBM-bound: 8 N_d bytes per output.
#include "obe_common.h"
#include "obe_models.h"

namespace obe {

constexpr int kMaxDraws = 2048;      // draws per column the sort kernels accept

// y-space from a device model: y[d][c][s] = model(x_s; particles[:, idx[d]]), exact form
template <class M>
__global__ __launch_bounds__(kBlock) void eval_draws_kernel(obe_model m, const double* __restrict__ settings,
                                                            int64_t ld_s, int64_t ns,
                                                            const double* __restrict__ particles, int64_t ld_p,
                                                            int64_t n_particles,
                                                            const int64_t* __restrict__ idx, int64_t nd,
                                                            double* __restrict__ ysp) {
    const int64_t total = nd * ns;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t d = e / ns, s = e - d * ns;
        int64_t src = idx[d];
        src = src < 0 ? 0 : (src >= n_particles ? n_particles - 1 : src);
        double x[M::NS], y[M::NC];
#pragma unroll
        for (int k = 0; k < M::NS; ++k) x[k] = settings[(int64_t)k * ld_s + s];
        M::eval(x, ParamRef{particles + src, ld_p}, m, y);
#pragma unroll
        for (int c = 0; c < M::NC; ++c) ysp[(d * M::NC + c) * ns + s] = y[c];
    }
}

// y[d][c][s] += noise[d][c]      (obe_base.py:714-715)
__global__ __launch_bounds__(kBlock) void add_noise_kernel(double* __restrict__ ysp, int64_t nd, int nc, int64_t ns,
                                                           const double* __restrict__ noise) {
    const int64_t total = nd * nc * ns;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
        const int64_t dc = e / ns;
        ysp[e] = ysp[e] + noise[dc];
    }
}

__global__ __launch_bounds__(kBlock) void maxmin_kernel(const double* __restrict__ ysp, int64_t nd, int64_t row,
                                                        double* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < row; e += (int64_t)gridDim.x * kBlock) {
        double lo = ysp[e], hi = lo;
        for (int64_t d0 = 1; d0 < nd; d0 += 8) {           // (eight draws' loads in flight together)
            double t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = ysp[(d0 + j < nd ? d0 + j : nd - 1) * row + e];
#pragma unroll
            for (int j = 0; j < 8; ++j) {                  // (a clamped repeat of the last draw changes nothing)
                const double v = t[j];
                lo = v < lo ? v : lo;          // np.max / np.min propagate NaN; a NaN model output is
                hi = v > hi ? v : hi;          // outside the supported domain here
            }
        }
        const double span = hi - lo;
        out[e] = span * span;
    }
}

// One thread per column: gather, insertion sort in a per-thread scratch column, estimate.
// `scratch` is (n, row) so that neighbouring threads touch neighbouring addresses.
__global__ __launch_bounds__(kBlock) void entropy_kernel(const double* __restrict__ ysp, int nd, int64_t row, int m,
                                                         int as_variance, double* __restrict__ scratch,
                                                         double* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < row; e += (int64_t)gridDim.x * kBlock) {
        double* col = scratch + e;                       // col[k * row]
        for (int d = 0; d < nd; ++d) {                    // insertion sort while copying
            const double v = ysp[(int64_t)d * row + e];
            int k = d;
            while (k > 0 && col[(int64_t)(k - 1) * row] > v) {
                col[(int64_t)k * row] = col[(int64_t)(k - 1) * row];
                --k;
            }
            col[(int64_t)k * row] = v;
        }
        // the estimators index the sorted column; read it through a small strided view
        struct View {
            const double* p;
            int64_t stride;
            __device__ double operator[](int i) const { return p[(int64_t)i * stride]; }
        } x{col, row};
        const double dn = (double)nd, dm = (double)m;
        double h;
        if (nd <= 10) {
            double acc = 0.0;
            for (int i = 0; i + m < nd; ++i) acc += log((dn + 1.0) / dm * (x[i + m] - x[i]));
            double harm = 0.0;
            for (int k = m; k <= nd; ++k) harm += 1.0 / (double)k;
            h = 1.0 / (dn - dm) * acc + harm + log(dm) - log(dn + 1.0);
        } else {
            double acc = 0.0;
            for (int i = 1; i <= nd; ++i) {
                const int lo = i - 1 - m < 0 ? 0 : i - 1 - m, hi = i - 1 + m > nd - 1 ? nd - 1 : i - 1 + m;
                const double diff = x[hi] - x[lo];
                if (nd <= 1000) {
                    double ci = 2.0;
                    if (i <= m) ci = 1.0 + (double)(i - 1) / dm;
                    if (i >= nd - m + 1) ci = 1.0 + (double)(nd - i) / dm;
                    acc += log(dn * diff / (ci * dm));
                } else {
                    acc += log(dn / (2.0 * dm) * diff);
                }
            }
            h = acc / dn;
        }
        // pseudo-utility: variance of the normal with the same entropy, exp(2H)/(2 pi e)
        out[e] = as_variance ? exp(2.0 * h) / (2.0 * 3.141592653589793 * 2.718281828459045) : h;
    }
}

// utility_full_kld: exp(H_y - H_noise[c]) - 1 over the flattened (C, N_s) array
__global__ __launch_bounds__(kBlock) void kld_kernel(const double* __restrict__ hy, int nc, int64_t ns,
                                                     const double* __restrict__ hn, double* __restrict__ out) {
    const int64_t total = (int64_t)nc * ns;
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock)
        out[e] = exp(hy[e] - hn[e / ns]) - 1.0;
}

}  // namespace obe

using namespace obe;

extern "C" {

int obe_eval_draws(const obe_model* m, const double* d_settings, int64_t ld_s, int64_t n_settings,
                   const double* d_particles, int64_t ld_p, int64_t n_particles, const int64_t* d_draw_idx,
                   int64_t n_draws, double* d_yspace, void* stream) {
    if (!m || !d_settings || !d_particles || !d_draw_idx || !d_yspace || n_settings <= 0 || n_draws <= 0)
        return bad_arg("obe_eval_draws: bad pointer/size");
    obe_model mm = *m;
    if (int rc = obe_model_validate(&mm)) return rc;
    hipStream_t st = as_stream(stream);
    return dispatch_model(mm, [&](auto M) -> int {
        using Model = decltype(M);
        eval_draws_kernel<Model><<<stream_blocks(n_draws * n_settings, kBlock), kBlock, 0, st>>>(
            mm, d_settings, ld_s, n_settings, d_particles, ld_p, n_particles, d_draw_idx, n_draws, d_yspace);
        OBE_CHECK_LAUNCH("eval_draws_kernel");
        return 0;
    });
}

int obe_yspace_add_noise(double* d_yspace, int64_t n_draws, int32_t n_channels, int64_t n_settings,
                         const double* d_noise, void* stream) {
    if (!d_yspace || !d_noise || n_draws <= 0 || n_channels < 1 || n_settings <= 0)
        return bad_arg("obe_yspace_add_noise: bad pointer/size");
    add_noise_kernel<<<stream_blocks(n_draws * n_channels * n_settings, kBlock), kBlock, 0, as_stream(stream)>>>(
        d_yspace, n_draws, n_channels, n_settings, d_noise);
    OBE_CHECK_LAUNCH("add_noise_kernel");
    return 0;
}

int obe_yspace_maxmin(const double* d_yspace, int64_t n_draws, int64_t n_columns, double* d_span2, void* stream) {
    if (!d_yspace || !d_span2 || n_draws <= 0 || n_columns <= 0) return bad_arg("obe_yspace_maxmin: bad pointer/size");
    maxmin_kernel<<<stream_blocks(n_columns, kBlock), kBlock, 0, as_stream(stream)>>>(d_yspace, n_draws, n_columns,
                                                                                      d_span2);
    OBE_CHECK_LAUNCH("maxmin_kernel");
    return 0;
}

int obe_yspace_entropy(const double* d_yspace, int64_t n_draws, int64_t n_columns, int32_t as_variance,
                       double* d_scratch, double* d_out, void* stream) {
    if (!d_yspace || !d_out || !d_scratch || n_columns <= 0) return bad_arg("obe_yspace_entropy: bad pointer/size");
    if (n_draws > kMaxDraws) return bad_arg("obe_yspace_entropy: more than 2048 draws");
    const int m = (int)floor(sqrt((double)n_draws) + 0.5);
    if (!(2 <= 2 * m && 2 * m < n_draws)) {
        // scipy: "Window length (m) must be positive and less than half the sample size (n)."
        return bad_arg("obe_yspace_entropy: window length must be positive and less than half the sample size");
    }
    entropy_kernel<<<stream_blocks(n_columns, kBlock), kBlock, 0, as_stream(stream)>>>(
        d_yspace, (int)n_draws, n_columns, m, as_variance, d_scratch, d_out);
    OBE_CHECK_LAUNCH("entropy_kernel");
    return 0;
}

int obe_kld_utility(const double* d_entropy_y, int32_t n_channels, int64_t n_settings, const double* d_entropy_noise,
                    double* d_utility, void* stream) {
    if (!d_entropy_y || !d_entropy_noise || !d_utility || n_channels < 1 || n_settings <= 0)
        return bad_arg("obe_kld_utility: bad pointer/size");
    kld_kernel<<<stream_blocks((int64_t)n_channels * n_settings, kBlock), kBlock, 0, as_stream(stream)>>>(
        d_entropy_y, n_channels, n_settings, d_entropy_noise, d_utility);
    OBE_CHECK_LAUNCH("kld_kernel");
    return 0;
}

}  // extern "C"
