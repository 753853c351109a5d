// K3 — weighted mean / std / covariance of the particle cloud (particlepdf.py:173-214).
// HBM-bound: pass 1 reads (D+1) rows once (8(D+1) B per particle) and produces
// sum w, sum w^2, sum w x_i, sum w x_i^2; pass 2 (covariance only, i.e. only on
// resample cycles) reads them again for sum w (x_i - mu_i)(x_j - mu_j).
// All accumulators live in registers (template on D), partials are folded by one
// wavefront per output value in a fixed order.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "obe_common.h"
#include "obe_moments.h"

namespace obe {

// values: [0] sum w, [1] sum w*w, [2+i] sum x_i*w, [2+D+i] sum (x_i*x_i)*w
// FOLD (round 4): the workgroup that arrives last folds everybody's block partials (written through) and
// derives mean / std in the same launch — the same sums in the same order as the separate one-workgroup
// fold kernels (which remain for streams without an arrival counter): identical bits, one launch less.
template <int D, bool FOLD>
__global__ __launch_bounds__(kBlock) void moments_pass1(const double* __restrict__ x, int64_t ld, int64_t n,
                                                        const double* __restrict__ w, double* partials,
                                                        MomentsOut mo) {
    double v[2 + 2 * D];
#pragma unroll
    for (int k = 0; k < 2 + 2 * D; ++k) v[k] = 0.0;
    // OBE_MOM_UNROLL particles per trip (a compile-time tuning aid, 1 in the product build: with 256
    // workgroups of waves already keeping (D + 1) loads each in flight, two per trip measured no
    // faster); any value adds in the same order per accumulator as one at a time
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += OBE_MOM_UNROLL * stride) {
        double wp[OBE_MOM_UNROLL], xi[OBE_MOM_UNROLL][D];
#pragma unroll
        for (int u = 0; u < OBE_MOM_UNROLL; ++u) {
            const int64_t q = p + u * stride;
            const bool ok = q < n;
            wp[u] = ok ? w[q] : 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) xi[u][i] = ok ? x[(int64_t)i * ld + q] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < OBE_MOM_UNROLL; ++u) {
            if (p + u * stride < n) accumulate_first_moments<D>(v, wp[u], xi[u]);
        }
    }
    store_block_partials<2 + 2 * D, FOLD>(v, partials);
    if constexpr (FOLD) {
        __shared__ int last;
        if (!arrive_last<false>(mo.counter, &last)) return;
        __shared__ double raw[kMaxMomentValues];
        fold_values_block<kBlock, true, 8>(partials, gridDim.x, 2 + 2 * D, raw);
        derive_first_moments(raw, D, mo.out, mo.host);
        raise_host_flag(mo.host_flag);
    }
}

// values: upper triangle (i <= j) of sum (x_i - mu_i) * ((x_j - mu_j) * w), row-major
template <int D, bool FOLD>
__global__ __launch_bounds__(kBlock) void moments_pass2(const double* __restrict__ x, int64_t ld, int64_t n,
                                                        const double* __restrict__ w,
                                                        const double* out /* mean at out+2 */, double* partials,
                                                        MomentsOut mo) {
    constexpr int NV = D * (D + 1) / 2;
    double v[NV];
    double mu[D];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) mu[i] = out[2 + i];
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += OBE_MOM_UNROLL * stride) {
        double wp[OBE_MOM_UNROLL], dev[OBE_MOM_UNROLL][D];
#pragma unroll
        for (int u = 0; u < OBE_MOM_UNROLL; ++u) {
            const int64_t q = p + u * stride;
            const bool ok = q < n;
            wp[u] = ok ? w[q] : 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) dev[u][i] = (ok ? x[(int64_t)i * ld + q] : mu[i]) - mu[i];
        }
#pragma unroll
        for (int u = 0; u < OBE_MOM_UNROLL; ++u) {
            if (p + u * stride < n) {
                int k = 0;
#pragma unroll
                for (int i = 0; i < D; ++i) {
#pragma unroll
                    for (int j = i; j < D; ++j) {
                        v[k] = fma(dev[u][i], dev[u][j] * wp[u], v[k]);
                        ++k;
                    }
                }
            }
        }
    }
    store_block_partials<NV, FOLD>(v, partials);
    if constexpr (FOLD) {
        __shared__ int last;
        if (!arrive_last<false>(mo.counter, &last)) return;
        __shared__ double raw[kMaxMomentValues];
        fold_values_block<kBlock, true, 8>(partials, gridDim.x, NV, raw);
        derive_covariance(raw, D, mo.out, mo.host);
        raise_host_flag(mo.host_flag);
    }
}

// fold + derive in one single-workgroup launch
// out: [0]=W [1]=W2 [2..) mean [2+D..) m1 [2+2D..) m2 [2+3D..) std
// (host: the device view of the caller's page-locked h_out, or NULL — then obe_moments copies)
__global__ __launch_bounds__(kFoldThreads) void fold_derive_pass1(const double* __restrict__ partials, int nb, int d,
                                                            MomentsOut mo) {
    __shared__ double raw[kMaxMomentValues];
    fold_values_block(partials, nb, 2 + 2 * d, raw);
    derive_first_moments(raw, d, mo.out, mo.host);
    raise_host_flag(mo.host_flag);
}

__global__ __launch_bounds__(kFoldThreads) void fold_derive_pass2(const double* __restrict__ partials, int nb, int d,
                                                            MomentsOut mo) {
    __shared__ double raw[kMaxMomentValues];
    fold_values_block(partials, nb, d * (d + 1) / 2, raw);
    derive_covariance(raw, d, mo.out, mo.host);
    raise_host_flag(mo.host_flag);
}

template <int D>
static int launch_moments(const double* x, int64_t ld, int64_t n, const double* w, int want_cov, double* partials,
                          const MomentsOut& mo, hipStream_t st) {
    const int nb = moment_blocks(n, D);
    MomentsOut first = mo;
    if (want_cov) first.host_flag = nullptr;            // (the flag belongs to the last launch of the call)
    if (want_cov != 2) {            // (2: `out` already holds the first moments of these weights)
        if (mo.counter) {
            moments_pass1<D, true><<<nb, kBlock, 0, st>>>(x, ld, n, w, partials, first);
            OBE_CHECK_LAUNCH("moments_pass1");
        } else {
            moments_pass1<D, false><<<nb, kBlock, 0, st>>>(x, ld, n, w, partials, first);
            OBE_CHECK_LAUNCH("moments_pass1");
            fold_derive_pass1<<<1, kFoldThreads, 0, st>>>(partials, nb, D, first);
            OBE_CHECK_LAUNCH("fold_derive_pass1");
        }
    }
    if (want_cov) {
        if (mo.counter) {
            moments_pass2<D, true><<<nb, kBlock, 0, st>>>(x, ld, n, w, mo.out, partials, mo);
            OBE_CHECK_LAUNCH("moments_pass2");
        } else {
            moments_pass2<D, false><<<nb, kBlock, 0, st>>>(x, ld, n, w, mo.out, partials, mo);
            OBE_CHECK_LAUNCH("moments_pass2");
            fold_derive_pass2<<<1, kFoldThreads, 0, st>>>(partials, nb, D, mo);
            OBE_CHECK_LAUNCH("fold_derive_pass2");
        }
    }
    return 0;
}

// obe_moments with an optional page-locked flag word that the call's last kernel raises (obe_resample_begin)
int moments_call(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles, const double* d_weights,
                 int32_t want_cov, double* d_out, double* h_out, uint64_t* h_flag, void* d_ws, int64_t ws_bytes,
                 hipStream_t st, bool* host_written) {
    if (!d_particles || !d_weights || !d_out || n_particles <= 0) return bad_arg("obe_moments: bad pointer/size");
    if (n_dims < 1 || n_dims > OBE_MAX_DIMS) return bad_arg("obe_moments: n_dims must be 1..16");
    const int64_t nv_max = std::max<int64_t>(2 + 2 * n_dims, (int64_t)n_dims * (n_dims + 1) / 2);
    const int64_t need = ((int64_t)kMomGridCap * nv_max + nv_max) * sizeof(double);
    if (!d_ws || ws_bytes < need) return bad_arg("obe_moments: workspace too small");
    double* partials = static_cast<double*>(d_ws);
    double* hv = static_cast<double*>(device_view_of_host(h_out));     // page-locked h_out: the kernels write it
    uint64_t* hf = hv ? static_cast<uint64_t*>(device_view_of_host(h_flag)) : nullptr;
    static const bool separate = getenv("OBE_MOMENTS_FOLD") && !strcmp(getenv("OBE_MOMENTS_FOLD"), "separate");
    const MomentsOut mo{d_out, hv, hf, separate ? nullptr : stream_control_words(st)};
    if (host_written) *host_written = hv != nullptr;
    int rc = -1;
#define OBE_MOM_CASE(DD) \
    case DD: rc = launch_moments<DD>(d_particles, ld_p, n_particles, d_weights, want_cov, partials, mo, st); break;
    switch (n_dims) {
        OBE_MOM_CASE(1) OBE_MOM_CASE(2) OBE_MOM_CASE(3) OBE_MOM_CASE(4) OBE_MOM_CASE(5) OBE_MOM_CASE(6)
        OBE_MOM_CASE(7) OBE_MOM_CASE(8) OBE_MOM_CASE(9) OBE_MOM_CASE(10) OBE_MOM_CASE(11) OBE_MOM_CASE(12)
        OBE_MOM_CASE(13) OBE_MOM_CASE(14) OBE_MOM_CASE(15) OBE_MOM_CASE(16)
    }
#undef OBE_MOM_CASE
    return rc;
}

}  // namespace obe

using namespace obe;

extern "C" {

int64_t obe_moments_len(int32_t n_dims) { return 2 + 4 * (int64_t)n_dims + (int64_t)n_dims * n_dims; }

int obe_moments(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                const double* d_weights, int32_t want_cov, double* d_out, double* h_out, void* d_ws,
                int64_t ws_bytes, void* stream) {
    hipStream_t st = as_stream(stream);
    bool host_written = false;
    if (int rc = moments_call(d_particles, ld_p, n_dims, n_particles, d_weights, want_cov, d_out, h_out, nullptr, d_ws,
                              ws_bytes, st, &host_written))
        return rc;
    if (h_out) {
        const int64_t len = want_cov ? obe_moments_len(n_dims) : 2 + 4 * (int64_t)n_dims;
        if (!host_written) OBE_HIP_TRY(hipMemcpyAsync(h_out, d_out, len * sizeof(double), hipMemcpyDeviceToHost, st));
        if (!defer_host_sync()) OBE_HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

}  // extern "C"
