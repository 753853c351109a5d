// K3 — weighted mean / std / covariance of the particle cloud (particlepdf.py:173-214).
// HBM-bound: pass 1 reads (D+1) rows once (8(D+1) B per particle) and produces
// sum w, sum w^2, sum w x_i, sum w x_i^2; pass 2 (covariance only, i.e. only on
// resample cycles) reads them again for sum w (x_i - mu_i)(x_j - mu_j).
// All accumulators live in registers (template on D), partials are folded by one
// wavefront per output value in a fixed order.
#include "obe_common.h"

namespace obe {

constexpr int kMomBlocks = 1024;   // grid cap for the moment passes

// The NV block sums with ONE barrier: every wavefront reduces all its values by shuffles and
// parks them in LDS, then thread k adds the wave sums of value k in wave order — the same
// arithmetic as NV calls of block_sum (shuffle tree, then the waves in order), which cost two
// barriers each (110 of them at D = 10).
template <int NV>
__device__ __forceinline__ void store_block_partials(double (&v)[NV], double* __restrict__ partials) {
    constexpr int NW = kBlock / kWave;
    __shared__ double red[NW][NV];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double s = wave_sum(v[k]);
        if (lane == 0) red[wid][k] = s;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NV; k += kBlock) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += red[i][k];
        partials[(int64_t)blockIdx.x * NV + k] = s;
    }
}

// values: [0] sum w, [1] sum w*w, [2+i] sum x_i*w, [2+D+i] sum (x_i*x_i)*w
template <int D>
__global__ __launch_bounds__(kBlock) void moments_pass1(const double* __restrict__ x, int64_t ld, int64_t n,
                                                        const double* __restrict__ w,
                                                        double* __restrict__ partials) {
    double v[2 + 2 * D];
#pragma unroll
    for (int k = 0; k < 2 + 2 * D; ++k) v[k] = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const double wp = w[p];
        v[0] += wp;
        v[1] += wp * wp;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double xi = x[(int64_t)i * ld + p];
            v[2 + i] += xi * wp;
            v[2 + D + i] += (xi * xi) * wp;
        }
    }
    store_block_partials<2 + 2 * D>(v, partials);
}

// values: upper triangle (i <= j) of sum (x_i - mu_i) * ((x_j - mu_j) * w), row-major
template <int D>
__global__ __launch_bounds__(kBlock) void moments_pass2(const double* __restrict__ x, int64_t ld, int64_t n,
                                                        const double* __restrict__ w,
                                                        const double* __restrict__ out /* mean at out+2 */,
                                                        double* __restrict__ partials) {
    constexpr int NV = D * (D + 1) / 2;
    double v[NV];
    double mu[D];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) mu[i] = out[2 + i];
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        const double wp = w[p];
        double dev[D];
#pragma unroll
        for (int i = 0; i < D; ++i) dev[i] = x[(int64_t)i * ld + p] - mu[i];
        int k = 0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
#pragma unroll
            for (int j = i; j < D; ++j) {
                v[k] = fma(dev[i], dev[j] * wp, v[k]);
                ++k;
            }
        }
    }
    store_block_partials<NV>(v, partials);
}

// raw[k] = sum_b partials[b*nv + k]: one wavefront per value (strided partial sums, then the
// shuffle tree: a fixed order), the four waves of the block taking the values in turn; `vals` in LDS.
__device__ __forceinline__ void fold_values_block(const double* __restrict__ partials, int nb, int nv,
                                                  double* __restrict__ vals) {
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    for (int k = wid; k < nv; k += kBlock / kWave) {
        double s = 0.0;
        for (int b = lane; b < nb; b += kWave) s += partials[(int64_t)b * nv + k];
        s = wave_sum(s);
        if (lane == 0) vals[k] = s;
    }
    __syncthreads();
}

constexpr int kMaxMomentValues = OBE_MAX_DIMS * (OBE_MAX_DIMS + 1) / 2;     // >= 2 + 2 D

// fold + derive in one single-workgroup launch
// out: [0]=W [1]=W2 [2..) mean [2+D..) m1 [2+2D..) m2 [2+3D..) std
__global__ __launch_bounds__(kBlock) void fold_derive_pass1(const double* __restrict__ partials, int nb, int d,
                                                            double* __restrict__ out) {
    __shared__ double raw[kMaxMomentValues];
    fold_values_block(partials, nb, 2 + 2 * d, raw);
    const int i = threadIdx.x;
    if (i == 0) {
        out[0] = raw[0];
        out[1] = raw[1];
    }
    if (i < d) {
        const double m1 = raw[2 + i], m2 = raw[2 + d + i];
        out[2 + i] = m1 / raw[0];             // np.average: sum(x w) / sum(w)
        out[2 + d + i] = m1;
        out[2 + 2 * d + i] = m2;
        out[2 + 3 * d + i] = sqrt(m2 - m1 * m1);   // particlepdf.py:211-214
    }
}

// cov = S * (1 / (W - W2/W))  (np.cov scales by the reciprocal)
__global__ __launch_bounds__(kBlock) void fold_derive_pass2(const double* __restrict__ partials, int nb, int d,
                                                            double* __restrict__ out) {
    __shared__ double raw[kMaxMomentValues];
    fold_values_block(partials, nb, d * (d + 1) / 2, raw);
    const double fact = out[0] - out[1] / out[0];
    const double scale = 1.0 / fact;
    double* cov = out + 2 + 4 * d;
    for (int e = threadIdx.x; e < d * d; e += kBlock) {
        int i = e / d, j = e % d;
        if (i > j) {
            const int t = i;
            i = j;
            j = t;
        }
        const int k = i * d - i * (i - 1) / 2 + (j - i);     // index of (i, j), i <= j, in the packed upper triangle
        cov[e] = raw[k] * scale;
    }
}

template <int D>
static int launch_moments(const double* x, int64_t ld, int64_t n, const double* w, int want_cov, double* out,
                          double* partials, double* raw, hipStream_t st) {
    const int nb = static_cast<int>(std::min<int64_t>(kMomBlocks, (n + kBlock - 1) / kBlock));
    constexpr int NV1 = 2 + 2 * D;
    moments_pass1<D><<<nb, kBlock, 0, st>>>(x, ld, n, w, partials);
    OBE_CHECK_LAUNCH("moments_pass1");
    fold_derive_pass1<<<1, kBlock, 0, st>>>(partials, nb, D, out);
    OBE_CHECK_LAUNCH("fold_derive_pass1");
    if (want_cov) {
        moments_pass2<D><<<nb, kBlock, 0, st>>>(x, ld, n, w, out, partials);
        OBE_CHECK_LAUNCH("moments_pass2");
        fold_derive_pass2<<<1, kBlock, 0, st>>>(partials, nb, D, out);
        OBE_CHECK_LAUNCH("fold_derive_pass2");
    }
    return 0;
}

}  // namespace obe

using namespace obe;

extern "C" {

int64_t obe_moments_len(int32_t n_dims) { return 2 + 4 * (int64_t)n_dims + (int64_t)n_dims * n_dims; }

int obe_moments(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                const double* d_weights, int32_t want_cov, double* d_out, double* h_out, void* d_ws,
                int64_t ws_bytes, void* stream) {
    if (!d_particles || !d_weights || !d_out || n_particles <= 0) return bad_arg("obe_moments: bad pointer/size");
    if (n_dims < 1 || n_dims > OBE_MAX_DIMS) return bad_arg("obe_moments: n_dims must be 1..16");
    const int64_t nv_max = std::max<int64_t>(2 + 2 * n_dims, (int64_t)n_dims * (n_dims + 1) / 2);
    const int64_t need = ((int64_t)kMomBlocks * nv_max + nv_max) * sizeof(double);
    if (!d_ws || ws_bytes < need) return bad_arg("obe_moments: workspace too small");
    double* partials = static_cast<double*>(d_ws);
    double* raw = partials + (int64_t)kMomBlocks * nv_max;
    hipStream_t st = as_stream(stream);
    int rc = -1;
#define OBE_MOM_CASE(DD) \
    case DD: rc = launch_moments<DD>(d_particles, ld_p, n_particles, d_weights, want_cov, d_out, partials, raw, st); break;
    switch (n_dims) {
        OBE_MOM_CASE(1) OBE_MOM_CASE(2) OBE_MOM_CASE(3) OBE_MOM_CASE(4) OBE_MOM_CASE(5) OBE_MOM_CASE(6)
        OBE_MOM_CASE(7) OBE_MOM_CASE(8) OBE_MOM_CASE(9) OBE_MOM_CASE(10) OBE_MOM_CASE(11) OBE_MOM_CASE(12)
        OBE_MOM_CASE(13) OBE_MOM_CASE(14) OBE_MOM_CASE(15) OBE_MOM_CASE(16)
    }
#undef OBE_MOM_CASE
    if (rc) return rc;
    if (h_out) {
        const int64_t len = want_cov ? obe_moments_len(n_dims) : 2 + 4 * (int64_t)n_dims;
        OBE_HIP_TRY(hipMemcpyAsync(h_out, d_out, len * sizeof(double), hipMemcpyDeviceToHost, st));
        if (!defer_host_sync()) OBE_HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

}  // extern "C"
