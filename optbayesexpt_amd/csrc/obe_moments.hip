// K3 — weighted mean / std / covariance of the particle cloud (particlepdf.py:173-214).
// HBM-bound: pass 1 reads (D+1) rows once (8(D+1) B per particle) and produces
// sum w, sum w^2, sum w x_i, sum w x_i^2; pass 2 (covariance only, i.e. only on
// resample cycles) reads them again for sum w (x_i - mu_i)(x_j - mu_j).
// All accumulators live in registers (template on D), partials are folded by one
// wavefront per output value in a fixed order.
#include <type_traits>

#include "obe_common.h"

namespace obe {

#ifndef OBE_MOM_BLOCKS
#define OBE_MOM_BLOCKS 256
#endif
// grid cap for the moment passes: one workgroup per CU.  The streaming itself is as fast with 1024
// (12-15 us at D = 10, 524 288 particles either way), but every workgroup ends with a block
// reduction of up to 136 values and leaves a row of partials for the single-workgroup fold.
constexpr int kMomBlocks = OBE_MOM_BLOCKS;
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

template <int NV>
__device__ __forceinline__ void store_block_partials(double (&v)[NV], double* __restrict__ partials) {
    constexpr int NW = kBlock / kWave;
    __shared__ double red[NW][NV];
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    reduce_value_groups<NV, 0>(v, red[wid], lane);
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
            if (p + u * stride < n) {
                v[0] += wp[u];
                v[1] += wp[u] * wp[u];
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    v[2 + i] += xi[u][i] * wp[u];
                    v[2 + D + i] += (xi[u][i] * xi[u][i]) * wp[u];
                }
            }
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
    store_block_partials<NV>(v, partials);
}

// raw[k] = sum_b partials[b*nv + k]: one wavefront per value (strided partial sums, then the
// shuffle tree: a fixed order).  One workgroup of 16 waves; a wave takes four values at a time and
// issues their loads together — the partials come straight from HBM / another XCD's L2, and a wave
// that walked its values one after the other spent ~1.5 us of latency on each (19 us for the 55
// values of a 10-parameter covariance).  `vals` in LDS.
constexpr int kFoldThreads = 1024;
constexpr int kFoldBatch = 4;
__device__ __forceinline__ void fold_values_block(const double* __restrict__ partials, int nb, int nv,
                                                  double* __restrict__ vals) {
    constexpr int NW = kFoldThreads / kWave;
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    for (int k0 = wid; k0 < nv; k0 += NW * kFoldBatch) {
        double s[kFoldBatch];
#pragma unroll
        for (int u = 0; u < kFoldBatch; ++u) s[u] = 0.0;
        for (int b = lane; b < nb; b += kWave) {
#pragma unroll
            for (int u = 0; u < kFoldBatch; ++u) {
                const int k = k0 + u * NW;
                if (k < nv) s[u] += partials[(int64_t)b * nv + k];
            }
        }
#pragma unroll
        for (int u = 0; u < kFoldBatch; ++u) s[u] = wave_sum(s[u]);
#pragma unroll
        for (int u = 0; u < kFoldBatch; ++u) {
            const int k = k0 + u * NW;
            if (lane == 0 && k < nv) vals[k] = s[u];
        }
    }
    __syncthreads();
}

constexpr int kMaxMomentValues = OBE_MAX_DIMS * (OBE_MAX_DIMS + 1) / 2;     // >= 2 + 2 D

// fold + derive in one single-workgroup launch
// out: [0]=W [1]=W2 [2..) mean [2+D..) m1 [2+2D..) m2 [2+3D..) std
// (host: the device view of the caller's page-locked h_out, or NULL — then obe_moments copies)
__global__ __launch_bounds__(kFoldThreads) void fold_derive_pass1(const double* __restrict__ partials, int nb, int d,
                                                            double* __restrict__ out, double* __restrict__ host) {
    __shared__ double raw[kMaxMomentValues];
    fold_values_block(partials, nb, 2 + 2 * d, raw);
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

// cov = S * (1 / (W - W2/W))  (np.cov scales by the reciprocal)
__global__ __launch_bounds__(kFoldThreads) void fold_derive_pass2(const double* __restrict__ partials, int nb, int d,
                                                            double* __restrict__ out, double* __restrict__ host) {
    __shared__ double raw[kMaxMomentValues];
    fold_values_block(partials, nb, d * (d + 1) / 2, raw);
    const double fact = out[0] - out[1] / out[0];
    const double scale = 1.0 / fact;
    double* cov = out + 2 + 4 * d;
    for (int e = threadIdx.x; e < d * d; e += kFoldThreads) {
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

template <int D>
static int launch_moments(const double* x, int64_t ld, int64_t n, const double* w, int want_cov, double* out,
                          double* partials, double* raw, double* host, hipStream_t st) {
    const int nb = static_cast<int>(std::min<int64_t>(kMomBlocks, (n + kBlock - 1) / kBlock));
    constexpr int NV1 = 2 + 2 * D;
    moments_pass1<D><<<nb, kBlock, 0, st>>>(x, ld, n, w, partials);
    OBE_CHECK_LAUNCH("moments_pass1");
    fold_derive_pass1<<<1, kFoldThreads, 0, st>>>(partials, nb, D, out, host);
    OBE_CHECK_LAUNCH("fold_derive_pass1");
    if (want_cov) {
        moments_pass2<D><<<nb, kBlock, 0, st>>>(x, ld, n, w, out, partials);
        OBE_CHECK_LAUNCH("moments_pass2");
        fold_derive_pass2<<<1, kFoldThreads, 0, st>>>(partials, nb, D, out, host);
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
    double* hv = static_cast<double*>(device_view_of_host(h_out));     // page-locked h_out: the kernels write it
    int rc = -1;
#define OBE_MOM_CASE(DD) \
    case DD: rc = launch_moments<DD>(d_particles, ld_p, n_particles, d_weights, want_cov, d_out, partials, raw, hv, st); break;
    switch (n_dims) {
        OBE_MOM_CASE(1) OBE_MOM_CASE(2) OBE_MOM_CASE(3) OBE_MOM_CASE(4) OBE_MOM_CASE(5) OBE_MOM_CASE(6)
        OBE_MOM_CASE(7) OBE_MOM_CASE(8) OBE_MOM_CASE(9) OBE_MOM_CASE(10) OBE_MOM_CASE(11) OBE_MOM_CASE(12)
        OBE_MOM_CASE(13) OBE_MOM_CASE(14) OBE_MOM_CASE(15) OBE_MOM_CASE(16)
    }
#undef OBE_MOM_CASE
    if (rc) return rc;
    if (h_out) {
        const int64_t len = want_cov ? obe_moments_len(n_dims) : 2 + 4 * (int64_t)n_dims;
        if (!hv) OBE_HIP_TRY(hipMemcpyAsync(h_out, d_out, len * sizeof(double), hipMemcpyDeviceToHost, st));
        if (!defer_host_sync()) OBE_HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

}  // extern "C"
