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
// (Round 4, measured and not kept: folding the block partials inside these kernels by the last workgroup to
// arrive — write-through partials, arrival counter — instead of the one-workgroup fold launches below.  At
// D = 10 the covariance pass took 32 us against 13.8 + 8.4 us: a 4-wave workgroup reads the 256 x 55
// partials more slowly than the 16-wave fold kernel, and every workgroup drains its write-through stores
// before it can take its ticket.  The update's normalisation pass, whose fold is 8-22 values, keeps it.)
template <int D>
__global__ __launch_bounds__(kMomThreads) void moments_pass1(const double* __restrict__ x, int64_t ld, int64_t n,
                                                        const double* __restrict__ w,
                                                        double* __restrict__ partials) {
    double v[2 + 2 * D];
#pragma unroll
    for (int k = 0; k < 2 + 2 * D; ++k) v[k] = 0.0;
    // OBE_MOM_UNROLL particles per trip (a compile-time tuning aid, 1 in the product build: with 256
    // workgroups of waves already keeping (D + 1) loads each in flight, two per trip measured no
    // faster); any value adds in the same order per accumulator as one at a time
    const int64_t stride = (int64_t)gridDim.x * kMomThreads;
    for (int64_t p = (int64_t)blockIdx.x * kMomThreads + threadIdx.x; p < n; p += OBE_MOM_UNROLL * stride) {
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
    store_block_partials<2 + 2 * D, false, kMomThreads>(v, partials);
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

// fold + derive in one single-workgroup launch
// out: [0]=W [1]=W2 [2..) mean [2+D..) m1 [2+2D..) m2 [2+3D..) std
// (host: the device view of the caller's page-locked h_out, or NULL — then obe_moments copies)
__global__ __launch_bounds__(kFoldThreads) void fold_derive_pass1(const double* __restrict__ partials, int nb, int d,
                                                            MomentsOut mo) {
    __shared__ double raw[kMaxMomentValues];
    fold_values_block(partials, nb, 2 + 2 * d, raw);
    derive_first_moments(raw, d, mo.out, mo.host);
}

__global__ __launch_bounds__(kFoldThreads) void fold_derive_pass2(const double* __restrict__ partials, int nb, int d,
                                                            MomentsOut mo) {
    __shared__ double raw[kMaxMomentValues];
    fold_values_block(partials, nb, d * (d + 1) / 2, raw);
    derive_covariance(raw, d, mo.out, mo.host);
}

// ---- wide clouds (D > OBE_FAST_DIMS): the same sums, tiled over kTile rows at a time ---------------------------
// The reference takes any number of parameters (particlepdf.py:105).  Beyond the widths the kernels above are
// compiled for, a launch handles one tile of 8 rows (first moments) or one pair of tiles (covariance: an 8 x 8
// block of sum w (x_i - mu_i)(x_j - mu_j)), each a pass over the cloud that reads only its rows: 2 ceil(D/8)
// launches for the first moments, T (T + 1) launches for the covariance, T = ceil(D/8) — slower than one
// pass with everything in registers, and any D.  Per-particle arithmetic, block reductions and folds are the ones
// above (accumulate_first_moments, store_block_partials, fold_values_block): a fixed order, run-to-run identical.
constexpr int kTile = 8;

// first moments of rows [d0, d0 + kTile): v[0] = sum w, v[1] = sum w^2 (every tile; tile 0's are used),
// v[2 + i] = sum x w, v[2 + kTile + i] = sum x^2 w   (rows past the end: zeros)
__global__ __launch_bounds__(kBlock) void moments_pass1_tile(const double* __restrict__ x, int64_t ld, int64_t n,
                                                             const double* __restrict__ w, int d, int d0,
                                                             double* __restrict__ partials) {
    double v[2 + 2 * kTile];
#pragma unroll
    for (int k = 0; k < 2 + 2 * kTile; ++k) v[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride) {
        const double wp = w[p];
        double xi[kTile];
#pragma unroll
        for (int i = 0; i < kTile; ++i) xi[i] = d0 + i < d ? x[(int64_t)(d0 + i) * ld + p] : 0.0;
        accumulate_first_moments<kTile>(v, wp, xi);
    }
    store_block_partials<2 + 2 * kTile>(v, partials);
}

__global__ __launch_bounds__(kFoldThreads) void fold_derive_pass1_tile(const double* __restrict__ partials, int nb,
                                                                       int d, int d0, MomentsOut mo) {
    __shared__ double raw[2 + 2 * kTile];
    fold_values_block(partials, nb, 2 + 2 * kTile, raw);
    const int i = threadIdx.x;
    double* __restrict__ out = mo.out;
    double* __restrict__ host = mo.host;
    // (the sums of the weights come from tile 0 and are in place before the other tiles' launches read them)
    if (i == 0 && d0 == 0) {
        out[0] = raw[0];
        out[1] = raw[1];
        if (host) {
            host[0] = raw[0];
            host[1] = raw[1];
        }
    }
    if (i < kTile && d0 + i < d) {
        const int r = d0 + i;
        const double m1 = raw[2 + i], m2 = raw[2 + kTile + i];
        const double mean = m1 / raw[0];             // np.average: sum(x w) / sum(w)
        const double sd = sqrt(m2 - m1 * m1);        // particlepdf.py:211-214
        out[2 + r] = mean;
        out[2 + d + r] = m1;
        out[2 + 2 * d + r] = m2;
        out[2 + 3 * d + r] = sd;
        if (host) {
            host[2 + r] = mean;
            host[2 + d + r] = m1;
            host[2 + 2 * d + r] = m2;
            host[2 + 3 * d + r] = sd;
        }
    }
}

// the kTile x kTile block (rows i0.., columns j0..) of sum (x_i - mu_i) * ((x_j - mu_j) * w)
__global__ __launch_bounds__(kBlock) void moments_pass2_tile(const double* __restrict__ x, int64_t ld, int64_t n,
                                                             const double* __restrict__ w,
                                                             const double* __restrict__ out /* mean at out + 2 */,
                                                             int d, int i0, int j0, double* __restrict__ partials) {
    constexpr int NV = kTile * kTile;
    double v[NV], mi[kTile], mj[kTile];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
#pragma unroll
    for (int i = 0; i < kTile; ++i) {
        mi[i] = i0 + i < d ? out[2 + i0 + i] : 0.0;
        mj[i] = j0 + i < d ? out[2 + j0 + i] : 0.0;
    }
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride) {
        const double wp = w[p];
        double di[kTile], dj[kTile];
#pragma unroll
        for (int i = 0; i < kTile; ++i) {
            di[i] = (i0 + i < d ? x[(int64_t)(i0 + i) * ld + p] : 0.0) - mi[i];
            dj[i] = (j0 + i < d ? x[(int64_t)(j0 + i) * ld + p] : 0.0) - mj[i];
        }
#pragma unroll
        for (int i = 0; i < kTile; ++i)
#pragma unroll
            for (int j = 0; j < kTile; ++j) v[i * kTile + j] = fma(di[i], dj[j] * wp, v[i * kTile + j]);
    }
    store_block_partials<NV>(v, partials);
}

// cov = S * (1 / (W - W2/W)), this block and its mirror image
__global__ __launch_bounds__(kFoldThreads) void fold_derive_pass2_tile(const double* __restrict__ partials, int nb,
                                                                       int d, int i0, int j0, MomentsOut mo) {
    __shared__ double raw[kTile * kTile];
    fold_values_block(partials, nb, kTile * kTile, raw);
    double* __restrict__ out = mo.out;
    const double fact = out[0] - out[1] / out[0];
    const double scale = 1.0 / fact;
    double* cov = out + 2 + 4 * d;
    const int e = threadIdx.x;
    if (e < kTile * kTile) {
        const int i = i0 + e / kTile, j = j0 + e % kTile;
        if (i < d && j < d && (i0 != j0 || i <= j)) {
            const double c = raw[e] * scale;
            cov[i * d + j] = c;
            cov[j * d + i] = c;
            if (mo.host) {
                mo.host[2 + 4 * d + i * d + j] = c;
                mo.host[2 + 4 * d + j * d + i] = c;
            }
        }
    }
}

static int launch_moments_tiled(const double* x, int64_t ld, int d, int64_t n, const double* w, int want_cov,
                                double* partials, const MomentsOut& mo, hipStream_t st) {
    const int nb = moment_blocks(n, kTile);
    const int tiles = (d + kTile - 1) / kTile;
    if (want_cov != 2) {
        for (int t = 0; t < tiles; ++t) {
            moments_pass1_tile<<<nb, kBlock, 0, st>>>(x, ld, n, w, d, t * kTile, partials);
            OBE_CHECK_LAUNCH("moments_pass1_tile");
            fold_derive_pass1_tile<<<1, kFoldThreads, 0, st>>>(partials, nb, d, t * kTile, mo);
            OBE_CHECK_LAUNCH("fold_derive_pass1_tile");
        }
    }
    if (want_cov) {
        for (int ti = 0; ti < tiles; ++ti) {
            for (int tj = ti; tj < tiles; ++tj) {
                moments_pass2_tile<<<nb, kBlock, 0, st>>>(x, ld, n, w, mo.out, d, ti * kTile, tj * kTile, partials);
                OBE_CHECK_LAUNCH("moments_pass2_tile");
                fold_derive_pass2_tile<<<1, kFoldThreads, 0, st>>>(partials, nb, d, ti * kTile, tj * kTile, mo);
                OBE_CHECK_LAUNCH("fold_derive_pass2_tile");
            }
        }
    }
    return 0;
}

template <int D>
static int launch_moments(const double* x, int64_t ld, int64_t n, const double* w, int want_cov, double* partials,
                          const MomentsOut& mo, hipStream_t st) {
    const int nb = moment_blocks(n, D);
    if (want_cov != 2) {            // (2: `out` already holds the first moments of these weights)
        const int nb1 = first_moment_blocks(n, D);
        moments_pass1<D><<<nb1, kMomThreads, 0, st>>>(x, ld, n, w, partials);
        OBE_CHECK_LAUNCH("moments_pass1");
        fold_derive_pass1<<<1, kFoldThreads, 0, st>>>(partials, nb1, D, mo);
        OBE_CHECK_LAUNCH("fold_derive_pass1");
    }
    if (want_cov) {
        moments_pass2<D><<<nb, kBlock, 0, st>>>(x, ld, n, w, mo.out, partials);
        OBE_CHECK_LAUNCH("moments_pass2");
        fold_derive_pass2<<<1, kFoldThreads, 0, st>>>(partials, nb, D, mo);
        OBE_CHECK_LAUNCH("fold_derive_pass2");
    }
    return 0;
}

// obe_moments without the final synchronisation (obe_resample_begin arms and watches the host words)
int moments_call(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles, const double* d_weights,
                 int32_t want_cov, double* d_out, double* h_out, void* d_ws, int64_t ws_bytes, hipStream_t st,
                 bool* host_written) {
    if (!d_particles || !d_weights || !d_out || n_particles <= 0) return bad_arg("obe_moments: bad pointer/size");
    if (n_dims < 1 || n_dims > OBE_CLOUD_MAX_DIMS) return bad_arg("obe_moments: n_dims must be 1..1024");
    const bool tiled = n_dims > kFastDims;
    const int64_t nv_max = tiled ? kTile * kTile : std::max<int64_t>(2 + 2 * n_dims, (int64_t)n_dims * (n_dims + 1) / 2);
    const int64_t need = ((int64_t)kMomGridCap * nv_max + nv_max) * sizeof(double);
    if (!d_ws || ws_bytes < need) return bad_arg("obe_moments: workspace too small");
    double* partials = static_cast<double*>(d_ws);
    double* hv = static_cast<double*>(device_view_of_host(h_out));     // page-locked h_out: the kernels write it
    const MomentsOut mo{d_out, hv};
    if (host_written) *host_written = hv != nullptr;
    if (tiled) return launch_moments_tiled(d_particles, ld_p, n_dims, n_particles, d_weights, want_cov, partials, mo, st);
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
    if (int rc = moments_call(d_particles, ld_p, n_dims, n_particles, d_weights, want_cov, d_out, h_out, d_ws, ws_bytes,
                              st, &host_written))
        return rc;
    if (h_out) {
        const int64_t len = want_cov ? obe_moments_len(n_dims) : 2 + 4 * (int64_t)n_dims;
        if (!host_written) OBE_HIP_TRY(hipMemcpyAsync(h_out, d_out, len * sizeof(double), hipMemcpyDeviceToHost, st));
        if (!defer_host_sync()) OBE_HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

}  // extern "C"
