// K2 — declarations shared by the two halves of the Bayes update (round 6):
//   obe_update.hip         what depends on the MODEL: pass A with the model fused (update_model_kernel<M>), the model
//                          evaluation wrappers, and the entry points that launch them;
//   obe_update_common.hip  everything else: normalisation, the fused first moments and their folds, the np.sum-ordered
//                          sums, the constraint masks, the y / likelihood forms of the update.
// A per-model plugin library compiles only the first (and the sweep / y-space sources) and links the second as the
// object the library build left behind (optbayesexpt_amd/build.py): its 40-odd kernel instantiations do not depend on
// the model, and recompiling them was 9 of the 10 seconds a user waited for a formula's first use.
#pragma once

#include <cstdlib>
#include <cstring>

#include "obe_common.h"
#include "obe_moments.h"

#ifndef OBE_NORM_UNROLL
#define OBE_NORM_UNROLL 2
#endif
#ifndef OBE_UPDATE_ONE_PASS_DEFAULT
#define OBE_UPDATE_ONE_PASS_DEFAULT 0
#endif

namespace obe {

struct LikArgs {
    int n_ch;          // channels entering the likelihood product
    int use_rows;      // sigma from particle rows (NoiseParameter) instead of sigma[]
    int use_choke;
    int noise_rows[OBE_MAX_CHANNELS];
    double y_meas[OBE_MAX_CHANNELS];
    double sigma[OBE_MAX_CHANNELS];
    double choke;
};

// obe_base.py:269-271 and :451-461 (obe_noiseparam.py:109-120 for use_rows)
__device__ __forceinline__ double likelihood_of(const double* y, const LikArgs& a,
                                                const double* particles, int64_t ld, int64_t p) {
    double lk = 1.0;
    for (int ch = 0; ch < a.n_ch; ++ch) {
        const double s = a.use_rows ? particles[(int64_t)a.noise_rows[ch] * ld + p] : a.sigma[ch];
        const double z = (y[ch] - a.y_meas[ch]) / s;
        const double e = exp(-(z * z) / 2.0);
        lk = lk * (e / s);
    }
    if (a.use_choke) lk = pow(lk, a.choke);
    return lk;
}

struct SettingArg {
    double x[OBE_MAX_SETDIMS];
};

// A sweep batch (obe_bayes_update_sweep) runs the resample test of particlepdf.py:236-258 on the
// device, in the prologue of the NEXT point's pass A: every workgroup folds the previous point's
// sum-of-squares partials (the same fixed-order fold as fold2_kernel, so all agree), and if that
// point asked for a resample the whole launch returns — workgroup 0 records it: scalars[0] = sum t,
// [1] = sum w'^2 of the last point applied, [2] = stop flag (sticky: later launches return at
// once), [3] = points applied.  Two launches per point instead of three.
struct SweepCtl {
    double* scalars;           // NULL: a single update, no sweep logic
    const double* pa;          // partial sums of t      (previous point's, until this launch overwrites them)
    const double* pb;          // partial sums of w'^2   (previous point's)
    int nb;
    int point;                 // index of the point this launch applies
    int auto_resample;
    double resample_threshold;
    double n_particles;
};

__device__ __forceinline__ bool resample_due(double sum_w2, double n_particles, double threshold) {
    const double n_eff = 1.0 / sum_w2;
    return n_eff < 0.1 * n_particles || n_eff / n_particles < threshold;
}

// true: this launch must not touch the weights
__device__ __forceinline__ bool sweep_prologue(const SweepCtl& c, double* red) {
    if (!c.scalars) return false;
    if (c.scalars[2] != 0.0) return true;
    if (c.point == 0 || !c.auto_resample) return false;
    const double b = block_sum_array(c.pb, c.nb, red);          // previous point's sum w'^2, in every thread
    if (!resample_due(b, c.n_particles, c.resample_threshold)) return false;
    if (blockIdx.x == 0) {
        __syncthreads();
        const double a = block_sum_array(c.pa, c.nb, red);
        if (threadIdx.x == 0) {
            c.scalars[0] = a;
            c.scalars[1] = b;
            c.scalars[3] = (double)c.point;
            c.scalars[2] = 1.0;
        }
    }
    return true;
}

// pass B fused with K3's first pass (obe_bayes_update_model_moments): normalise by the re-folded
// total and, in the same sweep over the cloud, accumulate the first moments of the NEW weights.
// Grid, per-particle arithmetic and block reductions are those of moments_pass1, so the moments
// are bit-identical to obe_moments() called on the updated weights; the weights themselves are
// the ones normalize_kernel writes.  One launch less per cycle and no second read of the weights
// (every cycle needs the moments: the sweep's shift, mean(), std(), the noise-parameter variance).
// FOLD (round 4): the launch also does what fold_update_moments_kernel did — the workgroup that arrives
// last folds everybody's partials (written through, so no release fence has to write back the weights just
// dirtied; obe_common.h: arrive_last) and delivers {sum t, sum w'^2} and the K3 block.  Same sums in the
// same order as the separate fold: identical bits, one dependent launch (its ~7 us) less per update.
struct UpdateFold {
    unsigned* counter;          // arrival counter of this stream (zero between launches)
    double* scalars;            // [0] sum t, [1] sum w'^2
    double* mom_out;            // K3 block on the device
    double* host_out;           // device view of the caller's page-locked h_out, or NULL
    // enqueue form (obe_bayes_update_model_moments_enqueue): the resample test of particlepdf.py:236-258 on
    // sum w'^2, left in the workspace's abort word (obe_common.h: ws_abort_word) for the sweep that was enqueued behind this update without
    // waiting for it (obe_sweep.hip: OBE_SWEEP_SPECULATIVE), and in host_out[4 + 4 d] as 0.0 / 1.0
    unsigned* abort_out;        // NULL: not the enqueue form
    double n_particles, threshold;
    int auto_resample;
};

// The tail of a launch that normalises and accumulates first moments: every workgroup publishes its row of
// partial sums (v[2 + 2 D] = sum w'^2 rides along), the one that arrives last folds all rows in a fixed order
// and delivers {sum t, sum w'^2}, the K3 block and (enqueue form) the resample decision.  One definition for
// the two-launch and the one-launch update: the same bits.
template <int D, int NT = kBlock>
__device__ __forceinline__ void publish_and_fold_update(double (&v)[3 + 2 * D], double total, double* partials_mom,
                                                        const UpdateFold& fold) {
    constexpr int NV = 3 + 2 * D;
    store_block_partials<NV, true, NT>(v, partials_mom);
    __shared__ int last;
    if (!arrive_last<false>(fold.counter, &last)) return;
    __shared__ double raw[kMaxMomentValues + 1];
    fold_values_block<NT, true, (NV + NT / kWave - 1) / (NT / kWave)>(partials_mom, gridDim.x, NV, raw);
    // delivery by ONE wave: K3 block to the device copy and to the host, one system-scope fence, then the
    // word the host watches (wait_host_word) — [0] sum t, [1] sum w'^2, [2..) K3 block
    if (threadIdx.x < kWave) {
        derive_first_moments(raw, D, fold.mom_out, fold.host_out ? fold.host_out + 2 : nullptr);
        const double b = raw[NV - 1];
        if (threadIdx.x == 0) {
            fold.scalars[0] = total;
            fold.scalars[1] = b;
            if (fold.host_out) fold.host_out[0] = total;
            if (fold.abort_out) {
                const bool due = fold.auto_resample && resample_due(b, fold.n_particles, fold.threshold);
                *fold.abort_out = due ? 1u : 0u;
                if (fold.host_out) fold.host_out[4 + 4 * D] = due ? 1.0 : 0.0;
            }
        }
        if (fold.host_out) {
            host_results_before_flag();     // (measured without it, round 5: no difference — the words are watched one by one anyway)
            if (threadIdx.x == 0) fold.host_out[1] = b;
        }
    }
}


struct UpdateWs {
    double* pa;
    double* pb;
    double* scalars;
    double* mom;        // block partials of the fused first moments (moments_dims > 0 only)
};

// ---- host side, defined in obe_update_common.hip ----------------------------------------------------------------
int fill_lik_args(LikArgs& la, const double* h_y_meas, const double* h_sigma, const int32_t* h_noise_rows,
                  int32_t n_lik_channels, double choke, int n_rows);
int64_t update_ws_bytes(int moments_dims);
int carve_update_ws(void* d_ws, int64_t ws_bytes, UpdateWs& w, int moments_dims = 0);
int update_blocks(int64_t n);
bool strict_sums_on();           // obe_strict_sums: the calling thread's unfused updates sum in np.sum's order
// pass B + pass C of an unfused update (np.sum-ordered when strict_sums_on()), results to h_out
int finish_update(const UpdateWs& w, int nb, int64_t n, double* d_weights, double* h_out, hipStream_t st);
// pass B fused with the first moments (normalize_moments_kernel<d, FOLD>) and, without an arrival counter, the
// separate fold launch: everything of obe_bayes_update_model_moments behind pass A
int launch_normalize_moments(int d, const UpdateWs& w, int nb, int nm, const double* d_particles, int64_t ld_p,
                             int64_t n_particles, double* d_weights, const UpdateFold& fold, double* d_moments, double* hv,
                             hipStream_t st);
// the model-independent launches of a sweep batch (obe_bayes_update_sweep): before the first point, behind every
// point's pass A, behind the last point
int launch_sweep_reset(const UpdateWs& w, hipStream_t st);
int launch_sweep_point_tail(const UpdateWs& w, int nb, int nfold, int64_t n_particles, double* d_weights, bool strict,
                            hipStream_t st);
int launch_sweep_end(const UpdateWs& w, int nfold, int64_t n_particles, int auto_resample, double resample_threshold,
                     int n_points, hipStream_t st);

}  // namespace obe
