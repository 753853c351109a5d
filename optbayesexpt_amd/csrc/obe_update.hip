// K2 — Bayes update of the particle weights, plus the model-evaluation wrappers and
// the OptBayesExptNoiseParameter constraint mask.  All HBM-bound streams:
//   pass A  read (D+1) rows, write t = nan_to_num(w * L), block partial sums of t
//   pass B  read t, write w' = nan_to_num(t / sum t), block partial sums of w'^2
//   pass C  one block folds the partials into two scalars
// Algorithmic traffic per particle: 8(D_read + 1) + 8 (A) + 16 (B) bytes.
// Reductions are fixed-order (no float atomics) so results are run-to-run identical.
// Measured alternative (round 1): folding the partials inside pass A / pass B by the last
// workgroup to arrive (atomicInc ticket) removes pass C and the per-block re-fold of pass B,
// but 2048 same-address device-scope atomics cost ~12 ns each across the 8 XCDs and an
// agent-scope release fence per workgroup writes back the 8 MB of weights just dirtied:
// 62-89 us per update instead of 21.7 us.  Separate launches win at this size.
// Also measured: all three passes in ONE workgroup for demo-size clouds (bit-identical by
// replaying the virtual workgroups' shuffle trees): 18 us per point at 5 000 particles and
// 100 us at 50 000 (one CU's bandwidth) against 15 us for the three launches at any size up to
// 1M — the launches are ~5 us each and already the floor; not kept.
#include "obe_models.h"
#include "obe_update.h"

namespace obe {

// pass A, model fused
// (round 4, measured and not kept: issuing the loads of 6 particles per thread together before their ~170
// dependent FP64 instructions each — 23.2 vs 23.0 us per update at 1 M particles, 17.4 vs 15.7 at 262 144:
// three waves per SIMD already hide the latency; t stored with __builtin_nontemporal_store so that the kernel
// boundary has less to write back: 21.8 vs 21.8-22.6 us at 1 M particles, 47.1 vs 50.2 us with 10 parameters)
template <class M>
__global__ __launch_bounds__(kBlock) void update_model_kernel(
    obe_model m, SettingArg st, LikArgs la, const double* __restrict__ particles, int64_t ld,
    int64_t n, double* __restrict__ weights, double* __restrict__ partials, SweepCtl ctl) {
    __shared__ double red[kBlock / kWave];
    if (sweep_prologue(ctl, red)) return;
    __syncthreads();
    double acc = 0.0;
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        double y[M::NC];
        M::eval(st.x, ParamRef{particles + p, ld}, m, y);
        const double t = nan_to_num(weights[p] * likelihood_of(y, la, particles, ld, p));
        weights[p] = t;
        acc += t;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// ---- both passes in ONE launch (round 5: built, measured, NOT in the product build) ----------------------------
// -DOBE_ONE_PASS_UPDATE compiles it in (tools/build_variant.py onepass -DOBE_ONE_PASS_UPDATE, then
// OBE_VARIANT=onepass OBE_FIRST_MOM_PER_CU=3 python tools/measure_onepass.py).  Measured on MI355X
// (profiles/r05_update_moments.txt): bit-identical to the two launches, and no faster — 1 048 576 particles,
// D = 3: 24.2-24.5 us against 22.1 us (25.1 us with the first moments on the 768-workgroup grid this form
// needs); 524 288 particles, D = 10: 39.2 us against ~33 us.  Half the bytes move, but the two launches are
// already bandwidth-bound halves that overlap arithmetic with streaming, while here the latencies line up one
// behind the other: all loads (2 us), six particles' likelihoods per thread (~170 dependent FP64 instructions
// each for NumPy-identical divisions and exp: ~5 us), a grid barrier of two device-scope tickets plus the
// release word across 8 XCDs (~5 us: a launch of 20 workgroups takes 10.2 us against 9.5 us for the two
// launches), the normalisation, and the same ticket + fold tail as before.  The bar was 18 us.
#ifdef OBE_ONE_PASS_UPDATE
// A grid barrier between the likelihood pass and the normalisation: every workgroup of a launch that is known
// to be co-resident (the host checks the occupancy before it chooses this form) takes a two-level arrival
// ticket (arrive_last's counters, which wrap back to zero) and then waits for the last arrival to bump a
// generation word next to the top counter.  The word is read BEFORE the ticket is taken, so a workgroup
// can never miss its own barrier's bump.
constexpr int kGenerationWord = kArriveStride * kArriveGroups + 16;
__device__ __forceinline__ void grid_barrier(unsigned* counter, int* flag) {
    unsigned* gen_word = counter + kGenerationWord;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's published partials have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(gen_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned g = blockIdx.x & (kArriveGroups - 1);
        const unsigned in_group = (gridDim.x - g + kArriveGroups - 1) / kArriveGroups;
        const unsigned groups = gridDim.x < (unsigned)kArriveGroups ? gridDim.x : (unsigned)kArriveGroups;
        int last = atomicInc(counter + g * kArriveStride, in_group - 1) == in_group - 1;
        if (last) last = atomicInc(counter + kArriveGroups * kArriveStride, groups - 1) == groups - 1;
        if (last) {
            __hip_atomic_store(gen_word, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen)
                __builtin_amdgcn_s_sleep(2);
        }
        *flag = last;
    }
    __syncthreads();
}

// update_model_kernel + normalize_moments_kernel<D, true> in one launch: a thread keeps its <= PPT particles
// (D rows each) and their t = nan_to_num(w L) in registers across the grid barrier, so the cloud is read once
// and t never travels: 8 (D + 1) N bytes read + 8 N written instead of twice that.  Same grid for both halves
// (the update's: first_moment_blocks() must agree — the host checks), same per-thread order of accumulation,
// same block reductions, same folds: the weights and the K3 block are the two-launch form's, bit for bit.
template <class M, int D, int PPT>
__global__ __launch_bounds__(kBlock) void update_moments_onepass_kernel(
    obe_model m, SettingArg st, LikArgs la, const double* __restrict__ particles, int64_t ld, int64_t n,
    double* __restrict__ weights, double* partials_t, double* partials_mom, UpdateFold fold) {
    __shared__ double red[kBlock / kWave];
    __shared__ int flag;
    double x[PPT][D], t[PPT];
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t p0 = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    double acc = 0.0;
    // every load of the thread's particles is issued before the first likelihood is evaluated (PPT (D + 1)
    // loads in flight per lane: the ~170 dependent FP64 instructions of one particle no longer sit between
    // two round trips to HBM); slots past the end of the cloud re-read the last particle and are not used
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int64_t p = p0 + j * stride;
        const int64_t q = p < n ? p : n - 1;
#pragma unroll
        for (int i = 0; i < D; ++i) x[j][i] = particles[(int64_t)i * ld + q];
        t[j] = weights[q];
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int64_t p = p0 + j * stride;
        if (p < n) {
            double y[M::NC];
            M::eval(st.x, ParamRef{&x[j][0], 1}, m, y);
            const double tt = nan_to_num(t[j] * likelihood_of(y, la, particles, ld, p));
            t[j] = tt;
            acc += tt;
        }
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) store_published(partials_t + blockIdx.x, s);
    grid_barrier(fold.counter, &flag);
    // the total, folded by every workgroup in block_sum_array's order (the partials were stored write-through
    // by other workgroups of this launch: read past this CU's L1)
    double tv = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock) tv += load_published_f64(partials_t + i);
    const double total = block_sum_all(tv, red);
    constexpr int NV = 3 + 2 * D;
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
    double acc2 = 0.0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int64_t p = p0 + j * stride;
        if (p < n) {
            const double w = nan_to_num(t[j] / total);
            weights[p] = w;
            acc2 += nan_to_num(w * w);
            accumulate_first_moments<D>(reinterpret_cast<double(&)[2 + 2 * D]>(v), w, x[j]);
        }
    }
    v[NV - 1] = acc2;
    publish_and_fold_update<D>(v, total, partials_mom, fold);
}

#endif  // OBE_ONE_PASS_UPDATE

template <class M>
__global__ __launch_bounds__(kBlock) void eval_particles_kernel(obe_model m, SettingArg st,
                                                                const double* __restrict__ particles, int64_t ld,
                                                                int64_t n, double* __restrict__ out, int64_t ld_y) {
    for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += (int64_t)gridDim.x * kBlock) {
        double y[M::NC];
        M::eval(st.x, ParamRef{particles + p, ld}, m, y);
#pragma unroll
        for (int c = 0; c < M::NC; ++c) out[(int64_t)c * ld_y + p] = y[c];
    }
}

struct ParamArg {
    double th[OBE_MAX_DIMS];
};

template <class M>
__global__ __launch_bounds__(kBlock) void eval_settings_kernel(obe_model m, ParamArg pa,
                                                               const double* __restrict__ settings, int64_t ld_s,
                                                               int64_t n, double* __restrict__ out, int64_t ld_y) {
    for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < n; s += (int64_t)gridDim.x * kBlock) {
        double x[M::NS], y[M::NC];
#pragma unroll
        for (int k = 0; k < M::NS; ++k) x[k] = settings[(int64_t)k * ld_s + s];
        M::eval(x, ParamRef{pa.th, 1}, m, y);
#pragma unroll
        for (int c = 0; c < M::NC; ++c) out[(int64_t)c * ld_y + s] = y[c];
    }
}

// which form the calling thread's fused updates take (obe_update_one_pass); default from OBE_UPDATE_ONE_PASS
static thread_local int g_update_one_pass = -1;
static thread_local int g_last_update_form = 0;      // 1: one launch, 2: two launches (diagnostic: obe_update_one_pass(-1))
static bool update_one_pass() {
    if (g_update_one_pass < 0) {
        static const int def = getenv("OBE_UPDATE_ONE_PASS") ? atoi(getenv("OBE_UPDATE_ONE_PASS")) != 0 : OBE_UPDATE_ONE_PASS_DEFAULT;
        g_update_one_pass = def;
    }
    return g_update_one_pass != 0;
}

#ifdef OBE_ONE_PASS_UPDATE
// 0: launched; 1: does not apply here (the caller takes the two launches); else an error code
template <class M, int D>
static int try_onepass(const obe_model& mm, const SettingArg& sa, const LikArgs& la, const double* d_particles,
                       int64_t ld_p, int64_t n, double* d_weights, const UpdateWs& w, int nb, const UpdateFold& fold,
                       hipStream_t st) {
    const int64_t per_thread = (n + (int64_t)nb * kBlock - 1) / ((int64_t)nb * kBlock);
    auto launch = [&](auto ppt_tag) -> int {
        constexpr int PPT = decltype(ppt_tag)::value;
        static int resident = -1;            // workgroups of this kernel the device holds at once
        if (resident < 0) {
            int per_cu = 0, dev = 0, n_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, update_moments_onepass_kernel<M, D, PPT>, kBlock, 0) != hipSuccess ||
                hipGetDevice(&dev) != hipSuccess ||
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
                (void)hipGetLastError();
                resident = 0;
            } else {
                resident = per_cu * n_cu;
            }
        }
        if (resident < nb) return 1;          // the grid barrier needs every workgroup on the chip at once
        update_moments_onepass_kernel<M, D, PPT><<<nb, kBlock, 0, st>>>(mm, sa, la, d_particles, ld_p, n, d_weights,
                                                                        w.pa, w.mom, fold);
        OBE_CHECK_LAUNCH("update_moments_onepass_kernel");
        return 0;
    };
    if (per_thread <= 3) return launch(std::integral_constant<int, 3>{});
    if (per_thread <= 6) return launch(std::integral_constant<int, 6>{});
    return 1;
}

#endif  // OBE_ONE_PASS_UPDATE

}  // namespace obe

using namespace obe;

extern "C" {

int obe_bayes_update_model(const obe_model* m, const double* d_particles, int64_t ld_p, int64_t n_particles,
                           double* d_weights, const double* h_setting, const double* h_y_meas,
                           const double* h_sigma, const int32_t* h_noise_rows, int32_t n_lik_channels,
                           double choke, void* d_ws, int64_t ws_bytes, double* h_out, void* stream) {
    if (!m || !d_particles || !d_weights || n_particles <= 0) return bad_arg("obe_bayes_update_model: bad pointer/size");
    obe_model mm = *m;
    if (int rc = obe_model_validate(&mm)) return rc;
    if (n_lik_channels > mm.n_channels) return bad_arg("n_lik_channels exceeds model channels");
    LikArgs la;
    if (int rc = fill_lik_args(la, h_y_meas, h_sigma, h_noise_rows, n_lik_channels, choke, mm.n_params)) return rc;
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w)) return rc;
    SettingArg sa{};
    for (int k = 0; k < mm.n_setdims; ++k) sa.x[k] = h_setting ? h_setting[k] : 0.0;
    hipStream_t st = as_stream(stream);
    const int nb = update_blocks(n_particles);
    int rc = dispatch_model(mm, [&](auto M) -> int {
        using Model = decltype(M);
        update_model_kernel<Model><<<nb, kBlock, 0, st>>>(mm, sa, la, d_particles, ld_p, n_particles, d_weights, w.pa,
                                                          SweepCtl{});
        OBE_CHECK_LAUNCH("update_model_kernel");
        return 0;
    });
    if (rc) return rc;
    return finish_update(w, nb, n_particles, d_weights, h_out, st);
}

static int update_model_moments(const obe_model* m, const double* d_particles, int64_t ld_p, int64_t n_particles,
                                double* d_weights, const double* h_setting, const double* h_y_meas,
                                const double* h_sigma, const int32_t* h_noise_rows, int32_t n_lik_channels,
                                double choke, double* d_moments, void* d_ws, int64_t ws_bytes, double* h_out,
                                void* stream, bool enqueue_only, int32_t auto_resample, double resample_threshold) {
    if (!m || !d_particles || !d_weights || !d_moments || n_particles <= 0)
        return bad_arg("obe_bayes_update_model_moments: bad pointer/size");
    obe_model mm = *m;
    if (int rc = obe_model_validate(&mm)) return rc;
    if (n_lik_channels > mm.n_channels) return bad_arg("n_lik_channels exceeds model channels");
    const int d = mm.n_params;
    if (d < 1 || d > kFastDims) return bad_arg("obe_bayes_update_model_moments: n_params must be 1..16 (OBE_FAST_DIMS)");
    LikArgs la;
    if (int rc = fill_lik_args(la, h_y_meas, h_sigma, h_noise_rows, n_lik_channels, choke, mm.n_params)) return rc;
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w, d)) return rc;
    SettingArg sa{};
    for (int k = 0; k < mm.n_setdims; ++k) sa.x[k] = h_setting ? h_setting[k] : 0.0;
    hipStream_t st = as_stream(stream);
    const int nb = update_blocks(n_particles);
    // Everything that can refuse the call is checked BEFORE the first launch (pass A multiplies the weights by
    // the likelihood in place: a refusal after it would leave them half updated, and a caller that then falls
    // back to the synchronous form would apply the likelihood twice).
    double* hv = static_cast<double*>(device_view_of_host(h_out));
    const int64_t n_words = 2 + 2 + 4 * (int64_t)d + (enqueue_only ? 1 : 0);
    if (enqueue_only && !hv) return bad_arg("obe_bayes_update_model_moments_enqueue: h_out must be page-locked");
    // the fold rides in the normalisation launch (its last workgroup to arrive) unless there is no counter
    // for this stream or OBE_UPDATE_FOLD=separate asks for the round-3 shape (A/B measurements)
    static const bool separate = getenv("OBE_UPDATE_FOLD") && !strcmp(getenv("OBE_UPDATE_FOLD"), "separate");
    unsigned* counter = separate && !enqueue_only ? nullptr : stream_control_words(st);
    if (enqueue_only && !counter) return bad_arg("obe_bayes_update_model_moments_enqueue: no control words for this stream");
    if (enqueue_only && ws_bytes < update_ws_bytes(d) + 16)
        return bad_arg("obe_bayes_update_model_moments_enqueue: the workspace needs 16 spare bytes at its end (OBE_WS_ABORT_WORD)");
    const int nm = first_moment_blocks(n_particles, d);
    if (hv) arm_host_words(h_out, n_words);      // every word of the result block is watched
    const UpdateFold fold{counter, w.scalars, d_moments, hv, enqueue_only ? ws_abort_word(d_ws, ws_bytes) : nullptr,
                          (double)n_particles, resample_threshold, auto_resample};
    // One launch for both passes (obe_update_one_pass(1) / OBE_UPDATE_ONE_PASS=1) where it applies: an arrival
    // counter for the stream, one grid for the likelihood pass and the first moments, a cloud whose threads hold
    // at most 6 particles each, D = the model's own parameters (+ 1 noise row), and a grid that is co-resident.
    // Anything else: the two launches below, which give the same bits.
    bool launched = false;
#ifdef OBE_ONE_PASS_UPDATE
    if (update_one_pass() && counter && nm == nb) {
        const int rc1 = dispatch_model(mm, [&](auto M) -> int {
            using Model = decltype(M);
            if (d == Model::NREAD)
                return try_onepass<Model, Model::NREAD>(mm, sa, la, d_particles, ld_p, n_particles, d_weights, w, nb, fold, st);
            if constexpr (Model::NREAD + 1 <= kFastDims) {
                if (d == Model::NREAD + 1)
                    return try_onepass<Model, Model::NREAD + 1>(mm, sa, la, d_particles, ld_p, n_particles, d_weights, w,
                                                                 nb, fold, st);
            }
            return 1;
        });
        if (rc1 != 0 && rc1 != 1) return rc1;
        launched = rc1 == 0;
    }
#endif
    g_last_update_form = launched ? 1 : 2;
    if (launched) goto delivered;
    {
    int rc = dispatch_model(mm, [&](auto M) -> int {
        using Model = decltype(M);
        update_model_kernel<Model><<<nb, kBlock, 0, st>>>(mm, sa, la, d_particles, ld_p, n_particles, d_weights, w.pa,
                                                          SweepCtl{});
        OBE_CHECK_LAUNCH("update_model_kernel");
        return 0;
    });
    if (rc) return rc;
    }
    if (int rc2 = launch_normalize_moments(d, w, nb, nm, d_particles, ld_p, n_particles, d_weights, fold, d_moments, hv, st))
        return rc2;
delivered:
    if (enqueue_only) return 0;
    if (h_out) {
        if (hv) return wait_host_words(h_out, n_words, st);
        {
            OBE_HIP_TRY(hipMemcpyAsync(h_out, w.scalars, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            OBE_HIP_TRY(hipMemcpyAsync(h_out + 2, d_moments, (2 + 4 * (int64_t)d) * sizeof(double),
                                       hipMemcpyDeviceToHost, st));
        }
        OBE_HIP_TRY(hipStreamSynchronize(st));
    }
    return 0;
}

int obe_update_one_pass(int32_t on) {
    const int prev = update_one_pass() ? 1 : 0;
    if (on < 0) return g_last_update_form;        // what the calling thread's last fused update actually did
    g_update_one_pass = on != 0;
    return prev;
}

int obe_bayes_update_model_moments(const obe_model* m, const double* d_particles, int64_t ld_p, int64_t n_particles,
                                   double* d_weights, const double* h_setting, const double* h_y_meas,
                                   const double* h_sigma, const int32_t* h_noise_rows, int32_t n_lik_channels,
                                   double choke, double* d_moments, void* d_ws, int64_t ws_bytes, double* h_out,
                                   void* stream) {
    return update_model_moments(m, d_particles, ld_p, n_particles, d_weights, h_setting, h_y_meas, h_sigma,
                                h_noise_rows, n_lik_channels, choke, d_moments, d_ws, ws_bytes, h_out, stream, false, 0,
                                0.0);
}

int obe_bayes_update_model_moments_enqueue(const obe_model* m, const double* d_particles, int64_t ld_p,
                                           int64_t n_particles, double* d_weights, const double* h_setting,
                                           const double* h_y_meas, const double* h_sigma,
                                           const int32_t* h_noise_rows, int32_t n_lik_channels, double choke,
                                           double* d_moments, void* d_ws, int64_t ws_bytes, double* h_pinned_out,
                                           int32_t auto_resample, double resample_threshold, void* stream) {
    if (!h_pinned_out) return bad_arg("obe_bayes_update_model_moments_enqueue: h_pinned_out is NULL");
    return update_model_moments(m, d_particles, ld_p, n_particles, d_weights, h_setting, h_y_meas, h_sigma,
                                h_noise_rows, n_lik_channels, choke, d_moments, d_ws, ws_bytes, h_pinned_out, stream,
                                true, auto_resample, resample_threshold);
}

int obe_bayes_update_sweep(const obe_model* m, const double* d_particles, int64_t ld_p, int64_t n_particles,
                           double* d_weights, const double* h_settings, const double* h_y_meas,
                           const double* h_sigma, const int32_t* h_noise_rows, int32_t n_lik_channels,
                           double choke, int64_t n_points, int32_t auto_resample, double resample_threshold,
                           void* d_ws, int64_t ws_bytes, double* h_out, void* stream) {
    if (!m || !d_particles || !d_weights || n_particles <= 0 || n_points <= 0 || !h_y_meas || !h_out)
        return bad_arg("obe_bayes_update_sweep: bad pointer/size");
    obe_model mm = *m;
    if (int rc = obe_model_validate(&mm)) return rc;
    if (n_lik_channels > mm.n_channels) return bad_arg("n_lik_channels exceeds model channels");
    UpdateWs w;
    if (int rc = carve_update_ws(d_ws, ws_bytes, w)) return rc;
    hipStream_t st = as_stream(stream);
    const int nb = update_blocks(n_particles);
    if (int rc = launch_sweep_reset(w, st)) return rc;
    // (obe_strict_sums: every point's sum t and sum w'^2 in np.sum's order, as the point-by-point calls form them)
    const bool strict = strict_sums_on();
    const int nfold = strict ? 1 : nb;
    for (int64_t k = 0; k < n_points; ++k) {
        LikArgs la;
        if (int rc = fill_lik_args(la, h_y_meas + k * OBE_MAX_CHANNELS, h_sigma, h_noise_rows, n_lik_channels, choke,
                                   mm.n_params))
            return rc;
        SettingArg sa{};
        for (int j = 0; j < mm.n_setdims; ++j) sa.x[j] = h_settings ? h_settings[k * OBE_MAX_SETDIMS + j] : 0.0;
        const SweepCtl ctl{w.scalars, w.pa, w.pb, nfold, (int)k, auto_resample, resample_threshold, (double)n_particles};
        int rc = dispatch_model(mm, [&](auto M) -> int {
            using Model = decltype(M);
            update_model_kernel<Model><<<nb, kBlock, 0, st>>>(mm, sa, la, d_particles, ld_p, n_particles, d_weights,
                                                              w.pa, ctl);
            OBE_CHECK_LAUNCH("update_model_kernel");
            return 0;
        });
        if (rc) return rc;
        if (int rc2 = launch_sweep_point_tail(w, nb, nfold, n_particles, d_weights, strict, st)) return rc2;
    }
    if (int rc = launch_sweep_end(w, nfold, n_particles, auto_resample, resample_threshold, (int)n_points, st)) return rc;
    OBE_HIP_TRY(hipMemcpyAsync(h_out, w.scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    OBE_HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int obe_eval_over_particles(const obe_model* m, const double* d_particles, int64_t ld_p, int64_t n_particles,
                            const double* h_setting, double* d_y_out, int64_t ld_y, void* stream) {
    if (!m || !d_particles || !d_y_out || n_particles <= 0) return bad_arg("obe_eval_over_particles: bad pointer/size");
    obe_model mm = *m;
    if (int rc = obe_model_validate(&mm)) return rc;
    SettingArg sa{};
    for (int k = 0; k < mm.n_setdims; ++k) sa.x[k] = h_setting ? h_setting[k] : 0.0;
    hipStream_t st = as_stream(stream);
    return dispatch_model(mm, [&](auto M) -> int {
        using Model = decltype(M);
        eval_particles_kernel<Model><<<stream_blocks(n_particles, kBlock), kBlock, 0, st>>>(
            mm, sa, d_particles, ld_p, n_particles, d_y_out, ld_y);
        OBE_CHECK_LAUNCH("eval_particles_kernel");
        return 0;
    });
}

int obe_eval_over_settings(const obe_model* m, const double* d_settings, int64_t ld_s, int64_t n_settings,
                           const double* h_params, double* d_y_out, int64_t ld_y, void* stream) {
    if (!m || !d_settings || !d_y_out || !h_params || n_settings <= 0) return bad_arg("obe_eval_over_settings: bad pointer/size");
    obe_model mm = *m;
    if (int rc = obe_model_validate(&mm)) return rc;
    ParamArg pa{};
    for (int i = 0; i < mm.n_params && i < OBE_MAX_DIMS; ++i) pa.th[i] = h_params[i];
    hipStream_t st = as_stream(stream);
    return dispatch_model(mm, [&](auto M) -> int {
        using Model = decltype(M);
        eval_settings_kernel<Model><<<stream_blocks(n_settings, kBlock), kBlock, 0, st>>>(
            mm, pa, d_settings, ld_s, n_settings, d_y_out, ld_y);
        OBE_CHECK_LAUNCH("eval_settings_kernel");
        return 0;
    });
}

}  // extern "C"
