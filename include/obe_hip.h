/*
 * obe_hip.h — C ABI of libobe_hip.so: the MI355X (gfx950) hot path of optbayesexpt.
 *
 * The reference (usnistgov/optbayesexpt 1.2.0) is pure Python + NumPy and has no
 * FFI of its own, so every entry point below names the reference *method* whose
 * arithmetic it replaces (paths relative to the reference root).  INTEGRATION.md
 * shows the ctypes stubs a maintainer of the reference would add.
 *
 * Conventions
 *  - All arithmetic is IEEE float64; indices are int64.
 *  - Every pointer named d_* is a DEVICE pointer owned by the caller (the Python
 *    host side allocates them as torch tensors); the library never allocates or
 *    frees device memory.  h_* are HOST pointers.  Result scalars (h_best, h_out, ...) may be
 *    pageable or page-locked: into page-locked memory (hipHostMalloc / hipHostRegister) the last
 *    kernel of the call writes them itself, otherwise they arrive by a small device-to-host copy.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *    kernels are enqueued on it; functions that return host scalars say so and
 *    synchronise that stream themselves.
 *  - Return value: 0 = success, otherwise a hipError_t (or -1 for an argument
 *    error); obe_last_error() describes the most recent failure on this thread.
 *  - No exceptions cross the ABI.  A handle-free, re-entrant design: all state
 *    lives in caller-owned buffers; calls on different streams are independent.
 *  - Particles are stored SoA: row i of `d_particles` (parameter i of every
 *    particle) starts at d_particles + i * ld_p   (particlepdf.py:101-105).
 *    Settings likewise: row k of `d_settings` at d_settings + k * ld_s
 *    (obe_base.py:174-176, meshgrid 'ij' flattened).
 */
#ifndef OBE_HIP_H
#define OBE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The only symbols the library (and every per-model plugin) exports: everything else is compiled with
 * -fvisibility=hidden (optbayesexpt_amd/build.py), tests/test_capi_symbols.py checks `nm -D` against this header. */
#define OBE_API __attribute__((visibility("default")))

/* 2 (round 6): obe_resample_begin gained d_aos; the sweep / arg-max calls write a 32-byte record into the
 * workspace tail and the speculative pair keeps its abort word there (OBE_WS_RESULT_TAIL, OBE_WS_ABORT_WORD);
 * limits raised.  A client compiled against another version must not call in: compare obe_abi_version(). */
#define OBE_ABI_VERSION 2
#define OBE_MAX_CONSTS 8
#define OBE_MAX_CHANNELS 8   /* output channels of a device model / of one measurement record */
#define OBE_MAX_SETDIMS 8    /* setting dimensions of a device model */
#define OBE_MAX_DIMS 32      /* parameter rows a device model may be given (obe_model.n_params) */
/* Up to OBE_FAST_DIMS parameter rows the cloud kernels (moments, the fused update, the pipelined resample) are
 * compiled for the exact row count, all accumulators in registers.  The calls that work on the cloud alone —
 * obe_moments, obe_resample_particles, obe_gather_columns, the masks, the unfused updates — take ANY number of rows
 * (the reference has no limit: particlepdf.py:105): beyond OBE_FAST_DIMS they run tiled over 8 rows at a time (one
 * pass over the cloud per tile, per tile pair for the covariance).  The fused / pipelined forms
 * (obe_bayes_update_model_moments*, obe_resample_begin, obe_resample_particles_aos*, obe_mask_renorm_moments)
 * refuse more than OBE_FAST_DIMS rows before anything is launched (-1); callers then use the plain forms. */
#define OBE_FAST_DIMS 16
#define OBE_CLOUD_MAX_DIMS 1024

/* ---- device model registry (replaces the Python `model_function` callable,
 *      obe_base.py:50-72, for the model families the reference's demos use) ---- */
enum obe_model_id {
    OBE_MODEL_LORENTZ = 1,      /* y = b + sum_k a/(((x-x0_k)/d)^2+1); aux = #peaks K (1..8);
                                   params x0_1..x0_K, a, b [, extras]; const d.
                                   K=1: demos/find_peak/sequentialLorentzian.py:53-75,
                                   demos/sweeper/sweeper.py:42-64                     */
    OBE_MODEL_LINE_AB = 2,      /* y = p0 + p1*x          tests/test_optbayesexpt.py:11-14 */
    OBE_MODEL_LINE_MB = 3,      /* y = p0*x + p1          demos/line_plus_noise/line_plus_noise.py:36-53 */
    OBE_MODEL_FIRST_PARAM = 4,  /* y = p0                 tests/test_zinference.py:21-26 */
    OBE_MODEL_RABI = 5,         /* demos/pipulse/pipulse.py:18-49; 2 settings, 2 params, 3 consts */
    OBE_MODEL_COIL = 6,         /* demos/lockin/lockin_of_coil.py:63-102; 2 channels */
    OBE_MODEL_PLUGIN = 100      /* a model generated from a user expression: served by its own
                                   plugin library (same entry points, compiled for that model) */
};

typedef struct obe_model {
    int32_t id;          /* enum obe_model_id */
    int32_t aux;         /* model-specific integer (number of Lorentzian peaks) */
    int32_t n_params;    /* D: rows of the particle array (may exceed what the model reads) */
    int32_t n_setdims;   /* S */
    int32_t n_channels;  /* C */
    int32_t n_consts;
    double consts[OBE_MAX_CONSTS];
} obe_model;

/* Checks id/aux/n_* consistency; fills n_setdims/n_channels if they are 0. */
OBE_API int obe_model_validate(obe_model* m);

/* ---- library ---- */
OBE_API int obe_abi_version(void);
/* Hash of the kernel sources this binary was compiled from (optbayesexpt_amd/build.py); the
 * loader compares it with the sources next to it and refuses a stale library. */
OBE_API const char* obe_source_fingerprint(void);
OBE_API const char* obe_last_error(void);
/* Deferred host results (per calling thread).  While on, the entry points that deliver results
 * to host memory and would synchronise `stream` for it — obe_moments (h_out), obe_weight_cdf
 * (h_total), obe_ziggurat_normal (h_consumed[2]) — only enqueue the copy and return; the values
 * are valid once the caller has synchronised the stream (pass page-locked host memory, or the
 * copy itself blocks).  resample() uses it to keep the device busy while the host factorises the
 * covariance.  Returns the previous state. */
OBE_API int obe_defer_host_sync(int32_t on);
/* The address under which kernels of the current device reach a page-locked host buffer
 * (hipHostMalloc / hipHostRegister; fails for pageable memory).  A `d_*` output argument that is
 * only a few values — the index of obe_draw_indices for good_setting() — may be given as this
 * address: the kernel then delivers the result to the host itself, with no copy back. */
OBE_API int obe_host_device_pointer(const void* h_pinned, void** d_out);
/* Waiting for such results by watching them: obe_host_words_arm stores a bit pattern no result has
 * (0x7ff8c0dec0dec0de: a NaN payload / an impossible index) into EVERY 8-byte word of a page-locked result
 * block, the call whose kernels write those words is enqueued, and obe_host_words_wait spins until none of
 * the words carries the pattern any more (bounded: after 400 us it synchronises `stream` instead, which also
 * reports a kernel that failed).  An 8-byte store arrives whole, so no ordering between the kernel's stores
 * is relied upon — watching one "last" word behind a system-scope fence was not safe for blocks that span
 * several 128-byte lines ("Waiting for a result by watching it").  5.9 us per round trip
 * of a short kernel instead of 11.1 us through hipStreamSynchronize (tools/microbench_sync.hip).  The entry
 * points that deliver host results into page-locked memory (obe_sweep_utility, obe_utility_argmax,
 * obe_argmax, obe_bayes_update_*, obe_weight_sums) wait this way themselves; obe_resample_begin and
 * obe_mask_nonpositive_moments arm their result words and leave the waiting to the caller.
 * obe_host_word_arm / _wait: the same for a single word. */
OBE_API int obe_host_words_arm(void* h_pinned_words, int64_t n_words);
OBE_API int obe_host_words_wait(const void* h_pinned_words, int64_t n_words, void* stream);
OBE_API int obe_host_word_arm(void* h_pinned_word);
OBE_API int obe_host_word_wait(const void* h_pinned_word, void* stream);
/* Name, CU count and memory of the current device; returns 0 if a gfx950 device is current. */
OBE_API int obe_device_info(char* name, int name_len, int* n_cu, int64_t* hbm_bytes);

/* Scratch (device) bytes any call below may need for these sizes.  The sweep keeps its draws,
 * packed once per call, in the workspace: pass n_particles >= the number of draws of any sweep
 * (N_DRAWS may exceed the particle count).  A plugin library answers for its own model. */
OBE_API int64_t obe_workspace_bytes(int64_t n_particles, int64_t n_settings, int32_t n_channels, int32_t n_dims);

/* ---- K2: Bayes update  (obe_base.py:385-394 eval_over_all_parameters + likelihood +
 *      particlepdf.py:136-139 _normalized_product + :243-244 N_eff) ------------------
 * Fused: y = model(setting; particle), L = prod_ch exp(-((y-y_meas)/sigma)^2/2)/sigma,
 * [L = L^choke], t = nan_to_num(w*L), w' = nan_to_num(t/sum t), in place in d_weights.
 * sigma per channel: h_sigma[ch] (known noise, obe_base.py:451-456) when
 * h_noise_rows == NULL, otherwise row h_noise_rows[ch] of the particle array
 * (OptBayesExptNoiseParameter.likelihood, obe_noiseparam.py:109-115).
 * n_lik_channels = number of channels entering the product (zip truncation,
 * obe_base.py:453-455).  choke: NaN = none (obe_base.py:458-459).
 * h_out[0] = sum t, h_out[1] = sum nan_to_num(w'^2)  (host; stream is synchronised). */
OBE_API int obe_bayes_update_model(const obe_model* m,
                           const double* d_particles, int64_t ld_p, int64_t n_particles,
                           double* d_weights,
                           const double* h_setting, const double* h_y_meas,
                           const double* h_sigma, const int32_t* h_noise_rows,
                           int32_t n_lik_channels, double choke,
                           void* d_ws, int64_t ws_bytes, double* h_out, void* stream);

/* obe_bayes_update_model followed by the first pass of obe_moments on the updated weights, with the
 * normalisation and the moment sums fused into one pass over the cloud (pdf_update() of
 * obe_base.py:340-399 + mean()/std() of particlepdf.py:173-214, which every caller's cycle asks for
 * next, and the sweep's per-setting shift).  d_moments receives the K3 block without the covariance
 * (layout of obe_moments, bit-identical to calling it afterwards); h_out (nullable; sync) receives
 * [0] = sum t, [1] = sum w'^2, [2 .. 2 + 2 + 4 n_params) = that block.  n_dims = m->n_params.
 * Two launches: the workgroup of the normalisation pass that finishes last folds everyone's partial sums
 * (write-through partials, an arrival counter the library keeps per stream).  The weights and the moments
 * are the bits of obe_bayes_update_model() + obe_moments(); h_out[1] is summed from this pass's own
 * workgroup partials (one per CU) and can differ from obe_bayes_update_model()'s h_out[1] in the last
 * few ulp — never in a resample decision over the experiments tested
 * (tests/test_gpu_units.py::test_fused_and_unfused_update_take_the_same_resample_decisions). */
OBE_API int obe_bayes_update_model_moments(const obe_model* m,
                                   const double* d_particles, int64_t ld_p, int64_t n_particles,
                                   double* d_weights,
                                   const double* h_setting, const double* h_y_meas,
                                   const double* h_sigma, const int32_t* h_noise_rows,
                                   int32_t n_lik_channels, double choke, double* d_moments,
                                   void* d_ws, int64_t ws_bytes, double* h_out, void* stream);

/* (Libraries built with -DOBE_ONE_PASS_UPDATE only — a measured alternative that is not in the product build,
 * where this switch is accepted and changes nothing: csrc/obe_update.hip, profiles/r05_update_moments.txt.)
 * The form obe_bayes_update_model_moments() and its enqueue variant take on the calling thread: on = 1 — both
 * passes in ONE launch (a grid barrier between the likelihood pass and the normalisation; every thread keeps its
 * particles and their unnormalised weights in registers, so the cloud is read once: 8 (D + 1) N bytes read + 8 N
 * written instead of twice that) wherever it applies (an arrival counter for the stream, at most 6 particles per
 * thread of the 768-workgroup grid, i.e. N <= 1 179 648, n_params = the model's own parameters or one more, a
 * co-resident grid, and the first-moment passes on the update's grid: OBE_FIRST_MOM_PER_CU=3), the two launches
 * otherwise; on = 0 — always two launches.  The results are the same bits either way.  Returns the previous
 * setting; on = -1 changes nothing and returns what the thread's last fused update did (1: one launch, 2: two). */
OBE_API int obe_update_one_pass(int32_t on);

/* strict sums (per calling thread; returns the previous setting, on < 0 only asks): while on, the UNFUSED updates —
 * obe_bayes_update_model, obe_bayes_update_y, obe_bayes_update_lik — form sum t and sum nan_to_num(w'^2) in the order
 * in which np.sum adds a contiguous float64 vector (pairwise_sum: pieces of 8192, runs of <= 128 in eight interleaved
 * running sums, halves split at multiples of 8; restated in oracle/obe_oracle.py: numpy_pairwise_sum and pinned there
 * against np.sum for every length to 5000), by one workgroup: the normalised weights t / np.sum(t) and N_eff are then
 * the reference's BITS (particlepdf.py:136-139, 243-244; its own tests compare them with assert_array_equal,
 * tests/test_optbayesexpt.py:58-69) wherever t itself is.  Meant for small clouds (the host side switches it on up to
 * 4096 particles: tuning_parameters['strict_sums']); any size works, one workgroup's speed.  The fused forms
 * (obe_bayes_update_model_moments*) ignore it. */
OBE_API int obe_strict_sums(int32_t on);

/* The same update, enqueued only: returns without waiting.  h_pinned_out (page-locked, 5 + 4 n_params
 * doubles) is armed here and written by the update's last kernel: [0] sum t, [1] sum w'^2, [2..) the K3
 * first-moment block, and [4 + 4 n_params] = 1.0 if auto_resample != 0 and the resample test of
 * particlepdf.py:236-258 on sum w'^2 (N_eff < 0.1 N or N_eff / N < resample_threshold) says "resample",
 * else 0.0 — wait with obe_host_words_wait(h_pinned_out, 5 + 4 n_params, stream).  The same decision stays
 * on the device, in the LAST 8-byte word of the caller's workspace (OBE_WS_ABORT_WORD: d_ws + ws_bytes - 8,
 * ws_bytes a multiple of 8), for an obe_sweep_utility(OBE_SWEEP_SPECULATIVE) call enqueued next on this
 * stream WITH THE SAME d_ws AND ws_bytes, which hides the host round trip of the update behind the sweep
 * (the reference's cycle is update -> resample test -> next opt_setting: obe_base.py:340-399, 733-756).
 * The word belongs to whoever owns the workspace: updates of other objects on the same stream do not touch it.
 * Refused (-1) BEFORE anything is launched — the weights are untouched and the caller uses the synchronous
 * form — when h_pinned_out is not page-locked, when the workspace has no 16 spare bytes behind what the
 * update uses (obe_workspace_bytes() always leaves them), or when the library has no arrival counter for
 * the stream (device allocation failed; a full table of 256 streams per device hands its least recently
 * used entry on after a device synchronisation, so it never refuses). */
#define OBE_WS_ABORT_WORD(d_ws, ws_bytes) ((unsigned*)((char*)(d_ws) + ((ws_bytes) & ~(int64_t)7) - 8))
OBE_API int obe_bayes_update_model_moments_enqueue(const obe_model* m, const double* d_particles, int64_t ld_p,
                                           int64_t n_particles, double* d_weights, const double* h_setting,
                                           const double* h_y_meas, const double* h_sigma,
                                           const int32_t* h_noise_rows, int32_t n_lik_channels, double choke,
                                           double* d_moments, void* d_ws, int64_t ws_bytes, double* h_pinned_out,
                                           int32_t auto_resample, double resample_threshold, void* stream);
/* A whole sweep of measurements (demos/sweeper/obe_sweeper.py:86-100: one pdf_update per point,
 * each followed by the resample test of particlepdf.py:236-258) enqueued back to back, no
 * host round trip between the points.  h_settings (n_points, OBE_MAX_SETDIMS) and h_y_meas
 * (n_points, OBE_MAX_CHANNELS) row-major; sigma / noise rows / choke as obe_bayes_update_model.
 * With auto_resample != 0 the test runs on the device after every point (in the prologue of
 * the next point's first pass: two launches per point): as soon as n_eff < 0.1 N or
 * n_eff / N < resample_threshold the remaining points are skipped (their kernels return at once).  h_out[0] = sum t and h_out[1] = sum w'^2 of the last point applied,
 * h_out[2] = 1 if a resample is due, h_out[3] = points applied (>= 1); the caller resamples and
 * submits the rest.  Sync. */
OBE_API int obe_bayes_update_sweep(const obe_model* m, const double* d_particles, int64_t ld_p,
                           int64_t n_particles, double* d_weights, const double* h_settings,
                           const double* h_y_meas, const double* h_sigma,
                           const int32_t* h_noise_rows, int32_t n_lik_channels, double choke,
                           int64_t n_points, int32_t auto_resample, double resample_threshold,
                           void* d_ws, int64_t ws_bytes, double* h_out, void* stream);
/* Same update from precomputed model outputs d_y (C, N_p) row-major, ld_y between
 * channels  (pdf_update(..., y_model_data), obe_base.py:384-385). */
OBE_API int obe_bayes_update_y(const double* d_y, int64_t ld_y, int32_t n_channels,
                       const double* d_particles, int64_t ld_p, int64_t n_particles,
                       double* d_weights, const double* h_y_meas,
                       const double* h_sigma, const int32_t* h_noise_rows,
                       int32_t n_lik_channels, double choke,
                       void* d_ws, int64_t ws_bytes, double* h_out, void* stream);

/* ParticlePDF.bayesian_update(likelihood) with a caller-supplied likelihood array
 * (particlepdf.py:216-234). */
OBE_API int obe_bayes_update_lik(const double* d_lik, int64_t n_particles, double* d_weights,
                         void* d_ws, int64_t ws_bytes, double* h_out, void* stream);

/* OptBayesExpt.likelihood(y_model, record) as an array (obe_base.py:418-461).  ANY number of channels (the
 * reference has no limit, obe_base.py:807-824): h_y_meas / h_sigma / h_noise_rows then hold n_lik_channels values
 * and a record wider than OBE_MAX_CHANNELS is multiplied up in groups of that many, in channel order, the choke
 * applied to the finished product. */
OBE_API int obe_likelihood_y(const double* d_y, int64_t ld_y, int32_t n_channels,
                     const double* d_particles, int64_t ld_p, int64_t n_particles,
                     const double* h_y_meas, const double* h_sigma,
                     const int32_t* h_noise_rows, int32_t n_lik_channels, double choke,
                     double* d_lik_out, void* stream);

/* h_out[0] = sum nan_to_num(w^2) (resample_test, particlepdf.py:243-244), h_out[1] = sum w. */
OBE_API int obe_weight_sums(const double* d_weights, int64_t n_particles,
                    void* d_ws, int64_t ws_bytes, double* h_out, void* stream);

/* ---- model evaluation wrappers (obe_base.py:298-338) ---- */
/* d_y_out (C, N_p): model(one setting; all particles). */
OBE_API int obe_eval_over_particles(const obe_model* m, const double* d_particles, int64_t ld_p,
                            int64_t n_particles, const double* h_setting,
                            double* d_y_out, int64_t ld_y, void* stream);
/* d_y_out (C, N_s): model(all settings; one parameter set h_params[D]). */
OBE_API int obe_eval_over_settings(const obe_model* m, const double* d_settings, int64_t ld_s,
                           int64_t n_settings, const double* h_params,
                           double* d_y_out, int64_t ld_y, void* stream);

/* ---- K3: weighted moments (particlepdf.py:173-214) ----
 * d_out layout (doubles): [0]=sum w, [1]=sum w^2, [2..2+D) mean (np.average),
 * [2+D..2+2D) m1 = sum w x, [2+2D..2+3D) m2 = sum w x^2, [2+3D..2+4D) std,
 * then (if want_cov) D*D covariance (np.cov aweights, ddof=1 form).  Stays on the
 * device (other kernels consume it); copied to h_out too when h_out != NULL (sync).
 * want_cov = 2: d_out already holds the first moments of these particles and weights (from an earlier
 * call or from obe_bayes_update_model_moments): only the covariance pass runs, about the mean found
 * there; of h_out only the covariance part is then written by a page-locked delivery. */
OBE_API int64_t obe_moments_len(int32_t n_dims);
OBE_API int obe_moments(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                const double* d_weights, int32_t want_cov,
                double* d_out, double* h_out, void* d_ws, int64_t ws_bytes, void* stream);

/* ---- K4: multinomial resampling (particlepdf.py:260-345) ----
 * CDF of Generator.choice: cumsum(w) / cumsum(w)[-1].  strict_order != 0 reproduces
 * np.cumsum's serial rounding bit for bit (slow, one wavefront); 0 = parallel
 * blocked scan (same value to ~1e-13, indices identical unless a uniform falls
 * inside that gap).  h_total (nullable) receives what numpy validates of p (sync): sum(w) — NaN if any
 * weight is NaN — or -inf when some weight is negative ("Probabilities are not non-negative"). */
OBE_API int obe_weight_cdf(const double* d_weights, int64_t n_particles, int32_t strict_order,
                   double* d_cdf, double* h_total, void* d_ws, int64_t ws_bytes, void* stream);
/* ---- sweeper composition (demos/sweeper/obe_sweeper.py:122-149) ----
 * obe_cumsum: out = np.cumsum(x) (strict_order as above; no normalisation).
 * obe_interval_utility: utility[i] = (cum[stop_i] - cum[start_i]) / cost_i for the
 * n_pairs (start, stop) index rows of d_pairs (int64, row-major (n_pairs, 2)) — the
 * `(ends[:, 1] - ends[:, 0]) / cost` of sweep_utility(); cost_i = d_cost[i], or
 * stop_i - start_i + cost_of_new_sweep (sweep_cost_estimate(), obe_sweeper.py:106-120)
 * when d_cost == NULL.  Follow with obe_argmax. */
OBE_API int obe_cumsum(const double* d_x, int64_t n, int32_t strict_order, double* d_out,
               void* d_ws, int64_t ws_bytes, void* stream);
OBE_API int obe_interval_utility(const double* d_cum, int64_t n_settings, const int64_t* d_pairs,
                         int64_t n_pairs, const double* d_cost, double cost_of_new_sweep,
                         double* d_utility, void* stream);
/* The index part of randdraw() for a small draw (n_draws <= 64; the reference's N_DRAWS = 30,
 * good_setting's single draw): the uniforms travel as kernel arguments, the CDF is rebuilt
 * unless cdf_is_fresh (d_cdf still holds the CDF of d_weights), and for clouds of up to
 * 14 336 particles scan and search are one launch (beyond that the three-kernel scan is faster).  Same CDF bits and indices as
 * obe_weight_cdf + obe_cdf_search.  Never synchronises: when the CDF is rebuilt and
 * h_total_pinned != NULL (pinned host memory: written by the last kernel itself; pageable: an asynchronous copy), sum(w) lands there and is
 * valid after the caller's next synchronisation of the stream — or, for pinned memory, once the word itself has arrived: arm it with
 * obe_host_word_arm() before the call and wait for IT with obe_host_word_wait().  The arrival of another word (the indices copied
 * back, a later call's result) does not imply this one's: stores to host memory do not arrive in the order they were issued. */
OBE_API int obe_draw_indices(const double* d_weights, int64_t n_particles, int32_t strict_order,
                     int32_t cdf_is_fresh, double* d_cdf, const double* h_uniforms,
                     int32_t n_draws, int64_t* d_idx, double* h_total_pinned,
                     void* d_ws, int64_t ws_bytes, void* stream);
/* Systematic resampling (extension; BASELINE.json north_star names it, the reference itself is
 * multinomial): idx[i] = searchsorted(cdf, (i + u0) / n_draws, 'right') for ONE uniform u0 in
 * [0, 1).  Selected with tuning_parameters['resample_method'] = 'systematic'. */
OBE_API int obe_systematic_indices(const double* d_cdf, int64_t n, double u0, int64_t n_draws,
                           int64_t* d_idx_out, void* stream);
/* idx[j] = #{i : cdf[i] <= u[j]}  (searchsorted side='right'), int64.  d_ws (nullable): with at
 * least n / 2 + 8 bytes of scratch, a search of many draws (n_draws >= n / 4, n >= 32 768: a
 * resample) first builds a guide table of n / 8 + 1 bucket starts and then searches only the handful
 * of entries a draw's bucket spans — the same indices, ~3 instead of ~11 cold accesses per draw. */
OBE_API int obe_cdf_search(const double* d_cdf, int64_t n, const double* d_uniforms, int64_t n_draws,
                   int64_t* d_idx_out, void* d_ws, int64_t ws_bytes, void* stream);
/* randdraw gather: d_out (D, n_draws) = particles[:, idx]  (particlepdf.py:332-343). */
OBE_API int obe_gather_columns(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                       const int64_t* d_idx, int64_t n_draws,
                       double* d_out, int64_t ld_out, void* stream);
/* resample(): new[i,p] = old[i, idx[p]] + sum_j z[p,j] F[i,j]   (F = u*sqrt(s) of the
 * SVD of (1-a^2) cov, h_factor row-major D*D; z = standard normals (N, D) row-major);
 * if scale: new = new*a + mean[i]*(1-a)   (particlepdf.py:296-305).  Then weights = 1/N.
 * d_ws (nullable): with 8 * n_dims * n_particles bytes of scratch a large cloud is first copied
 * to (N, D) order, so that the gather of a particle touches one or two 64-byte sectors instead
 * of n_dims of them. */
OBE_API int obe_resample_particles(const double* d_old, int64_t ld_old, int32_t n_dims, int64_t n_particles,
                           const int64_t* d_idx, const double* d_normals,
                           const double* h_factor, const double* h_mean,
                           double a_param, int32_t scale,
                           double* d_new, int64_t ld_new, double* d_weights,
                           void* d_ws, int64_t ws_bytes, void* stream);
/* The same with the (N, D) copy of the old cloud already made (obe_resample_begin's d_aos): no copy pass in
 * front of the gather. */
OBE_API int obe_resample_particles_aos(const double* d_old_aos, int32_t n_dims, int64_t n_particles,
                               const int64_t* d_idx, const double* d_normals,
                               const double* h_factor, const double* h_mean,
                               double a_param, int32_t scale,
                               double* d_new, int64_t ld_new, double* d_weights, void* stream);

/* ... and with the first half of OptBayesExptNoiseParameter.enforce_parameter_constraints
 * (obe_noiseparam.py:57-79), which pdf_update() runs right after a resample, done by the gather: a new particle
 * whose parameter h_rows[k] <= 0 for any k gets weight 0 instead of 1/N, and d_mask_partials (2 x 2048 doubles,
 * the caller's) receives the partial sums {sum w, count} that obe_mask_nonpositive()'s first kernel would leave —
 * same grid, same order, same bits — for obe_mask_renorm_moments().  ONLY for a resample that the constraint
 * follows: the reference's resample() on its own leaves uniform weights. */
OBE_API int obe_resample_particles_aos_masked(const double* d_old_aos, int32_t n_dims, int64_t n_particles,
                                      const int64_t* d_idx, const double* d_normals,
                                      const double* h_factor, const double* h_mean,
                                      double a_param, int32_t scale,
                                      double* d_new, int64_t ld_new, double* d_weights,
                                      const int32_t* h_rows, int32_t n_rows, double* d_mask_partials, void* stream);

/* The random numbers of a resample, enqueued AHEAD of it (round 6): the N uniforms and N x D normals that resample()
 * takes from the caller's generator (particlepdf.py:272, 296-301) depend only on the generator state and the cloud's
 * shape, and their chain is what a resample's gather ends up waiting for.  Enqueued here — on the library's side stream of
 * `stream`, typically when pdf_update() starts — it runs beside the update and the host round trips;
 * obe_resample_begin(h_pcg_state4 = NULL, the same d_uniforms / d_normals / d_zig_ws / h_i64, the same stream) then
 * launches only the cloud's chains and waits for this one.  The caller compares generator states itself and may keep
 * the numbers across updates that do not resample.  h_i64[0..1] (page-locked) are armed here; wait for them as after
 * obe_resample_begin.  -1 before anything is launched: no side streams, h_i64 not page-locked, n_dims > OBE_FAST_DIMS. */
OBE_API int obe_resample_randoms_enqueue(const uint64_t* h_pcg_state4, int64_t n_particles, int32_t n_dims, int64_t n_raw,
                                 double* d_uniforms, const void* d_zig_tables, double* d_normals, void* d_zig_ws,
                                 int64_t zig_ws_bytes, int64_t* h_i64, void* stream);
/* resample(), the device side up to the host's factorisation of the covariance, enqueued by ONE call
 * (particlepdf.py:260-301; RNG order as there: N uniforms for rng.choice, then N x D normals): the caller's
 * PCG64 stream continued on the device (h_pcg_state4 = {state hi, lo, increment hi, lo}; n_raw >= N + N D +
 * 4096 raw values are looked at, none is stored), the weight CDF into d_cdf (skipped when cdf_is_fresh), the N uniforms, the
 * search of the N draws (d_idx), the covariance of the PRE-resample cloud (have_first_moments: d_moments
 * already holds mean / std of these weights) and the N x D ziggurat normals (d_normals; d_zig_ws of
 * obe_ziggurat_workspace_bytes(n_raw - N) bytes).  Nothing is waited for.  Both host buffers must be
 * page-locked; every result word is armed here and the caller waits with obe_host_words_wait():
 *   h_f64[0]     sum(w) for numpy's validation of p (1.0, stored at once, when the CDF was fresh);
 *   h_f64[1..]   the K3 block: the covariance and, unless have_first_moments, the first moments — wait for
 *                the words h_f64 + 1 + lo .. h_f64 + 1 + obe_moments_len(D), lo = have_first ? 2 + 4 D : 0;
 *   h_i64[0..1]  {raw values the normals consumed, normals found}: wait for both, then obe_ziggurat_check().
 * d_aos (nullable, 8 n_dims N bytes): receives the (N, D) copy of the PRE-resample cloud for
 * obe_resample_particles_aos() — made here, while the host factorises, instead of in front of the gather; on
 * large clouds the covariance and this copy then run as a third chain beside the CDF / search and the random
 * numbers.  Same kernels, same numbers as the calls one by one. */
OBE_API int obe_resample_begin(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                       const double* d_weights, const uint64_t* h_pcg_state4, int32_t strict_cdf,
                       int32_t cdf_is_fresh, int32_t have_first_moments, int64_t n_raw,
                       double* d_cdf, double* d_uniforms, int64_t* d_idx, const void* d_zig_tables,
                       double* d_normals, void* d_zig_ws, int64_t zig_ws_bytes, double* d_moments,
                       double* h_f64, int64_t* h_i64, double* d_aos, void* d_ws, int64_t ws_bytes, void* stream);

/* ---- K6: OptBayesExptNoiseParameter extras ----
 * enforce_parameter_constraints (obe_noiseparam.py:57-79): zero the weight of every
 * particle whose row h_rows[k] <= 0 for any k, renormalise if anything changed.
 * *h_changed = number of particles zeroed (sync). */
OBE_API int obe_mask_nonpositive(const double* d_particles, int64_t ld_p, int64_t n_particles,
                         const int32_t* h_rows, int32_t n_rows, double* d_weights,
                         int64_t* h_changed, void* d_ws, int64_t ws_bytes, void* stream);

/* The same mask AND the first moments of the constrained cloud (what obe_moments(want_cov = 0) would
 * compute next: every cycle needs them for the sweep's shift and the noise-parameter variance), bit for
 * bit, in two launches without a host round trip in between (obe_noiseparam.py:57-79 + particlepdf.py:
 * 173-214).  Does not wait: page-locked h_changed (1 word) and h_moments (the K3 block's 2 + 4 n_dims
 * first-moment values; either may be NULL) are armed here and written by the second kernel — the caller
 * waits with obe_host_words_wait() on each.  Pageable host buffers: the two calls above, synchronously. */
OBE_API int obe_mask_nonpositive_moments(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                                 const int32_t* h_rows, int32_t n_rows, double* d_weights, double* d_moments,
                                 double* h_moments, int64_t* h_changed, void* d_ws, int64_t ws_bytes,
                                 void* stream);

/* yvar_noise_model (obe_noiseparam.py:122-136): d_out[c] = weighted mean of
 * (particle row h_rows[c])^2, read from the K3 block: m2[row] / sum w.  No sync. */
/* The second half of obe_mask_nonpositive_moments() on its own — renormalise if anything was zeroed, first moments
 * of the constrained cloud, nothing waited for — from the partial sums a masked gather left in d_mask_partials
 * (obe_resample_particles_aos_masked).  Refused (-1) before any launch without an arrival counter for the stream or
 * with pageable host outputs: the caller then calls obe_mask_nonpositive_moments(), which on weights the gather has
 * already masked zeroes the same particles and leaves the same bits. */
OBE_API int obe_mask_renorm_moments(const double* d_particles, int64_t ld_p, int32_t n_dims, int64_t n_particles,
                            const double* d_mask_partials, double* d_weights, double* d_moments,
                            double* h_moments, int64_t* h_changed, void* d_ws, int64_t ws_bytes, void* stream);
OBE_API int obe_noise_var_from_moments(const double* d_moments, int32_t n_dims, const int32_t* h_rows,
                               int32_t n_rows, double* d_out, void* stream);

/* ---- good_setting (obe_base.py:781-784): p = nan_to_num(u ** exponent); p /= sum(p) ---- */
OBE_API int obe_power_normalize(const double* d_u, int64_t n, double exponent, double* d_p_out,
                        void* d_ws, int64_t ws_bytes, void* stream);

/* ---- K1 + K5: utility sweep and argmax (obe_base.py:463-489, 628-655, 733-756) ----
 * Evaluates the model over settings [s_begin, s_begin + n_settings) x draws and
 * reduces to the per-setting variance of the model output, per channel.
 *   d_draw_idx == NULL : full sweep — every particle is a draw, weighted variance
 *                        sum_p w_p (y - ybar)^2 / sum_p w_p     
 *   d_draw_idx != NULL : reference semantics — the n_draws listed particles, unweighted
 *                        ddof=0 variance (np.var over utility_y_space, obe_base.py:488)
 * d_moments: output of obe_moments for the same particles/weights (mean parameters are
 * used as the variance shift, sum w as the normaliser).
 * `shifted` is a bit set: OBE_SWEEP_SHIFTED (1) and, for expression (plugin) models,
 * OBE_SWEEP_SAFE (2).
 * OBE_SWEEP_SHIFTED set: moments are accumulated about a per-setting shift (always accurate).
 * OBE_SWEEP_SHIFTED clear: one instruction fewer per evaluation, accurate only while the predicted
 * mean does not dominate the spread; *h_kappa (nullable) returns the worst
 * (mean of y)^2 / var over settings and channels — the factor by which an unshifted sweep
 * amplifies rounding — so the caller can choose the mode for the next sweep, or repeat
 * this one with shifted = 1 if an unshifted result came back with a large factor.
 * Then utility[s] = sum_c yvar[c,s] / noise_var[c(,s)] / cost[(s)]   (obe_base.py:650-655)
 * with d_noise_var (C) if noise_ld == 0, (C, n_settings) rows noise_ld apart if noise_ld > 0, and, if
 * noise_ld = OBE_NOISE_FROM_MOMENTS(r0, r1, r2, r3) < 0, taken from the K3 block d_noise_var then points to
 * (obe_moments layout for m->n_params parameters; normally d_moments itself): channel c's noise variance is
 * that block's m2[r_c] / sum w — the weighted mean of sigma_c^2 of obe_noiseparam.py:122-136, the division
 * obe_noise_var_from_moments() does, without a launch of its own — and
 * cost = cost_scalar if d_cost == NULL else d_cost[s]; and the first-maximum argmax
 * (np.argmax, obe_base.py:748): h_best[0] = value, h_best_idx[0] = index relative to
 * s_begin (sync) when h_best != NULL.  d_yvar (C, n_settings) and d_utility
 * (n_settings) stay on the device. */
/* After obe_sweep_utility / obe_utility_argmax / obe_argmax the workspace holds the result
 * as one 32-byte record at d_ws + OBE_WS_RESULT_OFFSET doubles:
 * {best value (f64), best local index (i64 bits), kappa (f64), 0} — what a sharded caller
 * all-gathers across ranks (RCCL) without copying it to the host first. */
#define OBE_WS_RESULT_OFFSET 2
/* ... and once more in the last 48 bytes of the workspace, 4 doubles at d_ws + ws_bytes - 48 (ws_bytes a multiple
 * of 8; written when the workspace has that much room behind what the call uses — obe_workspace_bytes() always
 * leaves it): the update calls reuse the HEAD of the workspace, so a record that has to outlive them — the sweep
 * enqueued behind an update, whose record a sharded caller all-gathers one host round trip later — is read
 * there.  The word behind it is OBE_WS_ABORT_WORD. */
#define OBE_WS_RESULT_TAIL(d_ws, ws_bytes) ((double*)((char*)(d_ws) + ((ws_bytes) & ~(int64_t)7) - 48))
/* (5 bits per channel, channel c's row at bit 5 c, up to OBE_MAX_CHANNELS channels; rows below 32) */
#define OBE_NOISE_FROM_MOMENTS(r0, r1, r2, r3) \
    (-(int64_t)1 - ((int64_t)(r0) | ((int64_t)(r1) << 5) | ((int64_t)(r2) << 10) | ((int64_t)(r3) << 15)))
/* bits of the `shifted` argument of obe_sweep_utility / obe_sweep_kernel_time.  A plugin
 * model's fast sweep form batches its divisions without a branch and poisons (NaN) a batch
 * whose denominators leave the range in which that is exact; a NaN variance comes back as
 * *h_kappa = NaN, and the caller repeats the sweep with OBE_SWEEP_SAFE (one IEEE
 * reciprocal per element, implies the shift).  Of the built-in models the coil and the Lorentzian with 3
 * or more peaks have such a pair of forms; the latter's safe form still shares one reciprocal among the
 * 16 denominators of two particles x 8 settings of ONE peak and is exact for |x - x0| / d up to ~2e9
 * (tested to 2.5e7, tests/test_gpu_units.py::test_multi_peak_lorentzian_sweep_forms); the others ignore
 * OBE_SWEEP_SAFE.
 *
 * Completion of the "(sync)" entry points: the HOST results of a call are complete when it returns — they
 * are written by the call's last kernel into page-locked memory and waited for there — while device outputs
 * (weights, moments, utility ...) are ordered on the caller's stream like any kernel's: consume them on the
 * same stream, or synchronise it first.  obe_mask_nonpositive() in particular returns as soon as the count
 * is known, while the renormalisation it implies may still be running. */
#define OBE_SWEEP_SHIFTED 1
#define OBE_SWEEP_SAFE 2
/* OBE_SWEEP_SPECULATIVE (full sweeps only): the call is enqueued behind
 * obe_bayes_update_model_moments_enqueue() on the same stream, with the same d_ws and ws_bytes, and returns
 * at once.  Its kernels read the resample decision that update left in OBE_WS_ABORT_WORD(d_ws, ws_bytes) and
 * do nothing if it was "resample" (the cloud is about to change: particlepdf.py:236-258).  Refused (-1, before
 * any launch) if the workspace has no 16 spare bytes behind what the sweep uses.  Page-locked h_best / h_best_idx / h_kappa are armed, not waited
 * for: the caller first learns from the update's host block (word 4 + 4 n_params) whether the sweep ran,
 * and only then waits for the words with obe_host_words_wait().  An aborted call leaves them armed and the
 * device outputs (yvar, utility, the result record) untouched. */
#define OBE_SWEEP_SPECULATIVE 8
/* OBE_SWEEP_NOWAIT (full sweeps only): enqueued and not waited for, like OBE_SWEEP_SPECULATIVE, but
 * unconditional — the caller knows the cloud is final (after a resample) and collects the page-locked
 * result words later with obe_host_words_wait(). */
#define OBE_SWEEP_NOWAIT 16
/* Settings one lane of the sweep kernel owns at most for a grid of n_settings (1, 2, 4 or 8; a sweep of
 * few draws may use fewer): the number of denominators a model's fast form inverts together, which a
 * caller that predicts whether a settings grid stays inside that form's range needs
 * (optbayesexpt_amd/models.py: range_hint). */
OBE_API int obe_sweep_settings_per_lane(int64_t n_settings);
/* ... and for a reference-semantics sweep of n_draws draws (obe_base.py:463-489 with N_DRAWS draws): 1 when the
 * one-workgroup kernel serves it.  A model whose fast form shares nothing but the reciprocal of a lane's
 * settings is IEEE as it is when this returns 1: no range check, no repeat (models.py: safe_sweep_min_spt). */
OBE_API int obe_sweep_settings_per_lane_for(int64_t n_settings, int64_t n_draws);
OBE_API int obe_sweep_utility(const obe_model* m,
                      const double* d_settings, int64_t ld_s, int64_t n_settings,
                      const double* d_particles, int64_t ld_p, int64_t n_particles,
                      const double* d_weights, const int64_t* d_draw_idx, int64_t n_draws,
                      const double* d_moments, int32_t shifted,
                      const double* d_noise_var, int64_t noise_ld,
                      const double* d_cost, double cost_scalar,
                      double* d_yvar, double* d_utility,
                      double* h_best, int64_t* h_best_idx, double* h_kappa,
                      void* d_ws, int64_t ws_bytes, void* stream);

/* Variance over the draw axis of a caller-filled y-space (N_d, C, N_s) — the
 * np.var(utility_y_space, axis=0) of obe_base.py:488 for host-callable models. */
OBE_API int obe_yspace_variance(const double* d_yspace, int64_t n_draws, int32_t n_channels,
                        int64_t n_settings, double* d_yvar, void* stream);
/* utility + argmax from an existing yvar (same conventions as obe_sweep_utility). */
OBE_API int obe_utility_argmax(const double* d_yvar, int32_t n_channels, int64_t n_settings,
                       const double* d_noise_var, int64_t noise_ld,
                       const double* d_cost, double cost_scalar,
                       double* d_utility, double* h_best, int64_t* h_best_idx,
                       void* d_ws, int64_t ws_bytes, void* stream);
/* first-maximum argmax of an arbitrary device vector (np.argmax semantics incl. NaN). */
OBE_API int obe_argmax(const double* d_v, int64_t n, double* h_best, int64_t* h_best_idx,
               void* d_ws, int64_t ws_bytes, void* stream);

/* ---- the non-default utilities on the explicit y-space ----
 * d_yspace (N_d, C, N_s) row-major = utility_y_space of the reference (obe_base.py:293-295). */
/* y[d][c][s] = model(setting s; particle d_draw_idx[d]) — the loop of obe_base.py:483-484
 * (eval_over_all_settings per drawn parameter set), exact NumPy operation order. */
OBE_API int obe_eval_draws(const obe_model* m, const double* d_settings, int64_t ld_s, int64_t n_settings,
                   const double* d_particles, int64_t ld_p, int64_t n_particles,
                   const int64_t* d_draw_idx, int64_t n_draws, double* d_yspace, void* stream);
/* y[d][c][:] += d_noise[d][c]  (utility_full_kld, obe_base.py:714-715). */
OBE_API int obe_yspace_add_noise(double* d_yspace, int64_t n_draws, int32_t n_channels, int64_t n_settings,
                         const double* d_noise, void* stream);
/* yvar_max_min (obe_base.py:520-535): (max - min)^2 over the draws, per column (C*N_s columns). */
OBE_API int obe_yspace_maxmin(const double* d_yspace, int64_t n_draws, int64_t n_columns, double* d_span2, void* stream);
/* scipy.stats.differential_entropy(axis=0, method='auto') per column (obe_base.py:516, 717-718);
 * as_variance != 0 returns exp(2H)/(2 pi e) (yvar_from_entropy, obe_base.py:517).
 * d_scratch: n_draws * n_columns doubles.  n_draws <= 2048. */
OBE_API int obe_yspace_entropy(const double* d_yspace, int64_t n_draws, int64_t n_columns, int32_t as_variance,
                       double* d_scratch, double* d_out, void* stream);
/* utility_full_kld (obe_base.py:720): exp(H_y[c,s] - H_noise[c]) - 1. */
OBE_API int obe_kld_utility(const double* d_entropy_y, int32_t n_channels, int64_t n_settings,
                    const double* d_entropy_noise, double* d_utility, void* stream);

/* ---- device-side continuation of the caller's numpy PCG64 stream ----
 * Replaces the host calls inside resample(): rng.random(n) (the uniforms Generator.choice
 * draws, particlepdf.py:330) and rng.standard_normal((n, d)) (inside multivariate_normal,
 * particlepdf.py:300).  h_state4 = {state_hi, state_lo, inc_hi, inc_lo} of
 * rng.bit_generator.state; d_raw[i] = the (i+1)-th next_uint64 of that generator. */
OBE_API int obe_pcg64_raw(const uint64_t* h_state4, int64_t n_raw, uint64_t* d_raw, void* stream);
/* d_out[i] = (d_raw[i] >> 11) * 2^-53  (numpy next_double / Generator.random). */
OBE_API int obe_pcg64_uniform(const uint64_t* d_raw, int64_t n, double* d_out, void* stream);
/* The same draws without a buffer of raw values, in two stages (what obe_resample_begin enqueues): stage 1 —
 * ONE launch for the n_uniform uniforms of Generator.choice and the classification of the n_raw_normal raw
 * positions behind them, every thread carrying the generator state of its position (h_state4 as for
 * obe_pcg64_raw); stage 2 — start flags, scan and compaction of the first n normals into d_out, h_consumed as
 * for obe_ziggurat_normal.  d_ws of obe_ziggurat_workspace_bytes(n_raw_normal) bytes, the same for both. */
OBE_API int obe_pcg64_uniforms_classify(const uint64_t* h_state4, int64_t n_uniform, int64_t n_raw_normal,
                                double* d_uniforms, const void* d_tables, void* d_ws, int64_t ws_bytes,
                                void* stream);
OBE_API int obe_ziggurat_finish(int64_t n_raw_normal, int64_t n, double* d_out, int64_t* h_consumed, void* d_ws,
                        int64_t ws_bytes, void* stream);
/* n standard normals by numpy's ziggurat from d_raw[offset...]: bit-identical values and
 * the exact number of raw values consumed (*h_consumed; sync), so the host generator can
 * be advanced to where numpy would have left it.  d_tables = ki[256] (uint64) | wi[256] |
 * fi[256] (float64).  Returns 1 if n_raw is too short (retry with a longer buffer).
 * With obe_defer_host_sync on: h_consumed[0..1] receive {raw values consumed, normals found}
 * asynchronously and the return value only reports launch errors; after synchronising, the
 * caller checks them with obe_ziggurat_check (1 = buffer too short). */
OBE_API int64_t obe_ziggurat_workspace_bytes(int64_t n_raw);
OBE_API int obe_ziggurat_normal(const uint64_t* d_raw, int64_t n_raw, int64_t offset, const void* d_tables,
                        int64_t n, double* d_out, int64_t* h_consumed,
                        void* d_ws, int64_t ws_bytes, void* stream);
OBE_API int obe_ziggurat_check(int64_t consumed, int64_t found, int64_t n, int64_t n_raw, int64_t offset);

/* ---- timing on the launch stream (bench.py roofline leg) ---- */
OBE_API int obe_timer_create(void** timer);
OBE_API int obe_timer_start(void* timer, void* stream);
OBE_API int obe_timer_stop(void* timer, void* stream, float* ms);   /* records, syncs, returns elapsed */
OBE_API int obe_timer_destroy(void* timer);
/* Timing of the dominant sweep kernel inside real cycles: while enabled, every obe_sweep_utility
 * call that returns its result to the host brackets the sweep kernel with two events on its
 * stream and adds the elapsed time to a running total (the events are read after the stream
 * synchronisation the result copy performs anyway).  Returns the total and the number of
 * launches accumulated so far, then: enable > 0 starts a fresh accumulation, enable == 0 stops
 * it, enable < 0 leaves the state alone (read only).  One accumulation per process (per loaded
 * library: a plugin keeps its own), not thread-safe.  bench.py: roofline.achieved. */
OBE_API int obe_sweep_timing(int32_t enable, double* h_total_ms, int64_t* h_launches);
/* Launches only the dominant sweep kernel `iters` times between two events on `stream`
 * and returns the average per-launch duration in ms; iters < 0: -iters isolated launches
 * (the stream is drained before each one), as a measurement cycle issues them. */
OBE_API int obe_sweep_kernel_time(const obe_model* m,
                          const double* d_settings, int64_t ld_s, int64_t n_settings,
                          const double* d_particles, int64_t ld_p, int64_t n_particles,
                          const double* d_weights, const double* d_moments, int32_t shifted,
                          void* d_ws, int64_t ws_bytes, int32_t iters, float* h_ms_avg,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OBE_HIP_H */
