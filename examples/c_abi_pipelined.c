/* The cycle pdf_update() -> opt_setting() with ONE host round trip, through the C ABI alone (INTEGRATION.md §2,
 * DESIGN.md §3 "Cycles that overlap"): the update is enqueued without waiting
 * (obe_bayes_update_model_moments_enqueue), the sweep over the updated cloud right behind it
 * (obe_sweep_utility with OBE_SWEEP_SPECULATIVE), and only then does the host wait for the update's sums.
 * The program runs that pair and, from the same starting weights, the two synchronous calls of
 * examples/c_abi_cycle.c, and exits non-zero unless both give the same bits.  A third run uses a resample
 * threshold that makes the update call for a resample: the sweep behind it must not have run.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_pipelined.c \
 *       -Loptbayesexpt_amd/lib -lobe_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/optbayesexpt_amd/lib -o /tmp/pipelined
 *   /tmp/pipelined [n_settings = 3000] [n_particles = 200000]
 *
 * Reference lines: obe_base.py:340-399 (pdf_update), particlepdf.py:236-258 (resample_test),
 * obe_base.py:733-756 -> :628-655 -> :463-489 (opt_setting, full-sweep form).
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "obe_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define OBE_OK(x) do { int rc_ = (x); if (rc_ != 0) { \
    fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, obe_last_error()); return 3; } } while (0)

static double uniform01(uint64_t* s) {      /* xorshift64*: any reproducible numbers will do */
    *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
    return (double)((*s * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

int main(int argc, char** argv) {
    const int64_t ns = argc > 1 ? atoll(argv[1]) : 3000, np = argc > 2 ? atoll(argv[2]) : 200000;
    enum { D = 3 };
    if (obe_abi_version() != OBE_ABI_VERSION) return 4;
    obe_model m = {OBE_MODEL_LORENTZ, 1, D, 0, 0, 1, {0.1}};
    OBE_OK(obe_model_validate(&m));

    double* h_settings = (double*)malloc(8 * ns);
    double* h_particles = (double*)malloc(8 * D * np);
    double* h_w0 = (double*)malloc(8 * np);
    uint64_t seed = 88172645463325252ull;
    double total = 0.0;
    for (int64_t s = 0; s < ns; ++s) h_settings[s] = 1.5 + 3.0 * (double)s / (double)(ns - 1);
    for (int64_t p = 0; p < np; ++p) {
        h_particles[p] = 2.0 + 2.0 * uniform01(&seed);               /* x0 */
        h_particles[np + p] = 1.0 + 2.0 * uniform01(&seed);          /* a */
        h_particles[2 * np + p] = 0.2 + 0.6 * uniform01(&seed);      /* b */
        h_w0[p] = 0.1 + uniform01(&seed);
        total += h_w0[p];
    }
    for (int64_t p = 0; p < np; ++p) h_w0[p] /= total;

    const int64_t ws_bytes = obe_workspace_bytes(np, ns, m.n_channels, D), mom_len = obe_moments_len(D);
    double *d_settings, *d_particles, *d_weights, *d_moments, *d_yvar, *d_utility, *d_noise;
    void* d_ws;
    HIP_OK(hipMalloc((void**)&d_settings, 8 * ns));
    HIP_OK(hipMalloc((void**)&d_particles, 8 * D * np));
    HIP_OK(hipMalloc((void**)&d_weights, 8 * np));
    HIP_OK(hipMalloc((void**)&d_moments, 8 * mom_len));
    HIP_OK(hipMalloc((void**)&d_yvar, 8 * ns));
    HIP_OK(hipMalloc((void**)&d_utility, 8 * ns));
    HIP_OK(hipMalloc((void**)&d_noise, 8));
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    HIP_OK(hipMemcpy(d_settings, h_settings, 8 * ns, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_particles, h_particles, 8 * D * np, hipMemcpyHostToDevice));
    const double sigma = 2.5, noise_var = sigma * sigma;      /* a mild measurement: N_eff / N stays above 0.5 */
    HIP_OK(hipMemcpy(d_noise, &noise_var, 8, hipMemcpyHostToDevice));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    /* page-locked landing zones: the kernels write their results there and the host watches the words */
    double *upd, *res;      /* upd: [0] sum w L, [1] sum w'^2, [2..) K3 block, [4 + 4 D] resample decision */
    HIP_OK(hipHostMalloc((void**)&upd, 8 * (5 + 4 * D), hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void**)&res, 8 * 4, hipHostMallocDefault));     /* best, index bits, kappa */
    double setting[OBE_MAX_SETDIMS] = {3.05}, y[OBE_MAX_CHANNELS] = {2.1}, s[OBE_MAX_CHANNELS] = {sigma};
    const double no_choke = NAN;
    double* h_a = (double*)malloc(8 * (np + ns));      /* weights' and utility of one run, for the comparison */
    double* h_b = (double*)malloc(8 * (np + ns));
    double sums_a[2], best_a[3];

    /* 1. the synchronous pair */
    HIP_OK(hipMemcpy(d_weights, h_w0, 8 * np, hipMemcpyHostToDevice));
    OBE_OK(obe_bayes_update_model_moments(&m, d_particles, np, np, d_weights, setting, y, s, NULL, 1, no_choke,
                                          d_moments, d_ws, ws_bytes, upd, stream));
    OBE_OK(obe_sweep_utility(&m, d_settings, ns, ns, d_particles, np, np, d_weights, NULL, 0, d_moments,
                             OBE_SWEEP_SHIFTED, d_noise, 0, NULL, 1.0, d_yvar, d_utility, &res[0], (int64_t*)&res[1],
                             &res[2], d_ws, ws_bytes, stream));
    HIP_OK(hipStreamSynchronize(stream));
    memcpy(sums_a, upd, 16);
    memcpy(best_a, res, 24);
    int64_t best_index;
    memcpy(&best_index, &res[1], 8);
    HIP_OK(hipMemcpy(h_a, d_weights, 8 * np, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_a + np, d_utility, 8 * ns, hipMemcpyDeviceToHost));

    /* 2. the same cycle enqueued: nothing is waited for until both calls are out */
    HIP_OK(hipMemcpy(d_weights, h_w0, 8 * np, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_utility, 0, 8 * ns));
    OBE_OK(obe_bayes_update_model_moments_enqueue(&m, d_particles, np, np, d_weights, setting, y, s, NULL, 1, no_choke,
                                                  d_moments, d_ws, ws_bytes, upd, 1 /* auto_resample */,
                                                  0.5 /* resample_threshold */, stream));
    OBE_OK(obe_sweep_utility(&m, d_settings, ns, ns, d_particles, np, np, d_weights, NULL, 0, d_moments,
                             OBE_SWEEP_SHIFTED | OBE_SWEEP_SPECULATIVE, d_noise, 0, NULL, 1.0, d_yvar, d_utility,
                             &res[0], (int64_t*)&res[1], &res[2], d_ws, ws_bytes, stream));
    OBE_OK(obe_host_words_wait(upd, 5 + 4 * D, stream));           /* the update's sums ... */
    const double n_eff = 1.0 / upd[1];
    const int resample_due = n_eff < 0.1 * (double)np || n_eff / (double)np < 0.5;   /* particlepdf.py:236-258 */
    if ((upd[4 + 4 * D] != 0.0) != resample_due) {
        fprintf(stderr, "the device's resample decision (%g) is not the host's (%d)\n", upd[4 + 4 * D], resample_due);
        return 5;
    }
    if (resample_due) {
        fprintf(stderr, "this measurement was meant to be mild (N_eff / N = %.3f)\n", n_eff / (double)np);
        return 5;
    }
    OBE_OK(obe_host_words_wait(res, 3, stream));                   /* ... then the sweep's result */
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(h_b, d_weights, 8 * np, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_b + np, d_utility, 8 * ns, hipMemcpyDeviceToHost));
    if (memcmp(sums_a, upd, 16) || memcmp(best_a, res, 24) || memcmp(h_a, h_b, 8 * (np + ns))) {
        fprintf(stderr, "the enqueued cycle differs from the synchronous one\n");
        return 6;
    }

    /* 3. an update that calls for a resample (threshold above 1): the sweep behind it does nothing */
    HIP_OK(hipMemcpy(d_weights, h_w0, 8 * np, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_utility, 0, 8 * ns));
    OBE_OK(obe_bayes_update_model_moments_enqueue(&m, d_particles, np, np, d_weights, setting, y, s, NULL, 1, no_choke,
                                                  d_moments, d_ws, ws_bytes, upd, 1, 1.5, stream));
    OBE_OK(obe_sweep_utility(&m, d_settings, ns, ns, d_particles, np, np, d_weights, NULL, 0, d_moments,
                             OBE_SWEEP_SHIFTED | OBE_SWEEP_SPECULATIVE, d_noise, 0, NULL, 1.0, d_yvar, d_utility,
                             &res[0], (int64_t*)&res[1], &res[2], d_ws, ws_bytes, stream));
    OBE_OK(obe_host_words_wait(upd, 5 + 4 * D, stream));
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(h_b + np, d_utility, 8 * ns, hipMemcpyDeviceToHost));
    int untouched = upd[4 + 4 * D] == 1.0;
    for (int64_t k = 0; k < ns && untouched; ++k) untouched = h_b[np + k] == 0.0;
    if (!untouched) {
        fprintf(stderr, "a sweep behind a resampling update ran\n");
        return 7;
    }
    printf("enqueued cycle identical to the synchronous one (best setting %lld, utility %.6e, N_eff / N %.3f); "
           "behind a resampling update the sweep did nothing\n", (long long)best_index, best_a[0], n_eff / (double)np);
    return 0;
}
