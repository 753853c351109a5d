/* One measurement cycle of the hot path through the C ABI alone — no Python, no torch:
 * the binding a C / C++ / cgo / JNI host would write.
 *
 *   hipcc examples/c_abi_cycle.c -Iinclude -Loptbayesexpt_amd/lib -lobe_hip \
 *         -Wl,-rpath,$PWD/optbayesexpt_amd/lib -o /tmp/c_abi_cycle
 *   /tmp/c_abi_cycle in.bin out.bin
 *
 * in.bin  (float64 / int64, little endian): n_settings, n_particles (int64); d, y_meas, sigma;
 *         settings[n_settings]; particles[3][n_particles] (x0, a, b rows); weights[n_particles]
 * out.bin: best index (int64); best utility, kappa, sum of w*L, sum of w'^2 (float64);
 *          utility[n_settings]; weights'[n_particles]; mean[3]; std[3]
 *
 * The calls are those of OptBayesExpt.opt_setting() (obe_base.py:733-756 ->
 * utility_variance :628-655 -> yvar_from_parameter_draws :463-489, full-sweep form) followed by
 * pdf_update() (obe_base.py:340-399) and mean()/std() (particlepdf.py:173-214).
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "obe_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define OBE_OK(x) do { int rc_ = (x); if (rc_ != 0) { \
    fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, obe_last_error()); return 3; } } while (0)

static int read_exact(FILE* f, void* p, size_t bytes) { return fread(p, 1, bytes, f) == bytes ? 0 : 1; }

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
        return 1;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int64_t ns = 0, np = 0;
    double head[3];
    if (read_exact(f, &ns, 8) || read_exact(f, &np, 8) || read_exact(f, head, 24)) return 1;
    const double d = head[0], y_meas = head[1], sigma = head[2];
    const int D = 3;
    double* h_settings = (double*)malloc(8 * ns);
    double* h_particles = (double*)malloc(8 * D * np);
    double* h_weights = (double*)malloc(8 * np);
    if (read_exact(f, h_settings, 8 * ns) || read_exact(f, h_particles, 8 * D * np) || read_exact(f, h_weights, 8 * np))
        return 1;
    fclose(f);

    if (obe_abi_version() != OBE_ABI_VERSION) return 4;
    obe_model m = {OBE_MODEL_LORENTZ, 1 /* one peak */, D, 0, 0, 1, {d}};
    OBE_OK(obe_model_validate(&m));

    /* device memory is the caller's: the library never allocates */
    const int64_t ws_bytes = obe_workspace_bytes(np, ns, m.n_channels, D);
    const int64_t mom_len = obe_moments_len(D);
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
    HIP_OK(hipMemcpy(d_weights, h_weights, 8 * np, hipMemcpyHostToDevice));
    const double noise_var = sigma * sigma;           /* yvar_noise_model(): default_noise_std ** 2 */
    HIP_OK(hipMemcpy(d_noise, &noise_var, 8, hipMemcpyHostToDevice));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    /* opt_setting(): moments (variance shift + normaliser), sweep, utility, first-maximum argmax */
    double* h_mom = (double*)malloc(8 * mom_len);
    OBE_OK(obe_moments(d_particles, np, D, np, d_weights, 0, d_moments, h_mom, d_ws, ws_bytes, stream));
    double best = 0.0, kappa = 0.0;
    int64_t best_idx = -1;
    OBE_OK(obe_sweep_utility(&m, d_settings, ns, ns, d_particles, np, np, d_weights, NULL, 0, d_moments,
                             OBE_SWEEP_SHIFTED, d_noise, 0, NULL, 1.0, d_yvar, d_utility, &best, &best_idx, &kappa,
                             d_ws, ws_bytes, stream));

    /* pdf_update((x_best, y_meas, sigma)) */
    double setting[OBE_MAX_SETDIMS] = {h_settings[best_idx]};
    double y[OBE_MAX_CHANNELS] = {y_meas}, s[OBE_MAX_CHANNELS] = {sigma};
    double upd[4] = {0, 0, 0, 0};
    const double no_choke = 0.0 / 0.0;                /* NaN = no choke (obe_base.py:458-459) */
    OBE_OK(obe_bayes_update_model(&m, d_particles, np, np, d_weights, setting, y, s, NULL, 1, no_choke, d_ws,
                                  ws_bytes, upd, stream));

    /* mean() / std() of the posterior */
    OBE_OK(obe_moments(d_particles, np, D, np, d_weights, 0, d_moments, h_mom, d_ws, ws_bytes, stream));

    double* h_utility = (double*)malloc(8 * ns);
    HIP_OK(hipMemcpy(h_utility, d_utility, 8 * ns, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_weights, d_weights, 8 * np, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f) return 1;
    double scal[4] = {best, kappa, upd[0], upd[1]};
    fwrite(&best_idx, 8, 1, f);
    fwrite(scal, 8, 4, f);
    fwrite(h_utility, 8, ns, f);
    fwrite(h_weights, 8, np, f);
    fwrite(h_mom + 2, 8, D, f);                       /* mean */
    fwrite(h_mom + 2 + 3 * D, 8, D, f);               /* std */
    fclose(f);
    printf("best setting %lld (x = %.6f), utility %.6e, kappa %.3g, sum w*L %.6e, N_eff %.1f\n", (long long)best_idx,
           h_settings[best_idx], best, kappa, upd[0], 1.0 / upd[1]);
    return 0;
}
