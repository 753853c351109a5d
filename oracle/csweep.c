/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Plain-C restatement of the full utility sweep for the Lorentzian family, used (1) as a
 * second, independent oracle for the NumPy restatement (tests/test_oracle_golden.py) and
 * (2) as the all-cores CPU baseline of bench.py (OpenMP over settings).  It follows the
 * reference's arithmetic: the model of demos/find_peak/sequentialLorentzian.py:53-75
 * (K peaks: SURVEY.md §8d) evaluated for every (setting, particle), and the variance over the
 * particle axis computed in two passes like np.var (obe_base.py:488) — weighted by the
 * particle weights.
 *
 *   gcc -O3 -mavx2 -ffp-contract=off -fopenmp -shared -fPIC oracle/csweep.c -o oracle/_build/libcsweep.so -lm
 */
#include <math.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int csweep_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* particles: (K + 2 [+ extra]) rows of length n_p, row stride ld: x0_1..x0_K, a, b */
static inline double model(double x, const double* particles, long ld, long p, int n_peaks, double d) {
    const double a = particles[(long)n_peaks * ld + p];
    double y = particles[(long)(n_peaks + 1) * ld + p];
    for (int k = 0; k < n_peaks; ++k) {
        const double t = (x - particles[(long)k * ld + p]) / d;
        y = y + a / (t * t + 1.0);
    }
    return y;
}

/* yvar[s] = sum_p w_p (y_sp - ybar_s)^2 / sum_p w_p
 *
 * Settings are taken CSWEEP_SB at a time so that one pass over the cloud serves CSWEEP_SB settings
 * (the cloud is 25-42 MB at the BASELINE sizes: one setting per pass is bound by the host's memory
 * system, not its dividers) and the compiler can vectorise ACROSS the settings of a block.  Every
 * setting keeps its own accumulators and sees the particles in index order: per setting the operation
 * sequence is exactly that of the scalar loop `for p: acc += w[p] * model(x, p)` (no reassociation;
 * built with -ffp-contract=off, so no fused multiply-adds either), i.e. np.average / np.var's
 * arithmetic up to NumPy's pairwise summation. */
#define CSWEEP_SB 8
void csweep_lorentz_yvar(const double* settings, long n_s, const double* particles, long ld, long n_p,
                         const double* weights, int n_peaks, double d, double* yvar) {
    double wsum = 0.0;
    for (long p = 0; p < n_p; ++p) wsum += weights[p];
    const double* pa = particles + (long)n_peaks * ld;
    const double* pb = particles + (long)(n_peaks + 1) * ld;
#pragma omp parallel for schedule(dynamic, 1)
    for (long s0 = 0; s0 < n_s; s0 += CSWEEP_SB) {
        double x[CSWEEP_SB], acc[CSWEEP_SB], ybar[CSWEEP_SB], v[CSWEEP_SB], y[CSWEEP_SB];
        for (int j = 0; j < CSWEEP_SB; ++j) {
            x[j] = settings[s0 + j < n_s ? s0 + j : n_s - 1];
            acc[j] = 0.0;
            v[j] = 0.0;
        }
        for (long p = 0; p < n_p; ++p) {
            const double a = pa[p], b = pb[p], w = weights[p];
            for (int j = 0; j < CSWEEP_SB; ++j) y[j] = b;
            for (int k = 0; k < n_peaks; ++k) {
                const double x0 = particles[(long)k * ld + p];
                for (int j = 0; j < CSWEEP_SB; ++j) {
                    const double t = (x[j] - x0) / d;
                    y[j] = y[j] + a / (t * t + 1.0);
                }
            }
            for (int j = 0; j < CSWEEP_SB; ++j) acc[j] += w * y[j];
        }
        for (int j = 0; j < CSWEEP_SB; ++j) ybar[j] = acc[j] / wsum;
        for (long p = 0; p < n_p; ++p) {
            const double a = pa[p], b = pb[p], w = weights[p];
            for (int j = 0; j < CSWEEP_SB; ++j) y[j] = b;
            for (int k = 0; k < n_peaks; ++k) {
                const double x0 = particles[(long)k * ld + p];
                for (int j = 0; j < CSWEEP_SB; ++j) {
                    const double t = (x[j] - x0) / d;
                    y[j] = y[j] + a / (t * t + 1.0);
                }
            }
            for (int j = 0; j < CSWEEP_SB; ++j) {
                const double dev = y[j] - ybar[j];
                v[j] += w * dev * dev;
            }
        }
        for (int j = 0; j < CSWEEP_SB && s0 + j < n_s; ++j) yvar[s0 + j] = v[j] / wsum;
    }
}

/* Bayes update with a known sigma (obe_base.py:385-394, 269-271; particlepdf.py:136-139):
 * returns sum of w*L; weights are normalised in place; *sum_w2 = sum of w'^2 */
double csweep_lorentz_update(double x, double y_meas, double sigma, const double* particles, long ld, long n_p,
                             double* weights, int n_peaks, double d, double* sum_w2) {
    double total = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (long p = 0; p < n_p; ++p) {
        const double z = (model(x, particles, ld, p, n_peaks, d) - y_meas) / sigma;
        weights[p] = weights[p] * (exp(-(z * z) / 2.0) / sigma);
        total += weights[p];
    }
    double s2 = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s2)
    for (long p = 0; p < n_p; ++p) {
        weights[p] = weights[p] / total;
        s2 += weights[p] * weights[p];
    }
    *sum_w2 = s2;
    return total;
}
