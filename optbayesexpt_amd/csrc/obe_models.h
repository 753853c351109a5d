// Device model registry: the HIP counterpart of the reference's Python
// `model_function(settings, parameters, constants)` (obe_base.py:50-72).
//
// Each model provides two evaluations of the same formula:
//   eval()      — the exact NumPy operation sequence of the corresponding demo
//                 function, one correctly-rounded op at a time (no FMA contraction);
//                 used by the HBM-bound kernels (Bayes update, eval_over_*), where
//                 it costs nothing and gives reference-identical bits.
//   sweep_eval() — the flop-bound sweep form, for the SPT settings one lane owns at
//                 once: per-setting and per-particle terms are hoisted (prep_setting /
//                 pack), the SPT divisions share ONE v_rcp_f64 + Newton (Montgomery's
//                 batch inversion), and products/sums are fused.  It returns
//                 sqrt(w) * (y - offset): the weight is folded into the packed particle
//                 (one multiply per particle instead of one per evaluation) and the
//                 offset is a particle-independent constant (the mean background),
//                 which leaves the variance unchanged and removes the cancellation.
#pragma once

#include <type_traits>

#include "obe_common.h"

namespace obe {

// 1/q for the root of a batch inversion in the sweep.  v_rcp_f64 seeds 2^-24.4 relative (measured on
// gfx950, tools/microbench_fp64.hip); ONE Newton step  r (1 + e),  e = 1 - q r,  takes that to
// e^2 <= 2^-48.8 = 2e-15.  Every inverse of the batch then carries the same relative error in [-2e-15, 0]:
// the sweep's variances move by <= 4e-15 relative (1e-10 is the parity bar; the random part of the error
// averages down by a further sqrt(N) over the particles).  The step sits on the one narrow dependent chain
// of the group (product tree -> v_rcp_f64 -> correction -> back-substitution), which the wide phases of
// the neighbouring pair have to cover: dropping the third FMA of the cubically convergent form
// r (1 + e + e^2) (2^-73) shortened that chain and took 1.6 % off the c3 sweep (13.75 -> 13.52 ms on one
// box) where the issue slot alone accounts for 0.85 %.  -DOBE_RCP_CUBIC restores the 1-ulp form.
__device__ __forceinline__ double fast_rcp(double q) {
    const double r = __builtin_amdgcn_rcp(q);
    const double e = fma(-q, r, 1.0);
#ifdef OBE_RCP_CUBIC
    const double t = fma(e, e, e);
    return fma(r, t, r);
#else
    return fma(r, e, r);
#endif
}

// The 1-ulp form, r (1 + e + e^2), where a reciprocal is not shared (N = 1: one setting per lane, i.e. the sweeps
// of at most a few hundred settings, whose time is launch latency): with a handful of draws nothing averages the
// one-sided 2e-15 of fast_rcp down, and at kappa = (mean of y)^2 / var ~ 1e9 (a cloud of 2 particles, 10 draws:
// tools/fuzz_parity.py 300 5, case 185) it showed as 1.7e-10 in one utility value.
__device__ __forceinline__ double exact_rcp(double q) {
    const double r = __builtin_amdgcn_rcp(q);
    const double e = fma(-q, r, 1.0);
    return fma(r, fma(e, e, e), r);
}

// Batch inversion: r[j] = 1/q[j] for N values from one reciprocal of their product
// (3(N-1) multiplies + 1 rcp instead of N rcp + 4N Newton FMAs).  q >= 1 on every
// direct caller, so the product cannot underflow; it overflows only beyond q ~ 1e77.
// Returns the reciprocal of the whole product (batch_rcp_guarded's range check).
template <int N>
__device__ __forceinline__ double batch_rcp(const double (&q)[N], double (&r)[N]) {
    if constexpr (N == 1) {
        r[0] = exact_rcp(q[0]);
        return r[0];
    } else if constexpr (N == 2) {
        const double inv = fast_rcp(q[0] * q[1]);
        r[0] = inv * q[1];
        r[1] = inv * q[0];
        return inv;
    } else if constexpr (N == 4) {
        const double p01 = q[0] * q[1], p23 = q[2] * q[3];
        const double inv = fast_rcp(p01 * p23);
        const double i01 = inv * p23, i23 = inv * p01;
        r[0] = i01 * q[1];
        r[1] = i01 * q[0];
        r[2] = i23 * q[3];
        r[3] = i23 * q[2];
        return inv;
    } else {
        static_assert(N == 8, "batch_rcp: N must be 1, 2, 4 or 8");
        const double p01 = q[0] * q[1], p23 = q[2] * q[3], p45 = q[4] * q[5], p67 = q[6] * q[7];
        const double p03 = p01 * p23, p47 = p45 * p67;
        const double inv = fast_rcp(p03 * p47);
        const double i03 = inv * p47, i47 = inv * p03;
        const double i01 = i03 * p23, i23 = i03 * p01, i45 = i47 * p67, i67 = i47 * p45;
        r[0] = i01 * q[1];
        r[1] = i01 * q[0];
        r[2] = i23 * q[3];
        r[3] = i23 * q[2];
        r[4] = i45 * q[5];
        r[5] = i45 * q[4];
        r[6] = i67 * q[7];
        r[7] = i67 * q[6];
        return inv;
    }
}

// Batch inversion of two groups that share one reciprocal, each result scaled by its group's
// factor: r[j] = s_lo / q[j] (j < N/2), s_hi / q[j] (j >= N/2).  The factors enter at the
// top of the back-substitution (two multiplies) instead of once per element afterwards.
template <int N>
__device__ __forceinline__ void batch_rcp_scaled2(const double (&q)[N], double s_lo, double s_hi, double (&r)[N]) {
    if constexpr (N == 2) {
        const double inv = fast_rcp(q[0] * q[1]);
        r[0] = inv * (q[1] * s_lo);
        r[1] = inv * (q[0] * s_hi);
    } else if constexpr (N == 4) {
        const double p01 = q[0] * q[1], p23 = q[2] * q[3];
        const double inv = fast_rcp(p01 * p23);
        const double i01 = inv * (p23 * s_lo), i23 = inv * (p01 * s_hi);
        r[0] = i01 * q[1];
        r[1] = i01 * q[0];
        r[2] = i23 * q[3];
        r[3] = i23 * q[2];
    } else {
        static_assert(N == 8, "batch_rcp_scaled2: N must be 2, 4 or 8");
        const double p01 = q[0] * q[1], p23 = q[2] * q[3], p45 = q[4] * q[5], p67 = q[6] * q[7];
        const double p03 = p01 * p23, p47 = p45 * p67;
        // (measured and rejected: the Newton step applied to the two half inverses instead of the root —
        // one slot more, one dependent operation less: +1.6 % at c3, profiles/r03_sweep_accuracy.txt)
        const double inv = fast_rcp(p03 * p47);
        const double i03 = inv * (p47 * s_lo), i47 = inv * (p03 * s_hi);
        const double i01 = i03 * p23, i23 = i03 * p01, i45 = i47 * p67, i67 = i47 * p45;
        r[0] = i01 * q[1];
        r[1] = i01 * q[0];
        r[2] = i23 * q[3];
        r[3] = i23 * q[2];
        r[4] = i45 * q[5];
        r[5] = i45 * q[4];
        r[6] = i67 * q[7];
        r[7] = i67 * q[6];
    }
}

// 1/b with IEEE behaviour at the edges (b = 0 -> inf, b = inf -> 0, NaN -> NaN): the raw
// v_rcp_f64 result is returned whenever the correction step produced a NaN.
__device__ __forceinline__ double guarded_rcp(double b) {
    const double r0 = __builtin_amdgcn_rcp(b);
    const double e = fma(-b, r0, 1.0);
    const double r = fma(r0, fma(e, e, e), r0);
    return r == r ? r : r0;
}

// Batch inversions for generated expression models, whose denominators are not known to be
// >= 1.  Straight-line code: the batch runs over the N/2 pair products and is accepted when
// their magnitudes sum to < 1e30 (no overflow, inf or NaN) and the reciprocal of the whole
// product is < 1e200 in magnitude — then every partial product inside the tree lies in
// (1e-290, 1e120): no zero, no subnormal that would lose bits.  Otherwise the root inverse is
// replaced by NaN, which poisons every result of the batch and, through the moments, the
// variance of the setting: the sweep reports kappa = NaN and the host repeats it with the
// model's sweep_eval_safe() (one guarded IEEE reciprocal per element), so a branch never
// sits in the hot loop.
__device__ __forceinline__ double poison_unless(bool ok, double v) { return ok ? v : __builtin_nan(""); }

// pair inverses ip[h] = s / (q[2h] q[2h+1]); the caller multiplies by the sibling (sibling_of)
template <int N>
__device__ __forceinline__ void batch_div_poisoned(const double (&q)[N], double s, double (&ip)[(N + 1) / 2]) {
    if constexpr (N == 1) {
        ip[0] = s * guarded_rcp(q[0]);
    } else if constexpr (N == 2) {
        const double pp = q[0] * q[1];
        const double inv = fast_rcp(pp);
        ip[0] = poison_unless(fabs(pp) < 1e30 && fabs(inv) < 1e200, inv) * s;
    } else if constexpr (N == 4) {
        const double p0 = q[0] * q[1], p1 = q[2] * q[3];
        const double inv = fast_rcp(p0 * p1);
        const double is = poison_unless(fabs(p0) + fabs(p1) < 1e30 && fabs(inv) < 1e200, inv) * s;
        ip[0] = is * p1;
        ip[1] = is * p0;
    } else {
        static_assert(N == 8, "batch_div_poisoned: N must be 1, 2, 4 or 8");
        const double p0 = q[0] * q[1], p1 = q[2] * q[3], p2 = q[4] * q[5], p3 = q[6] * q[7];
        const double p01 = p0 * p1, p23 = p2 * p3;
        const double inv = fast_rcp(p01 * p23);
        const double mag = (fabs(p0) + fabs(p1)) + (fabs(p2) + fabs(p3));
        const double is = poison_unless(mag < 1e30 && fabs(inv) < 1e200, inv) * s;
        const double i01 = is * p23, i23 = is * p01;
        ip[0] = i01 * p1;
        ip[1] = i01 * p0;
        ip[2] = i23 * p3;
        ip[3] = i23 * p2;
    }
}
template <int N>
__device__ __forceinline__ double sibling_of(const double (&q)[N], int j) {
    if constexpr (N == 1) return 1.0;
    else return q[j ^ 1];
}

// Two particles' denominators (N each, N even) through ONE inversion tree: pair inverses
// ipa[h] = sa / (qa[2h] qa[2h+1]), ipb likewise.  The range check is tighter than for one
// particle (sum of |pair products| < 1e18, total inverse < 1e150) because the root spans twice
// as many factors; every partial product then lies in (1e-276, 1e72).
template <int N>
__device__ __forceinline__ void batch_div_poisoned2(const double (&qa)[N], const double (&qb)[N], double sa, double sb,
                                                    double (&ipa)[N / 2], double (&ipb)[N / 2]) {
    static_assert(N == 2 || N == 4 || N == 8, "batch_div_poisoned2: N must be 2, 4 or 8");
    if constexpr (N == 2) {
        const double a0 = qa[0] * qa[1], b0 = qb[0] * qb[1];
        const double inv = fast_rcp(a0 * b0);
        const double ok = poison_unless(fabs(a0) + fabs(b0) < 1e18 && fabs(inv) < 1e150, inv);
        ipa[0] = ok * (b0 * sa);
        ipb[0] = ok * (a0 * sb);
    } else if constexpr (N == 4) {
        const double a0 = qa[0] * qa[1], a1 = qa[2] * qa[3], b0 = qb[0] * qb[1], b1 = qb[2] * qb[3];
        const double pa = a0 * a1, pb = b0 * b1;
        const double inv = fast_rcp(pa * pb);
        const double mag = (fabs(a0) + fabs(a1)) + (fabs(b0) + fabs(b1));
        const double ok = poison_unless(mag < 1e18 && fabs(inv) < 1e150, inv);
        const double ia = ok * (pb * sa), ib = ok * (pa * sb);
        ipa[0] = ia * a1;
        ipa[1] = ia * a0;
        ipb[0] = ib * b1;
        ipb[1] = ib * b0;
    } else {
        const double a0 = qa[0] * qa[1], a1 = qa[2] * qa[3], a2 = qa[4] * qa[5], a3 = qa[6] * qa[7];
        const double b0 = qb[0] * qb[1], b1 = qb[2] * qb[3], b2 = qb[4] * qb[5], b3 = qb[6] * qb[7];
        const double a01 = a0 * a1, a23 = a2 * a3, b01 = b0 * b1, b23 = b2 * b3;
        const double pa = a01 * a23, pb = b01 * b23;
        const double inv = fast_rcp(pa * pb);
        const double mag = ((fabs(a0) + fabs(a1)) + (fabs(a2) + fabs(a3))) + ((fabs(b0) + fabs(b1)) + (fabs(b2) + fabs(b3)));
        const double ok = poison_unless(mag < 1e18 && fabs(inv) < 1e150, inv);
        const double ia = ok * (pb * sa), ib = ok * (pa * sb);
        const double ia01 = ia * a23, ia23 = ia * a01, ib01 = ib * b23, ib23 = ib * b01;
        ipa[0] = ia01 * a1;
        ipa[1] = ia01 * a0;
        ipa[2] = ia23 * a3;
        ipa[3] = ia23 * a2;
        ipb[0] = ib01 * b1;
        ipb[1] = ib01 * b0;
        ipb[2] = ib23 * b3;
        ipb[3] = ib23 * b2;
    }
}
template <int N>
__device__ __forceinline__ void batch_rcp_poisoned2(const double (&qa)[N], const double (&qb)[N], double (&ra)[N],
                                                    double (&rb)[N]) {
    double ipa[N / 2], ipb[N / 2];
    batch_div_poisoned2<N>(qa, qb, 1.0, 1.0, ipa, ipb);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        ra[j] = ipa[j / 2] * qa[j ^ 1];
        rb[j] = ipb[j / 2] * qb[j ^ 1];
    }
}

// r[j] = 1 / q[j]
template <int N>
__device__ __forceinline__ void batch_rcp_poisoned(const double (&q)[N], double (&r)[N]) {
    double ip[(N + 1) / 2];
    batch_div_poisoned<N>(q, 1.0, ip);
#pragma unroll
    for (int j = 0; j < N; ++j) r[j] = ip[j / 2] * sibling_of<N>(q, j);
}

// Pair inverses ip[h] = s / (q[2h] q[2h+1]) for denominators known to be >= 1 (products of
// Lorentzian denominators): nothing can underflow, so the only range check is on the root product.
// Beyond 1e250 (where s * inv could leave the normal range) the root inverse is replaced by NaN; the
// NaN reaches the setting's variance, the sweep reports kappa = NaN and the host repeats it with the
// model's sweep_eval_safe().  An inf or NaN product fails the comparison as well.
template <int N>
__device__ __forceinline__ void batch_div_ge1(const double (&q)[N], double s, double (&ip)[(N + 1) / 2]) {
    if constexpr (N == 1) {
        ip[0] = s * exact_rcp(q[0]);
    } else if constexpr (N == 2) {
        const double pp = q[0] * q[1];
        ip[0] = poison_unless(pp < 1e250, fast_rcp(pp)) * s;
    } else if constexpr (N == 4) {
        const double p0 = q[0] * q[1], p1 = q[2] * q[3];
        const double root = p0 * p1;
        const double is = poison_unless(root < 1e250, fast_rcp(root)) * s;
        ip[0] = is * p1;
        ip[1] = is * p0;
    } else {
        static_assert(N == 8, "batch_div_ge1: N must be 1, 2, 4 or 8");
        const double p0 = q[0] * q[1], p1 = q[2] * q[3], p2 = q[4] * q[5], p3 = q[6] * q[7];
        const double p01 = p0 * p1, p23 = p2 * p3;
        const double root = p01 * p23;
        const double is = poison_unless(root < 1e250, fast_rcp(root)) * s;
        const double i01 = is * p23, i23 = is * p01;
        ip[0] = i01 * p1;
        ip[1] = i01 * p0;
        ip[2] = i23 * p3;
        ip[3] = i23 * p2;
    }
}

// sum_{k in [LO, HI)} 1/q[k] as ONE fraction num/den (all q > 0, so num and den are sums and
// products of positive numbers: no cancellation).  (na/da) + (nb/db) = (na db + nb da)/(da db):
// 3 operations per combination, 2 for a leaf pair, 2 for a fraction plus a single term — 2K
// operations for K terms, against the ~3.4 a separate batched inversion costs per term.
template <int LO, int HI>
__device__ __forceinline__ void sum_of_reciprocals(const double* q, double& num, double& den) {
    if constexpr (HI - LO == 1) {
        num = 1.0;
        den = q[LO];
    } else if constexpr (HI - LO == 2) {
        num = q[LO] + q[LO + 1];
        den = q[LO] * q[LO + 1];
    } else if constexpr (HI - LO == 3) {
        const double n2 = q[LO] + q[LO + 1], d2 = q[LO] * q[LO + 1];
        num = fma(n2, q[LO + 2], d2);
        den = d2 * q[LO + 2];
    } else {
        constexpr int MID = LO + (HI - LO + 1) / 2;
        double na, da, nb, db;
        sum_of_reciprocals<LO, MID>(q, na, da);
        sum_of_reciprocals<MID, HI>(q, nb, db);
        num = fma(na, db, nb * da);
        den = da * db;
    }
}

// ---- elementary functions for the inner level of generated sweep forms ----
// The exact forms (Bayes update, eval_over_*, sweep_eval_safe, everything hoisted to the setting
// or particle level) call ocml; the inner level of the flop-bound sweep uses these: accurate to
// ~2 ulp on the ranges stated, NaN outside them — a NaN reaches the setting's variance and the
// host repeats the sweep with the safe twin (OBE_SWEEP_SAFE), as for the batched divisions.

// sin r on |r| <= pi/2 (+1 %): odd Taylor polynomial through r^21 (truncation 2e-18)
__device__ __forceinline__ double sin_poly(double r) {
    const double r2 = r * r;
    double p = 0x1.71b8ef6dcf572p-66;
    p = fma(p, r2, -0x1.2f49b46814157p-57);
    p = fma(p, r2, 0x1.952c77030ad4ap-49);
    p = fma(p, r2, -0x1.ae7f3e733b81fp-41);
    p = fma(p, r2, 0x1.6124613a86d09p-33);
    p = fma(p, r2, -0x1.ae64567f544e4p-26);
    p = fma(p, r2, 0x1.71de3a556c734p-19);
    p = fma(p, r2, -0x1.a01a01a01a01ap-13);
    p = fma(p, r2, 0x1.1111111111111p-7);
    p = fma(p, r2, -0x1.5555555555555p-3);
    return fma(r * r2, p, r);
}
// sin x = (-1)^k sin(x - k pi): two-constant Cody-Waite reduction with FMA (the product k*pi_hi is
// exact inside the FMA, so the reduced argument is good to 1e-16 absolute for |x| < 1e9)
__device__ __forceinline__ double fast_sin(double x) {
    const double k = rint(x * 0x1.45f306dc9c883p-2);
    double r = fma(-k, 0x1.921fb54442d18p+1, x);
    r = fma(-k, 0x1.1a62633145c07p-53, r);
    const double s = sin_poly(r);
    const double v = (static_cast<int>(k) & 1) ? -s : s;
    return fabs(x) < 1e9 ? v : __builtin_nan("");
}
// cos x = sin(x + pi/2)
__device__ __forceinline__ double fast_cos(double x) {
    const double k = rint(fma(x, 0x1.45f306dc9c883p-2, 0.5));
    double r = fma(-k, 0x1.921fb54442d18p+1, x) + 0x1.921fb54442d18p+0;
    r = r + fma(-k, 0x1.1a62633145c07p-53, 0x1.1a62633145c07p-54);
    const double s = sin_poly(r);
    const double v = (static_cast<int>(k) & 1) ? -s : s;
    return fabs(x) < 1e9 ? v : __builtin_nan("");
}
// sqrt a = a * rsq(a): v_rsq_f64 (2^-23) + two Newton steps, then one correction of the root.
// 0 -> 0, negative -> NaN, +inf -> NaN (repeat with the safe twin).
__device__ __forceinline__ double fast_sqrt(double a) {
    double y = __builtin_amdgcn_rsq(a);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = fma(-(a * y), y, 1.0);
        y = fma(y * 0.5, e, y);
    }
    double s = a * y;
    s = fma(fma(-s, s, a), y * 0.5, s);
    return a == 0.0 ? 0.0 : s;
}
// hypot(a, b) with the larger magnitude scaled to [0.5, 1): no overflow / underflow of the squares
__device__ __forceinline__ double fast_hypot(double a, double b) {
    const double m = fmax(fabs(a), fabs(b));
    const int e = __builtin_amdgcn_frexp_exp(m);
    const double a1 = ldexp(a, -e), b1 = ldexp(b, -e);
    const double h = fast_sqrt(fma(a1, a1, b1 * b1));
    return m == 0.0 ? 0.0 : ldexp(h, e);
}

// ---------------------------------------------------------------------------
// y = b + sum_{k<K} a / (((x - x0_k)/d)^2 + 1)
// params: x0_0..x0_{K-1}, a, b [, anything]; const d
// demos/find_peak/sequentialLorentzian.py:53-75 (K = 1)
template <int K>
struct Lorentz {
    static constexpr int NS = 1, NC = 1, NREAD = K + 2, NCONST = 1;
    static constexpr int NXS = 1;        // prepared setting: x/d
    static constexpr int NPK = K + 2;    // packed particle: x0_k/d ..., sw*a, sw*(b - bbar)
    static constexpr int kSweepCost = K; // evaluation time relative to the one-peak model (grid planning)

    __device__ static void eval(const double* x, const ParamRef& th, const obe_model& m, double* y) {
        const double d = m.consts[0];
        const double a = th(K);
        double acc = th(K + 1);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double t = (x[0] - th(k)) / d;
            const double q = t * t + 1.0;
            acc = acc + a / q;
        }
        y[0] = acc;
    }
    __device__ static void prep_setting(const double* x, const obe_model& m, double* xs) {
        xs[0] = x[0] / m.consts[0];
    }
    __device__ static void pack(const ParamRef& th, const double* thbar, const obe_model& m, double sw,
                                double* pk) {
        const double d = m.consts[0];
#pragma unroll
        for (int k = 0; k < K; ++k) pk[k] = th(k) / d;
        pk[K] = sw * th(K);
        pk[K + 1] = sw * (th(K + 1) - thbar[K + 1]);
    }
    // K >= 3: the K peaks of one evaluation are combined into ONE fraction (sum_of_reciprocals,
    // 2K operations) and the SPT fractions of a lane share one reciprocal: a * num/den with the
    // amplitude folded into the pair inverses.  7 peaks: 35 instead of 39.6 FP64 issue slots per
    // evaluation.  The denominators are products of K factors q >= 1, so the inversion tree of 8
    // settings spans q^(8K): it is range-checked (batch_div_ge1, root < 1e250, i.e. |x - x0|/d up
    // to ~170 at K = 7) and a sweep that leaves the range is repeated with sweep_eval_safe() — the
    // peak-by-peak form, good to |x - x0|/d ~ 1e19.
    static constexpr bool kCombinePeaks = K >= 3;
    // Every K has a fast form that poisons a batch whose inversion tree leaves the double range and a SAFE twin
    // that is IEEE for any finite denominator (round 5: K = 1, 2 as well — their fast pair form used to return a
    // NaN variance beyond |x - x0|/d ~ 4e9 with nothing to repeat the sweep with):
    //   fast, K < 3 : two particles x SPT settings of one peak share one reciprocal (q^16).  No range check is
    //                 spent on it: every q >= 1, so the only failure is OVERFLOW of a product, an overflowed root
    //                 is +inf, fast_rcp(+inf) = NaN (0 * inf in its correction step) and the NaN reaches all 16
    //                 results, the setting's variance and kappa — the host repeats the sweep SAFE.  A finite root
    //                 is inverted to >= 50 bits even where 1/root is subnormal.
    //   fast, K >= 3: the combined fraction, range-checked (batch_div_ge1).
    //   SAFE, all K : the pair form per peak with a branch: a batch whose root stays below 1e250 shares its
    //                 reciprocal as before, any other is inverted element by element (guarded_rcp) — the
    //                 reference's numbers for |x - x0|/d up to sqrt(DBL_MAX) ~ 1e154, where t*t itself overflows
    //                 and NumPy, too, gets a / inf = 0.
    static constexpr bool kHasSafeEval = true;
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval(const double (&xs)[SPT][NXS], const double* pk, double sw,
                                                      const obe_model& m, double (&v)[SPT][NC]) {
        if constexpr (!kCombinePeaks) {
            sweep_eval_one<SPT, false>(xs, pk, v);
        } else {
            double num[SPT], den[SPT];
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                double q[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double t = xs[j][0] - pk[k];
                    q[k] = fma(t, t, 1.0);
                }
                sum_of_reciprocals<0, K>(q, num[j], den[j]);
            }
            double ip[(SPT + 1) / 2];
            batch_div_ge1<SPT>(den, pk[K], ip);                       // a / (den[2h] den[2h+1])
#pragma unroll
            for (int j = 0; j < SPT; ++j) v[j][0] = fma(ip[j / 2] * sibling_of<SPT>(den, j), num[j], pk[K + 1]);
        }
    }
    // the peak-by-peak form: one batched inversion per peak (SAFE: range-guarded, see above)
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval_safe(const double (&xs)[SPT][NXS], const double* pk, double,
                                                           const obe_model&, double (&v)[SPT][NC]) {
        sweep_eval_one<SPT, true>(xs, pk, v);
    }
    template <int SPT, bool GUARDED>
    __device__ __forceinline__ static void sweep_eval_one(const double (&xs)[SPT][NXS], const double* pk,
                                                          double (&v)[SPT][NC]) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) v[j][0] = pk[K + 1];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double q[SPT];
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const double t = xs[j][0] - pk[k];
                q[j] = fma(t, t, 1.0);
            }
            if constexpr (SPT == 1) {
                // (nothing shared: the IEEE-edged reciprocal in BOTH forms — a lane that owns one setting cannot
                // leave the fast form's range, which is what lets small sweeps go unchecked: models.py,
                // safe_sweep_min_spt)
                v[0][0] = fma(pk[K], guarded_rcp(q[0]), v[0][0]);
            } else {
                // Batch inversion stopped one level early: invert the SPT/2 pair products, fold
                // the amplitude into each pair inverse, and let the final multiply by the sibling
                // be the FMA that accumulates the peak:  a/q0 = (a / (q0 q1)) * q1.
                double pp[SPT / 2], ip[SPT / 2];
                double root = 1.0;
#pragma unroll
                for (int h = 0; h < SPT / 2; ++h) {
                    pp[h] = q[2 * h] * q[2 * h + 1];
                    if constexpr (GUARDED) root *= pp[h];
                }
                if (!GUARDED || root < 1e250) {
                    batch_rcp<SPT / 2>(pp, ip);
#pragma unroll
                    for (int h = 0; h < SPT / 2; ++h) {
                        const double g = pk[K] * ip[h];
                        v[2 * h][0] = fma(g, q[2 * h + 1], v[2 * h][0]);
                        v[2 * h + 1][0] = fma(g, q[2 * h], v[2 * h + 1][0]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < SPT; ++j) v[j][0] = fma(pk[K], guarded_rcp(q[j]), v[j][0]);
                }
            }
        }
    }
    // Two particles at once: their 2*SPT/2 pair products share ONE reciprocal, and each
    // particle's amplitude enters the inversion tree at its root (2 multiplies per 16 evaluations).
    // The fast form of the peak-by-peak models (a combined fraction already spans q^(8K) per
    // particle) and, range-guarded, the form every SAFE repeat streams its particles through.
    static constexpr bool kHasPairEval = !kCombinePeaks;
    static constexpr bool kSafePairEval = true;
    template <int SPT, bool GUARDED = false>
    __device__ __forceinline__ static void sweep_eval_pair(const double (&xs)[SPT][NXS], const double* pa,
                                                           const double* pb, double (&va)[SPT][NC],
                                                           double (&vb)[SPT][NC]) {
        static_assert(SPT >= 2 && SPT <= 8, "pair evaluation batches SPT pair products");
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            va[j][0] = pa[K + 1];
            vb[j][0] = pb[K + 1];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double qa[SPT], qb[SPT], pp[SPT], ip[SPT];
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const double ta = xs[j][0] - pa[k], tb = xs[j][0] - pb[k];
                qa[j] = fma(ta, ta, 1.0);
                qb[j] = fma(tb, tb, 1.0);
            }
            double root = 1.0;
#pragma unroll
            for (int h = 0; h < SPT / 2; ++h) {
                pp[h] = qa[2 * h] * qa[2 * h + 1];
                pp[SPT / 2 + h] = qb[2 * h] * qb[2 * h + 1];
                if constexpr (GUARDED) root *= pp[h] * pp[SPT / 2 + h];
            }
            if (!GUARDED || root < 1e250) {
                batch_rcp_scaled2<SPT>(pp, pa[K], pb[K], ip);      // ip = amplitude / pair product
#pragma unroll
                for (int h = 0; h < SPT / 2; ++h) {
                    const double ga = ip[h], gb = ip[SPT / 2 + h];
                    va[2 * h][0] = fma(ga, qa[2 * h + 1], va[2 * h][0]);
                    va[2 * h + 1][0] = fma(ga, qa[2 * h], va[2 * h + 1][0]);
                    vb[2 * h][0] = fma(gb, qb[2 * h + 1], vb[2 * h][0]);
                    vb[2 * h + 1][0] = fma(gb, qb[2 * h], vb[2 * h + 1][0]);
                }
            } else {        // (SAFE only) this lane's 16 denominators do not fit one product: one by one
#pragma unroll
                for (int j = 0; j < SPT; ++j) {
                    va[j][0] = fma(pa[K], guarded_rcp(qa[j]), va[j][0]);
                    vb[j][0] = fma(pb[K], guarded_rcp(qb[j]), vb[j][0]);
                }
            }
        }
    }
};

// y = p0 + p1 * x        tests/test_optbayesexpt.py:11-14
struct LineAB {
    static constexpr int NS = 1, NC = 1, NREAD = 2, NXS = 1, NPK = 2, NCONST = 0;
    __device__ static void eval(const double* x, const ParamRef& th, const obe_model&, double* y) {
        const double bx = th(1) * x[0];
        y[0] = th(0) + bx;
    }
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) { xs[0] = x[0]; }
    __device__ static void pack(const ParamRef& th, const double* thbar, const obe_model&, double sw, double* pk) {
        pk[0] = sw * (th(0) - thbar[0]);
        pk[1] = sw * th(1);
    }
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval(const double (&xs)[SPT][NXS], const double* pk, double,
                                                      const obe_model&, double (&v)[SPT][NC]) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) v[j][0] = fma(pk[1], xs[j][0], pk[0]);
    }
};

// y = p0 * x + p1        demos/line_plus_noise/line_plus_noise.py:36-53
struct LineMB {
    static constexpr int NS = 1, NC = 1, NREAD = 2, NXS = 1, NPK = 2, NCONST = 0;
    __device__ static void eval(const double* x, const ParamRef& th, const obe_model&, double* y) {
        const double mx = th(0) * x[0];
        y[0] = mx + th(1);
    }
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) { xs[0] = x[0]; }
    __device__ static void pack(const ParamRef& th, const double* thbar, const obe_model&, double sw, double* pk) {
        pk[0] = sw * th(0);
        pk[1] = sw * (th(1) - thbar[1]);
    }
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval(const double (&xs)[SPT][NXS], const double* pk, double,
                                                      const obe_model&, double (&v)[SPT][NC]) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) v[j][0] = fma(pk[0], xs[j][0], pk[1]);
    }
};

// y = p0                 tests/test_zinference.py:21-26
struct FirstParam {
    static constexpr int NS = 1, NC = 1, NREAD = 1, NXS = 1, NPK = 1, NCONST = 0;
    __device__ static void eval(const double*, const ParamRef& th, const obe_model&, double* y) { y[0] = th(0); }
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) { xs[0] = x[0]; }
    __device__ static void pack(const ParamRef& th, const double* thbar, const obe_model&, double sw, double* pk) {
        pk[0] = sw * (th(0) - thbar[0]);
    }
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval(const double (&)[SPT][NXS], const double* pk, double,
                                                      const obe_model&, double (&v)[SPT][NC]) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) v[j][0] = pk[0];
    }
};

// Rabi oscillation counts, demos/pipulse/pipulse.py:18-49
// settings (pulsetime, delta_f); params (B1, f_center); consts (baseline, contrast, T1)
struct Rabi {
    static constexpr int NS = 2, NC = 1, NREAD = 2, NXS = 3, NPK = 3, NCONST = 3;
    static constexpr int kSweepCost = 7;   // exp, cos, hypot per evaluation (the generator's estimate for this formula)
    // the fraction removed from the baseline: y = baseline * (1 - frac)
    __device__ __forceinline__ static double frac(double tau, double df, double b1, double fc,
                                                   double contrast, double t1) {
        const double det = df - fc;
        const double rz = det / b1;
        const double zz = rz * rz;
        const double f = hypot(det, b1);
        const double e = exp(-tau / t1);
        const double arg = (6.283185307179586 * f) * tau;
        const double osc = 1.0 - cos(arg);
        const double amp = ((e * contrast) / 2.0) * osc;
        return amp / (zz + 1.0);
    }
    __device__ static void eval(const double* x, const ParamRef& th, const obe_model& m, double* y) {
        y[0] = m.consts[0] * (1.0 - frac(x[0], x[1], th(0), th(1), m.consts[1], m.consts[2]));
    }
    // Sweep form.  With f^2 = det^2 + B1^2:  1/(zz + 1) = B1^2 / f^2  and  1 - cos(2 pi f tau) =
    // 2 sin^2(pi f tau), so
    //   y - baseline = -baseline * [contrast exp(-tau/T1)] * [B1^2] * sin^2(pi f tau) / f^2
    // with the first bracket per setting and the second per particle.  One v_rsq_f64 + two
    // Newton steps give both f and 1/f^2 (no division, no hypot), and sin^2(pi x) needs no
    // quadrant logic: r = x - rint(x) is exact and sin^2(pi r) is even in r.  ~35 issue slots per
    // evaluation instead of ~150 for the literal formula (exp, hypot, cos, three IEEE divisions).
    __device__ static void prep_setting(const double* x, const obe_model& m, double* xs) {
        xs[0] = x[0];
        xs[1] = x[1];
        xs[2] = exp(-x[0] / m.consts[2]) * m.consts[1];       // contrast * exp(-tau / T1)
    }
    __device__ static void pack(const ParamRef& th, const double*, const obe_model& m, double sw, double* pk) {
        pk[0] = th(0) * th(0);                                  // B1^2
        pk[1] = th(1);                                          // f_center
        pk[2] = -(sw * m.consts[0]) * pk[0];                    // -sqrt(w) * baseline * B1^2
    }
    // sin(pi r) / r on |r| <= 1/2 as a polynomial in r^2: coefficients (-1)^k pi^(2k+1) / (2k+1)!,
    // k = 0..10; the first neglected term is 1e-17 at the edge.
    __device__ __forceinline__ static double sinpi_over_r(double r2) {
        double p = 0x1.2877020d52cf0p-31;
        p = fma(p, r2, -0x1.8a404211f9547p-26);
        p = fma(p, r2, 0x1.aaec32af93359p-21);
        p = fma(p, r2, -0x1.6fadb9f155744p-16);
        p = fma(p, r2, 0x1.e8f434d018d63p-12);
        p = fma(p, r2, -0x1.e3074fde8871fp-8);
        p = fma(p, r2, 0x1.50783487ee782p-4);
        p = fma(p, r2, -0x1.32d2cce62bd86p-1);
        p = fma(p, r2, 0x1.466bc6775aae2p+1);
        p = fma(p, r2, -0x1.4abbce625be53p+2);
        return fma(p, r2, 0x1.921fb54442d18p+1);
    }
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval(const double (&xs)[SPT][NXS], const double* pk, double,
                                                      const obe_model&, double (&v)[SPT][NC]) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const double det = xs[j][1] - pk[1];
            const double f2 = fma(det, det, pk[0]);
            double y = __builtin_amdgcn_rsq(f2);                // 1/f to ~2^-23, then two Newton steps
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const double e = fma(-(f2 * y), y, 1.0);
                y = fma(y * 0.5, e, y);
            }
            const double x = (f2 * y) * xs[j][0];               // f * tau
            const double r = x - rint(x);
            const double s = r * sinpi_over_r(r * r);           // +- sin(pi f tau)
            v[j][0] = ((xs[j][2] * pk[2]) * (s * s)) * (y * y);
        }
    }
};

// Parallel RLC coil impedance, demos/lockin/lockin_of_coil.py:63-102
// setting w; params (L, R, C [, noise]); channels (Re Z, Im Z)
struct Coil {
    static constexpr int NS = 1, NC = 2, NREAD = 3, NXS = 2, NPK = 7, NCONST = 0;
    static constexpr int kSweepCost = 3;   // two channels, one shared complex reciprocal
    // (1 + 0j) / (c + dj) the way NumPy's complex divide loop does it (Smith's method,
    // numpy/_core/src/umath/loops.c.src, complex _divide)
    __device__ __forceinline__ static void crecip(double c, double d, double& re, double& im) {
        if (fabs(c) >= fabs(d)) {
            const double rat = d / c;
            const double scl = 1.0 / (c + d * rat);
            re = (1.0 + 0.0 * rat) * scl;
            im = (0.0 - 1.0 * rat) * scl;
        } else {
            const double rat = c / d;
            const double scl = 1.0 / (d + c * rat);
            re = (1.0 * rat + 0.0) * scl;
            im = (0.0 * rat - 1.0) * scl;
        }
    }
    __device__ __forceinline__ static void formula(double w, double L, double R, double C, double* y) {
        double y1r, y1i;
        crecip(R, w * L, y1r, y1i);
        const double yr = y1r + 0.0;
        const double yi = y1i + w * C;
        crecip(yr, yi, y[0], y[1]);
    }
    __device__ static void eval(const double* x, const ParamRef& th, const obe_model&, double* y) {
        formula(x[0], th(0), th(1), th(2), y);
    }
    // Sweep form, one reciprocal per evaluation instead of two complex divisions:
    //   Y = 1/(R + jwL) + jwC = (R - jwL)/n + jwC,  n = R^2 + w^2 L^2
    //   Z = conj(Y)/|Y|^2 = n (R - jE) / (R^2 + E^2),  E = w (C n - L)
    // The reciprocals of D = R^2 + E^2 are batched over the lane's settings; D has no fixed
    // scale (units!), so the poisoned batch + exact twin of the generated models is used.
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) {
        xs[0] = x[0];
        xs[1] = x[0] * x[0];
    }
    __device__ static void pack(const ParamRef& th, const double*, const obe_model&, double sw, double* pk) {
        const double L = th(0), R = th(1), C = th(2);
        pk[0] = L * L;
        pk[1] = R * R;
        pk[2] = C;
        pk[3] = L;
        pk[4] = sw * R;
        pk[5] = -sw;
        pk[6] = R;
    }
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval(const double (&xs)[SPT][NXS], const double* pk, double,
                                                      const obe_model&, double (&v)[SPT][NC]) {
        double n[SPT], e[SPT], d[SPT], rd[SPT];
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            n[j] = fma(xs[j][1], pk[0], pk[1]);
            e[j] = xs[j][0] * fma(pk[2], n[j], -pk[3]);
            d[j] = fma(e[j], e[j], pk[1]);
        }
        batch_rcp_poisoned<SPT>(d, rd);
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const double t = n[j] * rd[j];
            v[j][0] = pk[4] * t;
            v[j][1] = (pk[5] * e[j]) * t;
        }
    }
    // the exact NumPy operation sequence: the repeat after a poisoned batch (OBE_SWEEP_SAFE)
    static constexpr bool kHasSafeEval = true;
    template <int SPT>
    __device__ __forceinline__ static void sweep_eval_safe(const double (&xs)[SPT][NXS], const double* pk, double sw,
                                                           const obe_model&, double (&v)[SPT][NC]) {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            double y[2];
            formula(xs[j][0], pk[3], pk[6], pk[2], y);
            v[j][0] = sw * y[0];
            v[j][1] = sw * y[1];
        }
    }
};

// relative cost of one sweep evaluation (1 unless the model says otherwise): the sweep plans its grid by time
template <class M, class = void>
struct sweep_cost { static constexpr int value = 1; };
template <class M>
struct sweep_cost<M, std::void_t<decltype(M::kSweepCost)>> { static constexpr int value = M::kSweepCost; };

template <class M, class = void>
struct has_pair_eval { static constexpr bool value = false; };
template <class M>
struct has_pair_eval<M, std::enable_if_t<M::kHasPairEval>> { static constexpr bool value = true; };
// models whose SAFE sweep may run its particles two at a time through sweep_eval_pair()
template <class M, class = void>
struct safe_pair_eval { static constexpr bool value = false; };
template <class M>
struct safe_pair_eval<M, std::enable_if_t<M::kSafePairEval>> { static constexpr bool value = true; };
// models whose sweep_eval() may poison out-of-range batches bring a branchy, always-IEEE twin
template <class M, class = void>
struct has_safe_eval { static constexpr bool value = false; };
template <class M>
struct has_safe_eval<M, std::enable_if_t<M::kHasSafeEval>> { static constexpr bool value = true; };

#ifdef OBE_PLUGIN_MODEL_HEADER
// ---- plugin build -------------------------------------------------------------------
// The same kernel sources compiled once more for ONE model generated from a user's
// expression (optbayesexpt_amd/models.py: from_expression): the header defines
// obe::PluginModel with the interface of the structs above, and every entry point of the
// resulting shared library serves that model whatever m.id says.
}  // namespace obe
#include OBE_PLUGIN_MODEL_HEADER
namespace obe {
template <class F>
int dispatch_model(const obe_model&, F&& f) {
    return f(PluginModel{});
}
#else
// Host-side dispatch: f(ModelType{}) for the model named by m.id / m.aux.
template <class F>
int dispatch_model(const obe_model& m, F&& f) {
    switch (m.id) {
        case OBE_MODEL_LORENTZ:
            switch (m.aux) {
                case 1: return f(Lorentz<1>{});
                case 2: return f(Lorentz<2>{});
                case 3: return f(Lorentz<3>{});
                case 4: return f(Lorentz<4>{});
                case 5: return f(Lorentz<5>{});
                case 6: return f(Lorentz<6>{});
                case 7: return f(Lorentz<7>{});
                case 8: return f(Lorentz<8>{});
                default: return bad_arg("Lorentz: aux (number of peaks) must be 1..8");
            }
        case OBE_MODEL_LINE_AB: return f(LineAB{});
        case OBE_MODEL_LINE_MB: return f(LineMB{});
        case OBE_MODEL_FIRST_PARAM: return f(FirstParam{});
        case OBE_MODEL_RABI: return f(Rabi{});
        case OBE_MODEL_COIL: return f(Coil{});
        default: return bad_arg("unknown model id");
    }
}
#endif

}  // namespace obe
