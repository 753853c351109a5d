// Device model registry: the HIP counterpart of the reference's Python
// `model_function(settings, parameters, constants)` (obe_base.py:50-72).
//
// Each model provides two evaluations of the same formula:
//   eval()      — the exact NumPy operation sequence of the corresponding demo
//                 function, one correctly-rounded op at a time (no FMA contraction);
//                 used by the HBM-bound kernels (Bayes update, eval_over_*), where
//                 it costs nothing and gives reference-identical bits.
//   eval_fast() — the flop-bound sweep form: per-setting and per-particle terms are
//                 hoisted (prep_setting / pack), divisions become v_rcp_f64 + Newton,
//                 and products/sums are fused.  Outputs are *shifted* by a
//                 per-particle-independent constant (the mean background), which
//                 leaves the variance unchanged and removes the cancellation.
#pragma once

#include <type_traits>

#include "obe_common.h"

namespace obe {

// 1/q to ~1 ulp: v_rcp_f64 seeds ~2^-23 relative (ISA: "2**29 ULP"), two Newton steps
// square that twice.
__device__ __forceinline__ double fast_rcp(double q) {
    double r = __builtin_amdgcn_rcp(q);
    double e = fma(-q, r, 1.0);
    r = fma(r, e, r);
    e = fma(-q, r, 1.0);
    r = fma(r, e, r);
    return r;
}

// ---------------------------------------------------------------------------
// y = b + sum_{k<K} a / (((x - x0_k)/d)^2 + 1)
// params: x0_0..x0_{K-1}, a, b [, anything]; const d
// demos/find_peak/sequentialLorentzian.py:53-75 (K = 1)
template <int K>
struct Lorentz {
    static constexpr int NS = 1, NC = 1, NREAD = K + 2;
    static constexpr int NXS = 1;        // prepared setting: x/d
    static constexpr int NPK = K + 2;    // packed particle: x0_k/d ..., a, b - bbar

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
    __device__ static void pack(const ParamRef& th, const double* thbar, const obe_model& m, double* pk) {
        const double d = m.consts[0];
#pragma unroll
        for (int k = 0; k < K; ++k) pk[k] = th(k) / d;
        pk[K] = th(K);
        pk[K + 1] = th(K + 1) - thbar[K + 1];
    }
    __device__ __forceinline__ static void eval_fast(const double* xs, const double* pk, double* y) {
        double acc = pk[K + 1];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double t = xs[0] - pk[k];
            const double q = fma(t, t, 1.0);
            acc = fma(pk[K], fast_rcp(q), acc);
        }
        y[0] = acc;
    }
};

// y = p0 + p1 * x        tests/test_optbayesexpt.py:11-14
struct LineAB {
    static constexpr int NS = 1, NC = 1, NREAD = 2, NXS = 1, NPK = 2;
    __device__ static void eval(const double* x, const ParamRef& th, const obe_model&, double* y) {
        const double bx = th(1) * x[0];
        y[0] = th(0) + bx;
    }
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) { xs[0] = x[0]; }
    __device__ static void pack(const ParamRef& th, const double* thbar, const obe_model&, double* pk) {
        pk[0] = th(0) - thbar[0];
        pk[1] = th(1);
    }
    __device__ __forceinline__ static void eval_fast(const double* xs, const double* pk, double* y) {
        y[0] = fma(pk[1], xs[0], pk[0]);
    }
};

// y = p0 * x + p1        demos/line_plus_noise/line_plus_noise.py:36-53
struct LineMB {
    static constexpr int NS = 1, NC = 1, NREAD = 2, NXS = 1, NPK = 2;
    __device__ static void eval(const double* x, const ParamRef& th, const obe_model&, double* y) {
        const double mx = th(0) * x[0];
        y[0] = mx + th(1);
    }
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) { xs[0] = x[0]; }
    __device__ static void pack(const ParamRef& th, const double* thbar, const obe_model&, double* pk) {
        pk[0] = th(0);
        pk[1] = th(1) - thbar[1];
    }
    __device__ __forceinline__ static void eval_fast(const double* xs, const double* pk, double* y) {
        y[0] = fma(pk[0], xs[0], pk[1]);
    }
};

// y = p0                 tests/test_zinference.py:21-26
struct FirstParam {
    static constexpr int NS = 1, NC = 1, NREAD = 1, NXS = 1, NPK = 1;
    __device__ static void eval(const double*, const ParamRef& th, const obe_model&, double* y) { y[0] = th(0); }
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) { xs[0] = x[0]; }
    __device__ static void pack(const ParamRef& th, const double* thbar, const obe_model&, double* pk) {
        pk[0] = th(0) - thbar[0];
    }
    __device__ __forceinline__ static void eval_fast(const double*, const double* pk, double* y) { y[0] = pk[0]; }
};

// Rabi oscillation counts, demos/pipulse/pipulse.py:18-49
// settings (pulsetime, delta_f); params (B1, f_center); consts (baseline, contrast, T1)
struct Rabi {
    static constexpr int NS = 2, NC = 1, NREAD = 2, NXS = 2, NPK = 2;
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
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) {
        xs[0] = x[0];
        xs[1] = x[1];
    }
    __device__ static void pack(const ParamRef& th, const double*, const obe_model&, double* pk) {
        pk[0] = th(0);
        pk[1] = th(1);
    }
    // needs the constants: handled through eval_fast_m below
    static constexpr bool kFastNeedsModel = true;
    __device__ __forceinline__ static void eval_fast_m(const double* xs, const double* pk,
                                                       const obe_model& m, double* y) {
        // y - baseline: same variance, without the 1 - frac cancellation
        y[0] = -(m.consts[0] * frac(xs[0], xs[1], pk[0], pk[1], m.consts[1], m.consts[2]));
    }
};

// Parallel RLC coil impedance, demos/lockin/lockin_of_coil.py:63-102
// setting w; params (L, R, C [, noise]); channels (Re Z, Im Z)
struct Coil {
    static constexpr int NS = 1, NC = 2, NREAD = 3, NXS = 1, NPK = 3;
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
    __device__ static void prep_setting(const double* x, const obe_model&, double* xs) { xs[0] = x[0]; }
    __device__ static void pack(const ParamRef& th, const double*, const obe_model&, double* pk) {
        pk[0] = th(0);
        pk[1] = th(1);
        pk[2] = th(2);
    }
    __device__ __forceinline__ static void eval_fast(const double* xs, const double* pk, double* y) {
        formula(xs[0], pk[0], pk[1], pk[2], y);
    }
};

template <class M, class = void>
struct fast_needs_model { static constexpr bool value = false; };
template <class M>
struct fast_needs_model<M, std::enable_if_t<M::kFastNeedsModel>> { static constexpr bool value = true; };

template <class M>
__device__ __forceinline__ void model_eval_fast(const double* xs, const double* pk, const obe_model& m, double* y) {
    if constexpr (fast_needs_model<M>::value) M::eval_fast_m(xs, pk, m, y);
    else M::eval_fast(xs, pk, y);
}

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

}  // namespace obe
