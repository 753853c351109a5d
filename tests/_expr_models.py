"""Expression models used by the GPU tests (and pre-built by __graft_entry__.build() so that
the plugin libraries travel with the repository snapshot)."""


def expression_models():
    from optbayesexpt_amd import models
    den = "(R**2 + (w*L)**2)"
    g = f"(R / {den})"
    b = f"(w*C - w*L / {den})"
    # the device limits at once: 4 setting dimensions, 16 parameter rows (12 named + 4 noise
    # parameters), 4 output channels
    pn = tuple(f"p{i}" for i in range(12))
    big = models.from_expression(
        ("p0 + p1*s0 + p2*s1*s1 + p3*cos(s2 + p4)",
         "p5*exp(-s3*abs(p6)) + p7*s0*s1",
         "p8/(1 + (s2 - p9)**2) + p10",
         "p11*s3 + p0*p1 - p2"),
        settings=("s0", "s1", "s2", "s3"), parameters=pn, name="limits_4x16x4")
    import _fn_models
    # beyond the widths the fused cloud kernels are compiled for (OBE_FAST_DIMS = 16) and the old channel limit: 20
    # parameters, 5 channels — channel c is the cubic p[4c] + p[4c+1] x + p[4c+2] x^2 + p[4c+3] x^3
    wide = models.from_expression(
        tuple(f"p{4 * c} + p{4 * c + 1}*x + p{4 * c + 2}*x*x + p{4 * c + 3}*x*x*x" for c in range(5)),
        settings=("x",), parameters=tuple(f"p{i}" for i in range(20)), name="wide_20x5")
    # the round-6 limits at once: 8 setting dimensions, 8 channels, 24 named parameters (+ 8 noise rows in the test = 32)
    limits8 = models.from_expression(
        tuple(f"p{3 * c} + p{3 * c + 1}*s{c} + p{3 * c + 2}*s{(c + 1) % 8}*s{(c + 3) % 8}" for c in range(8)),
        settings=tuple(f"s{i}" for i in range(8)), parameters=tuple(f"p{i}" for i in range(24)), name="limits_8x32x8")
    return {
        "limits": big,
        "limits8": limits8,
        "wide": wide,
        # translated from the source of plain reference-style functions (models.from_function)
        "fn_lorentzian": models.from_function(_fn_models.lorentzian),
        "fn_rabi": models.from_function(_fn_models.rabi),
        # sin / cos / sqrt / hypot at the inner level of the sweep (the fast elementary functions)
        "trig": models.from_expression("a*sin(w*t + p) + b*cos(w*t)*sqrt(t + c) + hypot(a*t, b)", settings=("t",),
                                       parameters=("w", "p", "a", "b"), constants=("c",)),
        # 11 parameters: five Lorentzian peaks with their own amplitudes on one background (the fuzz recipes'
        # many-parameter expression model: tools/fuzz_parity.py)
        "peaks11": models.from_expression(
            "b + " + " + ".join(f"a{k} / (((x - c{k}) / d)**2 + 1)" for k in range(5)), settings=("x",),
            parameters=tuple(f"c{k}" for k in range(5)) + tuple(f"a{k}" for k in range(5)) + ("b",), constants=("d",)),
        # a model with a true pole (tests the NaN semantics of the sweep)
        "pole": models.from_expression("a / (x - x0)", settings=("x",), parameters=("x0", "a")),
        # demos/find_peak/sequentialLorentzian.py:53-75 as a formula
        "lorentzian": models.from_expression("b + a / (((x - x0) / d)**2 + 1)", settings=("x",),
                                             parameters=("x0", "a", "b"), constants=("d",)),
        # demos/pipulse/pipulse.py:18-49, same operation order as the NumPy function
        "rabi": models.from_expression(
            "baseline*(1 - exp(-t/T1)*contrast/2*(1 - cos(pi*2*hypot(df - fc, B1)*t))/(((df - fc)/B1)**2 + 1))",
            settings=("t", "df"), parameters=("B1", "fc"), constants=("baseline", "contrast", "T1")),
        # demos/lockin/lockin_of_coil.py:63-102 in real arithmetic: Z = (G - jB)/(G^2 + B^2)
        "coil": models.from_expression((f"{g} / ({g}**2 + {b}**2)", f"-{b} / ({g}**2 + {b}**2)"),
                                       settings=("w",), parameters=("L", "R", "C")),
    }
