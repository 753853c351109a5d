"""How much accuracy the unshifted sweep actually loses as a function of kappa = (mean y)^2 / var
(developer aid; calibrates the KAPPA_ENTER / KAPPA_LEAVE thresholds of OptBayesExpt._sweep_device)."""
import os, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import optbayesexpt_amd as obe
g = np.random.default_rng(0)
for ns, n in ((4096, 262144), (8192, 1048576)):
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    o = obe.OptBayesExpt(obe.models.lorentzian(), (np.linspace(1.5, 4.5, ns),), prior, (0.1,), scale=False,
                         utility_method="variance_full", default_noise_std=500.0)
    o.rng = np.random.default_rng(1); sim = np.random.default_rng(2)
    true = (3.05, -1000.0, 50000.0)
    rows = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for c in range(400):
            o.tuning_parameters["sweep_shift"] = "auto"
            x = o.opt_setting()
            o.pdf_update((x, float(o.model_function(x, true, (0.1,))) + 500.0 * sim.standard_normal(), 500.0))
            if c % 20 == 19:
                o.tuning_parameters["sweep_shift"] = "always"
                ref = o.yvar_from_parameter_draws()[0]
                kap = o.last_sweep["kappa"]
                o.tuning_parameters["sweep_shift"] = "never"
                got = o.yvar_from_parameter_draws()[0]
                rel = np.abs(got - ref) / np.maximum(ref, 1e-300)
                # error relative to the largest variance too (what moves the utility ranking)
                rows.append((c + 1, kap, rel.max(), np.abs(got - ref).max() / ref.max(), int(np.argmax(got) == np.argmax(ref))))
    print(f"{ns} settings x {n} particles: cycle, kappa, max rel. error of the unshifted variance, max error / max variance, same argmax")
    for r in rows:
        print("  %4d  kappa %10.3g   %9.2e   %9.2e   %d" % r)
