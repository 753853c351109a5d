"""Long runs (developer aid): many cycles in both sweep modes, watching for NaNs, the adaptive
shift's decisions and convergence."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import optbayesexpt_amd as obe
g = np.random.default_rng(0)
for label, ns, n, method, cycles in (("draws 201 x 50000", 201, 50000, "variance_approx", 3000),
                                     ("full 4096 x 262144", 4096, 262144, "variance_full", 600)):
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    o = obe.OptBayesExpt(obe.models.lorentzian(), (np.linspace(1.5, 4.5, ns),), prior, (0.1,), scale=False,
                         utility_method=method, default_noise_std=500.0)
    o.rng = np.random.default_rng(1); sim = np.random.default_rng(2)
    true = (3.05, -1000.0, 50000.0)
    shifted = resamples = 0
    t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for c in range(cycles):
            x = o.opt_setting()
            y = float(o.model_function(x, true, (0.1,))) + 500.0 * sim.standard_normal()
            o.pdf_update((x, y, 500.0))
            shifted += bool(o.last_sweep["shifted"]); resamples += bool(o.just_resampled)
            if c in (9, 99, cycles - 1):
                m, s = o.mean(), o.std()
                assert np.all(np.isfinite(m)) and np.all(np.isfinite(s))
                print(f"  {label} cycle {c + 1:5d}: x0 = {m[0]:.5f} +/- {s[0]:.5f}, a = {m[1]:.1f} +/- {s[1]:.1f}, kappa {o.last_sweep['kappa']:.3g}")
    dt = time.perf_counter() - t0
    print(f"{label}: {cycles} cycles in {dt:.2f} s ({1e3 * dt / cycles:.3f} ms/cycle), {resamples} resamples, shifted sweeps {shifted}")
    assert abs(o.mean()[0] - true[0]) < 6 * o.std()[0]
