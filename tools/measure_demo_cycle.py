"""The reference's find_peak demo loop at its own size (demos/find_peak/sequentialLorentzian.py:
200 settings, 50 000 particles, N_DRAWS = 30, good_setting(pickiness=19) + pdf_update + std() per
cycle) against this package, and the oracle class on one host core beside it (developer aid).
BASELINE.md: the reference itself takes 3.3 ms per cycle of this loop on one core."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import optbayesexpt_amd as obe
import oracle
from oracle import models as omodels

g = np.random.default_rng(0)
n, ns = 50000, 200
prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
sv = (np.linspace(1.5, 4.5, ns),)
true, cons, sigma = (3.0, -1000.0, 50000.0), (0.1,), 500.0
warnings.simplefilter("ignore")
for label, make in (("optbayesexpt_amd (MI355X)", lambda: obe.OptBayesExpt(obe.models.lorentzian(), sv, prior.copy(), cons, scale=False)),
                    ("oracle class (NumPy, 1 core)", lambda: oracle.OracleOptBayesExpt(omodels.lorentzian, sv, prior.copy(), cons, scale=False))):
    o = make()
    o.rng = np.random.default_rng(1)
    sim = np.random.default_rng(2)
    cycles = 3000 if "amd" in label else 300
    tt, res = [], 0
    for c in range(cycles + 50):
        t0 = time.perf_counter()
        x = o.good_setting(pickiness=19)
        y = float(omodels.lorentzian(x, true, cons)) + sigma * sim.standard_normal()
        o.pdf_update((x, y, sigma))
        s = o.std()
        if c >= 50:
            tt.append(time.perf_counter() - t0); res += bool(o.just_resampled)
    tt = np.array(tt)
    print(f"{label:30s} {1e3 * tt.mean():7.3f} ms per cycle (median {1e3 * np.median(tt):.3f}), {res} resamples in {cycles} cycles, "
          f"x0 = {o.mean()[0]:.4f} +/- {o.std()[0]:.4f}")
