"""Demo-size cycle (201 settings x 5000 particles, N_DRAWS = 30): where the time goes
(developer aid)."""
import cProfile, pstats, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import optbayesexpt_amd as obe
settings, prior, cons, true, sigma = bench.make_workload("c1")
o = obe.OptBayesExpt(obe.models.lorentzian(), settings, prior.copy(), cons, scale=False, default_noise_std=sigma)
o.rng = np.random.default_rng(1); sim = np.random.default_rng(2)
def cycle():
    x = o.opt_setting()
    o.pdf_update((x, float(o.model_function(x, true, cons)) + sigma * sim.standard_normal(), sigma))
for _ in range(20): cycle()
ts = []
for _ in range(300):
    t0 = time.perf_counter(); cycle(); ts.append(time.perf_counter() - t0)
print("cycle median %.1f us, min %.1f us" % (1e6 * np.median(ts), 1e6 * min(ts)))
t_opt, t_upd = [], []
for _ in range(300):
    t0 = time.perf_counter(); x = o.opt_setting(); t1 = time.perf_counter()
    o.pdf_update((x, 49000.0, sigma)); t2 = time.perf_counter()
    t_opt.append(t1 - t0); t_upd.append(t2 - t1)
print("opt_setting median %.1f us, pdf_update median %.1f us" % (1e6 * np.median(t_opt), 1e6 * np.median(t_upd)))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): cycle()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
