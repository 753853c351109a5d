"""Sweeper cycle timings: opt_setting() over all (start, stop) pairs and pdf_update() of a
whole sweep, demo size and large sizes (developer aid; numbers quoted in DESIGN.md)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import optbayesexpt_amd as obe

g = np.random.default_rng(0)
cases = [("demo  100 x   50000 approx", 100, 50000, "variance_approx"),
         ("     4096 x  262144 full  ", 4096, 262144, "variance_full"),
         ("    16384 x  524288 full  ", 16384, 524288, "variance_full")]
for label, ns, n, method in cases:
    prior = np.array([g.uniform(2, 4, n), g.uniform(400, 2000, n), g.normal(500, 1000, n), g.exponential(500, n)])
    x = np.linspace(1.5, 4.5, ns)
    t0 = time.perf_counter()
    o = obe.OptBayesExptSweeper(obe.models.lorentzian(), (x,), prior, (0.1,), 3, scale=False, utility_method=method)
    t_ctor = time.perf_counter() - t0
    o.rng = np.random.default_rng(1)
    sim = np.random.default_rng(2)
    t_opt, t_upd, n_pts, n_res = [], [], [], 0
    for cyc in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, e = o.opt_setting()
        torch.cuda.synchronize(); t_opt.append(time.perf_counter() - t0)
        e = min(e, s + 300)
        xs = x[s:e]
        ys = 300.0 + 1200.0 / (((xs - 3.1) / 0.1) ** 2 + 1) + 800.0 * sim.standard_normal(len(xs))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            o.pdf_update(((xs,), ys))
            torch.cuda.synchronize(); t_upd.append(time.perf_counter() - t0)
        n_pts.append(len(xs))
    per_pt = np.array(t_upd) / np.array(n_pts)
    print(f"{label}: pairs {len(o.start_stop_indices):9d}  ctor {t_ctor:6.2f} s  opt_setting median {1e3*np.median(t_opt[1:]):8.3f} ms"
          f"  sweep update {1e6*np.median(per_pt):7.1f} us/point (sweeps of {n_pts})")
