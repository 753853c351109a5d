import os, sys, time, warnings
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import optbayesexpt_amd as obe
g = np.random.default_rng(0)
for n in (5000, 20000, 50000):
    prior = np.array([g.uniform(2, 4, n), g.uniform(400, 2000, n), g.normal(500, 1000, n), g.exponential(500, n)])
    x = np.linspace(1.5, 4.5, 100)
    o = obe.OptBayesExptSweeper(obe.models.lorentzian(), (x,), prior, (0.1,), 3, scale=False)
    o.tuning_parameters["auto_resample"] = False
    sim = np.random.default_rng(2)
    xs = x[10:90]
    ys = 300.0 + 1200.0 / (((xs - 3.1) / 0.1) ** 2 + 1) + 800.0 * sim.standard_normal(len(xs))
    ts = []
    for rep in range(20):
        o.particle_weights = np.full(n, 1.0 / n)
        o._weights.tensor()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o.pdf_update(((xs,), ys))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"n={n}: sweep of {len(xs)} points, no resample: {1e6*np.median(ts)/len(xs):.1f} us/point")
