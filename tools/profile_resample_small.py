"""Wall-clock pieces of one resample() at demo size (5000 particles, host RNG) (developer aid)."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import optbayesexpt_amd as obe
settings, prior, cons, true, sigma = bench.make_workload("c1")
pdf = obe.ParticlePDF(prior.copy(), scale=False)
g = np.random.default_rng(3)
w = g.exponential(1.0, prior.shape[1]); w /= w.sum()
pdf.rng = np.random.default_rng(5)
ts = []
for rep in range(50):
    pdf.particle_weights = w
    pdf._weights.tensor(); torch.cuda.synchronize()
    t0 = time.perf_counter(); pdf.resample(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("resample median us", 1e6 * np.median(ts))
pr = cProfile.Profile(); pr.enable()
for rep in range(100):
    pdf.particle_weights = w
    pdf.resample()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
