"""Finer breakdown of the demo-size opt_setting() (developer aid): host time of each step with
no device synchronisation except where the path itself has one."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import optbayesexpt_amd as obe
from optbayesexpt_amd import _lib
settings, prior, cons, true, sigma = bench.make_workload("c1")
o = obe.OptBayesExpt(obe.models.lorentzian(), settings, prior.copy(), cons, scale=False, default_noise_std=sigma)
o.rng = np.random.default_rng(1)
for _ in range(20):
    x = o.opt_setting(); o.pdf_update((x, 49000.0, sigma))
N = 300
acc = {}
def lap(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
for _ in range(N):
    o.pdf_update((x, 49000.0, sigma)); torch.cuda.synchronize()
    t0 = time.perf_counter(); idx = o._draw_indices(o.N_DRAWS, defer_validation=True); lap("draw_indices (rng + 1 launch)", t0)
    t0 = time.perf_counter(); mom = o._moments_on_device(); lap("moments_on_device (2 launches)", t0)
    t0 = time.perf_counter(); p, w = o._pw_tensors(); nz = o._noise_var_device(); c = o._cost_device(); lap("tensors/noise/cost", t0)
    t0 = time.perf_counter(); torch.cuda.synchronize(); lap("sync (drain)", t0)
    t0 = time.perf_counter(); o._check_pending_total(); lap("check total", t0)
    t0 = time.perf_counter(); o.opt_setting(); lap("whole opt_setting (second draw)", t0)
for k, v in acc.items():
    print(f"{k:40s} {1e6 * v / N:8.1f} us")
t1, t2 = [], []
for _ in range(N):
    o.pdf_update((x, 49000.0, sigma)); torch.cuda.synchronize()
    t0 = time.perf_counter(); o.opt_setting(); t1.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); o.opt_setting(); t2.append(time.perf_counter() - t0)
print(f"opt_setting after an update   {1e6*np.median(t1):8.1f} us")
print(f"opt_setting again (fresh CDF) {1e6*np.median(t2):8.1f} us")
