"""Where a reference-semantics cycle spends its host time: the library calls (launch + wait) vs the
Python around them (developer aid).   python tools/profile_host_split.py [c1|c2] [cycles]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
settings, prior, cons, true, sigma = bench.make_workload(cfg)
obe = bench.build_obe(cfg, None, settings, prior.copy(), cons)
obe.rng = np.random.default_rng(1234)
sim = np.random.default_rng(4321)
fn = obe.model_function
acc = {}
for lib in {id(obe._lib._lib): obe._lib._lib, id(obe._mlib._lib): obe._mlib._lib}.values():
    orig = lib.call
    def timed(name, *a, _orig=orig):
        t0 = time.perf_counter()
        r = _orig(name, *a)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        acc["#" + name] = acc.get("#" + name, 0) + 1
        return r
    lib.call = timed
warnings.simplefilter("ignore")
t_opt = t_upd = t_user = 0.0
for c in range(cycles + 200):
    if c == 200:
        acc.clear(); t_opt = t_upd = t_user = 0.0; t_all = time.perf_counter()
    t0 = time.perf_counter()
    x = obe.opt_setting()
    t1 = time.perf_counter()
    y = float(fn(x, true, cons)) + sigma * sim.standard_normal()
    t2 = time.perf_counter()
    obe.pdf_update((x, y, sigma))
    t3 = time.perf_counter()
    t_opt += t1 - t0; t_user += t2 - t1; t_upd += t3 - t2
total = time.perf_counter() - t_all
us = lambda s: 1e6 * s / cycles
print(f"{cfg}: cycle {us(total):.1f} us = opt_setting {us(t_opt):.1f} + measurement simulation {us(t_user):.1f} + pdf_update {us(t_upd):.1f}")
in_calls = sum(v for k, v in acc.items() if not k.startswith("#"))
print(f"  inside library calls (launch + device wait): {us(in_calls):.1f} us; Python around them: {us(total - in_calls) - us(t_user):.1f} us")
for k in sorted((k for k in acc if not k.startswith("#")), key=lambda k: -acc[k]):
    print(f"    {k:28s} {acc['#' + k] / cycles:5.2f} calls/cycle  {1e6 * acc[k] / acc['#' + k]:7.1f} us each")
