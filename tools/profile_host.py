"""Host-side (Python) cost of a measurement cycle: cProfile over real cycles (developer aid).
    python tools/profile_host.py [c1|c2|c3] [cycles]"""
import cProfile, os, pstats, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 200
settings, prior, cons, true, sigma = bench.make_workload(cfg)
obe = bench.build_obe(cfg, None, settings, prior.copy(), cons)
obe.rng = np.random.default_rng(1234)
sim = np.random.default_rng(4321)
fn = obe.model_function


def cycle():
    x = obe.opt_setting()
    y = float(np.atleast_1d(fn(x, true, cons))[0]) + sigma * sim.standard_normal()
    obe.pdf_update((x, y, sigma) if cfg != "c5" else (x, y))


warnings.simplefilter("ignore")
for _ in range(20):
    cycle()
pr = cProfile.Profile()
pr.enable()
for _ in range(cycles):
    cycle()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
