"""cProfile of the host side of a cycle (developer aid): which Python functions the ~35 us around
the library calls are spent in.   python tools/profile_python_overhead.py [c1|c2] [cycles]"""
import cProfile, os, pstats, sys, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
settings, prior, cons, true, sigma = bench.make_workload(cfg)
obe = bench.build_obe(cfg, None, settings, prior.copy(), cons)
obe.rng = np.random.default_rng(1234)
sim = np.random.default_rng(4321)
fn = obe.model_function
warnings.simplefilter("ignore")


def loop(n):
    for _ in range(n):
        x = obe.opt_setting()
        y = float(fn(x, true, cons)) + sigma * sim.standard_normal()
        obe.pdf_update((x, y, sigma))


loop(200)
pr = cProfile.Profile()
pr.enable()
loop(cycles)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
