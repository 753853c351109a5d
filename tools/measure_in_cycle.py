"""In-cycle duration of the K1 sweep kernel (obe_sweep_timing) for different host patterns between
sweeps (developer aid): real cycles, sweeps only, sweeps with an idle gap."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
settings, prior, cons, true, sigma = bench.make_workload(cfg)
obe = bench.build_obe(cfg, None, settings, prior.copy(), cons)
obe.rng = np.random.default_rng(1234)
sim = np.random.default_rng(4321)
fn = obe.model_function


def cycle():
    x = obe.opt_setting()
    y = float(np.atleast_1d(fn(x, true, cons))[0]) + sigma * sim.standard_normal()
    obe.pdf_update((x, y, sigma) if cfg != "c5" else (x, y))


def timed(label, body, n=8):
    tot, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
    torch.cuda.synchronize()
    obe._mlib.call("obe_sweep_timing", 1, None, None)
    t0 = time.perf_counter()
    for _ in range(n):
        body()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    obe._mlib.call("obe_sweep_timing", 0, ctypes.byref(tot), ctypes.byref(cnt))
    print(f"{label:34s} K1 in-cycle {tot.value / max(cnt.value, 1):8.3f} ms ({cnt.value} launches)   wall per trip {1e3 * wall:8.3f} ms")


for _ in range(4):
    cycle()
timed("real cycles", cycle)
timed("opt_setting only", lambda: obe.opt_setting())
for gap in (50, 100, 200, 400):
    def body(gap=gap):
        obe.opt_setting()
        t = time.perf_counter()
        while time.perf_counter() - t < gap * 1e-6:
            pass
    timed(f"opt_setting + {gap} us host spin", body)
timed("real cycles again", cycle)
