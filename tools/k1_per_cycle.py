"""K1 of every cycle of one rank's slice, next to what kind of cycle it was (developer aid, GPU):

    python tools/k1_per_cycle.py [c5|c3] [world=8] [cycles=30]

Reads obe_sweep_timing after every cycle (which waits for the sweep enqueued ahead: the cycle times of this
tool mean nothing, the per-launch kernel times do)."""
import ctypes
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch                    # noqa: E402
import bench                    # noqa: E402
from shard_cycle import _Solo   # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 30
settings, prior, cons, true, sigma = bench.make_workload(cfg)
obe = bench.build_obe(cfg, _Solo(rank=0, world_size=world) if world > 1 else None, settings, prior.copy(), cons)
obe.rng = np.random.default_rng(1234)
sim = np.random.default_rng(4321)
fn = obe.model_function
noise_rec = bench.CONFIGS[cfg][2] == "lorentzian"
tot, cnt = ctypes.c_double(0.0), ctypes.c_int64(0)
obe._mlib.call("obe_sweep_timing", 1, None, None)
prev_t, prev_n = 0.0, 0
warnings.simplefilter("ignore")
for c in range(cycles):
    x = obe.opt_setting()
    sweep = dict(obe.last_sweep)
    y = float(np.atleast_1d(fn(x, true, cons))[0]) + sigma * sim.standard_normal()
    obe.pdf_update((x, y, sigma) if noise_rec else (x, y))
    obe._mlib.call("obe_sweep_timing", -1, ctypes.byref(tot), ctypes.byref(cnt))
    dn = cnt.value - prev_n
    dt = tot.value - prev_t
    prev_t, prev_n = tot.value, cnt.value
    print(f"cycle {c:3d}: {dn} timed launch(es) {dt / max(dn, 1):8.3f} ms each   opt_setting's sweep: shifted={sweep['shifted']} "
          f"safe={sweep['safe']} kappa={sweep['kappa']:.3g}   resampled={bool(obe.just_resampled)}   {obe.sweep_state()['pending']}")
