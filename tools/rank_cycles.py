"""N cycles of ONE rank's slice of a sharded job, nothing else (developer aid; the program rocprofv3 traces for the
timeline of a rank's cycle):   python tools/rank_cycles.py [c5|c3] [world=8] [cycles=14]"""
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
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 14
settings, prior, cons, true, sigma = bench.make_workload(cfg)
obe = bench.build_obe(cfg, _Solo(rank=0, world_size=world) if world > 1 else None, settings, prior.copy(), cons)
obe.rng = np.random.default_rng(1234)
sim = np.random.default_rng(4321)
fn = obe.model_function
noise_rec = bench.CONFIGS[cfg][2] == "lorentzian"
bench.warm_clocks(obe, 60.0)
warnings.simplefilter("ignore")
flags = []
for c in range(cycles):
    x = obe.opt_setting()
    y = float(np.atleast_1d(fn(x, true, cons))[0]) + sigma * sim.standard_normal()
    obe.pdf_update((x, y, sigma) if noise_rec else (x, y))
    flags.append(int(obe.just_resampled))
torch.cuda.synchronize()
print("resampled:", flags)
