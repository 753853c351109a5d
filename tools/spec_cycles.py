"""Per-cycle wall time of a bench config with and without the speculative sweep (developer aid).

    python tools/spec_cycles.py c5 [steps]

prints, for tuning_parameters['speculative_sweep'] = False / 'auto', every cycle's time, whether it resampled
and whether its sweep was the speculative one; then the medians."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import torch  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ns, n_p, model, _ = bench.CONFIGS[cfg]
settings, prior, cons, true, sigma = bench.make_workload(cfg)
for mode in (False, "auto"):
    obe = bench.build_obe(cfg, None, settings, prior.copy(), cons)
    obe.tuning_parameters["speculative_sweep"] = mode
    obe.rng = np.random.default_rng(1234)
    sim = np.random.default_rng(4321)
    fn = obe.model_function
    taken = []
    take = obe._take_speculative_sweep

    def counting(shifted):
        got = take(shifted)
        taken.append(got is not None)
        return got

    obe._take_speculative_sweep = counting
    rows = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for c in range(steps):
            torch.cuda.synchronize() if mode is False else None
            t0 = time.perf_counter()
            x = obe.opt_setting()
            t1 = time.perf_counter()
            y = float(fn(x, true, cons)) + sigma * sim.standard_normal()
            obe.pdf_update((x, y, sigma) if model == "lorentzian" else (x, y))
            t2 = time.perf_counter()
            rows.append((1e3 * (t2 - t0), 1e3 * (t1 - t0), 1e3 * (t2 - t1), bool(obe.just_resampled),
                         bool(taken and taken[-1]), dict(obe.last_sweep)))
    torch.cuda.synchronize()
    print(f"== {cfg} speculative_sweep={mode}")
    for c, r in enumerate(rows):
        print(f"  {c:3d}  cycle {r[0]:8.3f} ms = opt_setting {r[1]:8.3f} + update {r[2]:7.3f}   "
              f"resampled={int(r[3])} speculative={int(r[4])} {r[5]}")
    ms = np.array([r[0] for r in rows[4:]])
    res = np.array([r[3] for r in rows[4:]])
    print(f"  median plain {np.median(ms[~res]) if (~res).any() else float('nan'):.3f} ms, "
          f"resample {np.median(ms[res]) if res.any() else float('nan'):.3f} ms, mean {ms.mean():.3f} ms")
