"""What ONE rank of an 8-GPU settings-sharded job does per cycle, measured on one GPU.

    python tools/shard_cycle.py [c3|c5] [world=8] [steps]       (c3 sharded 8 ways = BASELINE config c4)

1. the full cycle on this GPU (what `bench.py --gpus 1` times): ms per step, K1 inside the cycles, resamples;
   the chosen settings and the simulated measurements are logged;
2. the same experiment — the same measurements, hence the same posterior, the same resample decisions and
   the same generator stream — through an object that owns only rank 0's 1/world slice of the settings
   (`settings_shard`): its sweep, update, resample, constraint mask and host work are exactly what every
   rank of the sharded job executes; the collective is replaced by the device-to-host read of the 32-byte
   record that follows it in the real path (no RCCL peer on a one-GPU box);
3. prediction: speed-up = full cycle / (rank cycle + collective), with the all-gather latency of a real
   RCCL communicator (tools/profile_collective.py: world of one, i.e. a lower bound) stated next to it.
"""
import ctypes
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                        # noqa: E402
import bench                                        # noqa: E402
from optbayesexpt_amd.dist import SettingsShard     # noqa: E402


class _Solo(SettingsShard):
    """Rank `rank` of `world_size` without peers: the record of this rank's slice is read back from
    device memory as combine_records() does after the all-gather."""

    def _gather_records(self, record):
        g = self.__dict__.get("_g")
        if g is None:                     # (made once, like the real path's landing zone)
            g = self._g = np.full((self.world_size, 4), float("-inf"))
            g[:, 2] = 0.0
        g[self.rank] = record.cpu().numpy()
        return g


def run(cfg, shard, log, steps, warmup, events=True):
    settings, prior, cons, true, sigma = bench.make_workload(cfg)
    obe = bench.build_obe(cfg, shard, settings, prior.copy(), cons)
    obe.rng = np.random.default_rng(1234)
    sim = np.random.default_rng(4321)
    fn = obe.model_function
    noise_rec = bench.CONFIGS[cfg][2] == "lorentzian"
    if not os.environ.get("OBE_NO_CLOCK_WARM_UP"):
        bench.warm_resample_path(cfg, settings, prior, cons)      # (as bench.py does: the resample kernels' code objects)
        bench.warm_clocks(obe)         # (as bench.py does: the chip's clocks ramp for ~35 ms after an idle second)
    times, res = [], []
    record = log is None
    if record:
        log = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for c in range(warmup + steps):
            if c == warmup:
                torch.cuda.synchronize()
                if events:         # (HIP events around every sweep kernel: two barrier packets, ~6 us each, per cycle)
                    obe._mlib.call("obe_sweep_timing", 1, None, None)
            t0 = time.perf_counter()
            x = obe.opt_setting()
            if record:
                y = float(fn(x, true, cons)) + sigma * sim.standard_normal()
                log.append((x, y))
            else:
                x, y = log[c]              # the job's global choice and its measurement
            rec = (x, y, sigma) if noise_rec else (x, y)
            if c in (warmup - 1, warmup + steps - 1):      # no sweep enqueued across the ends of the timed region
                bench.update_at_boundary(obe, rec)
            else:
                obe.pdf_update(rec)
            if c >= warmup:
                times.append(1e3 * (time.perf_counter() - t0))
                res.append(bool(obe.just_resampled))
    torch.cuda.synchronize()
    k1_ms, k1_n = ctypes.c_double(0.0), ctypes.c_int64(0)
    obe._mlib.call("obe_sweep_timing", 0, ctypes.byref(k1_ms), ctypes.byref(k1_n))
    times, res = np.array(times), np.array(res)
    return dict(ms=float(times.mean()), k1=k1_ms.value / max(k1_n.value, 1), resamples=int(res.sum()),
                plain=float(np.median(times[~res])) if (~res).any() else float("nan"),
                resample=float(np.median(times[res])) if res.any() else float("nan"), flags=res,
                n_local=obe._s_end - obe._s_begin), log


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else (20 if cfg == "c3" else 24)
    warmup = 5
    full, log = run(cfg, None, None, steps, warmup)
    rank, _ = run(cfg, _Solo(rank=0, world_size=world), log, steps, warmup)
    # the rank's cycles once more WITHOUT the events around its sweep kernel: what a rank of a real job runs
    # (the events are this tool's instrument for "K1 in cycle"; the same trajectory, so the same cycles)
    bare, _ = run(cfg, _Solo(rank=0, world_size=world), log, steps, warmup, events=False)
    assert (bare["flags"] == rank["flags"]).all()
    assert (full["flags"] == rank["flags"]).all(), "the shard did not follow the full run's resample decisions"
    ns, n_p = bench.CONFIGS[cfg][0], bench.CONFIGS[cfg][1]
    name = "c4 (= c3 sharded)" if cfg == "c3" else cfg
    print(f"{name}: {ns} settings x {n_p} particles, {steps} timed cycles after {warmup}, {full['resamples']} of them resample")
    print(f"  one GPU, all settings   : {full['ms']:8.3f} ms/cycle   K1 in cycle {full['k1']:8.3f} ms   everything else "
          f"{full['ms'] - full['k1']:6.3f} ms   (plain cycle {full['plain']:.3f}, resample cycle {full['resample']:.3f})")
    print(f"  one rank of {world} ({rank['n_local']:5d} settings): {rank['ms']:8.3f} ms/cycle   K1 in cycle {rank['k1']:8.3f} ms   "
          f"everything else {rank['ms'] - rank['k1']:6.3f} ms   (plain cycle {rank['plain']:.3f}, resample cycle {rank['resample']:.3f})")
    print(f"  the same rank without the events around K1: {bare['ms']:8.3f} ms/cycle   (plain cycle {bare['plain']:.3f}, "
          f"resample cycle {bare['resample']:.3f}) — the figure the prediction uses")
    print(f"  K1 of the rank / (K1 of one GPU / {world}) = {rank['k1'] / (full['k1'] / world):.3f}")
    for coll_us in (0.0, 30.0, 60.0):
        t = bare["ms"] + 1e-3 * coll_us
        print(f"  predicted speed-up at {world} GPUs with a {coll_us:4.0f} us all-gather per cycle: {full['ms'] / t:5.2f}x "
              f"({full['ms'] / t / world:.0%} of linear)")
    print("  (the 32-byte all-gather through a real RCCL communicator of one rank: tools/profile_collective.py; "
          "8 ranks over xGMI add the ring latency — 30-60 us brackets it)")
    # machine-readable, merged into the file `bench.py --gpus N` quotes next to its measured value
    # (OBE_PROJECTION_OUT=path; default: print only)
    entry = dict(config=cfg, world=world, steps=steps, warmup=warmup, resamples=int(full["resamples"]),
                 one_gpu_ms_per_cycle=full["ms"], one_gpu_k1_ms=full["k1"], rank_ms_per_cycle=bare["ms"],
                 rank_k1_ms=rank["k1"], rank_plain_ms=bare["plain"], rank_resample_ms=bare["resample"],
                 settings_per_rank=int(rank["n_local"]))
    import json
    print("PROJECTION " + json.dumps(entry))
    out = os.environ.get("OBE_PROJECTION_OUT")
    if out:
        table = json.load(open(out)) if os.path.exists(out) else {}
        table.setdefault(cfg, {})[str(world)] = entry
        with open(out, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
