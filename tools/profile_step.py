"""Wall-clock breakdown of one measurement cycle at a BASELINE config (developer aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
settings, prior, cons, true, sigma = bench.make_workload(cfg)
shard = None
if world > 1:      # rank 0 of `world`, collectives stubbed: what one rank of an N-GPU run computes
    from optbayesexpt_amd.dist import SettingsShard
    class _Solo(SettingsShard):
        def _gather_records(self, record):          # no communication: only this rank's record counts
            import torch
            g = torch.full((self.world_size, 4), float("-inf"), dtype=torch.float64)
            g[self.rank] = record.cpu()
            return g
    shard = _Solo(rank=0, world_size=world)
obe = bench.build_obe(cfg, shard, settings, prior.copy(), cons)
obe.rng = np.random.default_rng(1234)
sim = np.random.default_rng(4321)
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T.setdefault(name, []).append(time.perf_counter() - t0)
for cyc in range(24):
    t0 = time.perf_counter(); x = obe.opt_setting(); tick("opt_setting", t0)
    y = float(obe.model_function(x, true, cons)) + sigma * sim.standard_normal()
    rec = (x, y, sigma) if cfg != "c5" else (x, y)
    t0 = time.perf_counter(); obe.pdf_update(rec); tick("pdf_update+resample" if obe.just_resampled else "pdf_update", t0)
    t0 = time.perf_counter(); obe.mean(); obe.std(); tick("mean+std", t0)
for k, v in T.items():
    print(f"{k:22s} n={len(v):2d}  median {1e3*np.median(v):8.3f} ms   min {1e3*min(v):8.3f}  max {1e3*max(v):8.3f}")
# resample pieces
n, d = obe.n_particles, obe.n_dims
t0 = time.perf_counter(); u = obe.rng.random(n); print(f"rng.random({n})           {1e3*(time.perf_counter()-t0):8.3f} ms")
t0 = time.perf_counter(); z = obe.rng.standard_normal((n, d)); print(f"rng.standard_normal      {1e3*(time.perf_counter()-t0):8.3f} ms")
t0 = time.perf_counter(); zd = torch.from_numpy(z).cuda(); torch.cuda.synchronize(); print(f"H2D normals              {1e3*(time.perf_counter()-t0):8.3f} ms")
t0 = time.perf_counter(); zp = torch.from_numpy(z).pin_memory(); print(f"pin normals              {1e3*(time.perf_counter()-t0):8.3f} ms")
t0 = time.perf_counter(); zd.copy_(zp, non_blocking=True); torch.cuda.synchronize(); print(f"H2D pinned               {1e3*(time.perf_counter()-t0):8.3f} ms")
t0 = time.perf_counter(); obe._weights.mark_device_written(); obe._cdf(); torch.cuda.synchronize(); print(f"cdf                      {1e3*(time.perf_counter()-t0):8.3f} ms")
t0 = time.perf_counter(); obe._particles.mark_device_written(); obe.covariance(); print(f"moments+cov              {1e3*(time.perf_counter()-t0):8.3f} ms")
