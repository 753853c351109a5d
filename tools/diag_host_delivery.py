"""Two identical sweeper objects through the same point-by-point updates with frequent resamples; every
resample's intermediate results are logged and compared (diagnostic for an intermittent mismatch)."""
import sys, os, warnings, zlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
warnings.simplefilter("ignore")
import numpy as np, torch
import optbayesexpt_amd as obe
from optbayesexpt_amd import particlepdf as pp
from oracle import models as om

logs = {}
orig_apply = pp.ParticlePDF._resample_apply
def apply_logged(self, idx, z_dev, factor, mean):
    torch.cuda.synchronize()
    b = self.__dict__.get("_rs_bufs")
    d = self.n_dims
    host_cov = np.array(self._moments_host[2 + 4 * d:2 + 4 * d + d * d]).tobytes()
    dev_block = self._moments_dev.cpu().numpy()
    dev_cov = dev_block[2 + 4 * d:2 + 4 * d + d * d].tobytes()
    entry = dict(host_cov=host_cov, dev_cov=dev_cov, host_eq_dev=host_cov == dev_cov, first=dev_block[:2 + 4 * d].tobytes(),
                 idx=zlib.crc32(idx.cpu().numpy().tobytes()), z=zlib.crc32(z_dev.cpu().numpy().tobytes()),
                 factor=factor.tobytes(), mean=mean.tobytes(), pin_i=tuple(int(v) for v in b["pin_i"]) if b else None,
                 total=float(b["pin_f"][0]) if b else None,
                 w=zlib.crc32(self._weights.tensor().cpu().numpy().tobytes()),
                 cdf=zlib.crc32(self._cdf_dev.cpu().numpy().tobytes()), uni=zlib.crc32(b["uni"].cpu().numpy().tobytes()) if b else None)
    logs.setdefault(id(self), []).append(entry)
    return orig_apply(self, idx, z_dev, factor, mean)
pp.ParticlePDF._resample_apply = apply_logged

mode = sys.argv[1] if len(sys.argv) > 1 else "points"
fails = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    g = np.random.default_rng(1000 + trial)
    n, ns = int(g.integers(1500, 15000)), 200
    prior = np.array([g.uniform(2, 4, n), g.uniform(400, 2000, n), g.normal(500, 300, n), g.exponential(150, n) + 5.0])
    xvals = np.linspace(1.5, 4.5, ns)
    kw = dict(scale=False, n_draws=18, selection_method="good", pickiness=4, utility_method="variance_approx", resample_threshold=0.9)
    a = obe.OptBayesExptSweeper(obe.models.lorentzian(), (xvals,), prior.copy(), (0.1,), 3, **kw)
    a2 = obe.OptBayesExptSweeper(obe.models.lorentzian(), (xvals,), prior.copy(), (0.1,), 3, **kw)
    if mode == "points":
        a._sweep_batch_inputs = lambda points: None
    a2._sweep_batch_inputs = lambda points: None
    a.rng, a2.rng = np.random.default_rng(trial), np.random.default_rng(trial)
    sim = np.random.default_rng(trial + 7)
    sx = xvals[20:170]
    sy = om.lorentzian((sx,), (3.1, 1000.0, 500.0, 150.0), (0.1,)) + 150.0 * sim.standard_normal(len(sx))
    logs.clear()
    a.pdf_update(((sx,), sy)); a2.pdf_update(((sx,), sy))
    la, lb = logs.get(id(a), []), logs.get(id(a2), [])
    same = a.rng.bit_generator.state == a2.rng.bit_generator.state and np.array_equal(np.array(a.particles), np.array(a2.particles))
    if not same:
        fails += 1
        print(f"trial {trial} n={n}: MISMATCH; resamples {len(la)} vs {len(lb)}")
        for k, (ea, eb) in enumerate(zip(la, lb)):
            diff = [key for key in ea if ea[key] != eb[key]]
            if diff:
                print("  first differing resample", k, "fields", diff, "host==dev:", ea["host_eq_dev"], eb["host_eq_dev"])
                if "host_cov" in diff:
                    print("   a host", np.frombuffer(ea["host_cov"]), "\n   a dev ", np.frombuffer(ea["dev_cov"]), "\n   b host", np.frombuffer(eb["host_cov"]), "\n   b dev ", np.frombuffer(eb["dev_cov"]))
                break
print("trials done, failures:", fails)
