"""opt_setting() / pdf_update() split of the 10-parameter NoiseParameter config (developer aid)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
settings, prior, cons, true, sigma = bench.make_workload(cfg)
o = bench.build_obe(cfg, None, settings, prior.copy(), cons)
o.rng = np.random.default_rng(1); sim = np.random.default_rng(2)
t_opt, t_upd, res = [], [], []
fn = o.model_function
for cyc in range(25):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = o.opt_setting(); torch.cuda.synchronize(); t1 = time.perf_counter()
    y = float(np.atleast_1d(fn(x, true, cons))[0]) + sigma * sim.standard_normal()
    rec = (x, y) if cfg == "c5" else (x, y, sigma)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        t2 = time.perf_counter(); o.pdf_update(rec); torch.cuda.synchronize(); t3 = time.perf_counter()
    t_opt.append(t1 - t0); t_upd.append(t3 - t2); res.append(bool(o.just_resampled))
t_opt, t_upd, res = np.array(t_opt[5:]), np.array(t_upd[5:]), np.array(res[5:])
print(f"{cfg}: opt_setting median {1e3*np.median(t_opt):.3f} ms; pdf_update plain {1e3*np.median(t_upd[~res]) if (~res).any() else float('nan'):.3f} ms, "
      f"with resample {1e3*np.median(t_upd[res]) if res.any() else float('nan'):.3f} ms ({res.sum()} of {len(res)} cycles resampled)")
# where opt_setting's time goes beyond the sweep call itself
import optbayesexpt_amd.obe_base as ob
acc = {}
orig_call = o._mlib.call
def timed_call(name, *a):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig_call(name, *a); torch.cuda.synchronize()
    acc.setdefault(name, []).append(time.perf_counter() - t0); return r
o._mlib.call = timed_call
orig_lib_call = o._lib.call
if o._lib is not o._mlib:
    o._lib.call = timed_call
for cyc in range(10):
    x = o.opt_setting()
    y = float(np.atleast_1d(fn(x, true, cons))[0]) + sigma * sim.standard_normal()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        o.pdf_update((x, y) if cfg == "c5" else (x, y, sigma))
for k, v in acc.items():
    print(f"  {k:32s} calls {len(v):3d}  median {1e3*np.median(v):8.3f} ms  total/cycle {1e3*np.sum(v)/10:8.3f} ms")
print("  last_sweep", o.last_sweep)
