import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
import optbayesexpt_amd as obe
settings, prior, cons, true, sigma = bench.make_workload("c3")
o = obe.OptBayesExpt(obe.models.lorentzian(), settings, prior.copy(), cons, scale=False, default_noise_std=sigma)
o.rng = np.random.default_rng(1); sim = np.random.default_rng(2)
ts = []
for cyc in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = o.opt_setting(); t1 = time.perf_counter()
    o.pdf_update((x, float(o.model_function(x, true, cons)) + sigma * sim.standard_normal(), sigma))
    torch.cuda.synchronize(); ts.append((t1 - t0, time.perf_counter() - t1, o.just_resampled))
print("reference-semantics cycle at 65536 x 1M particles, N_DRAWS=30:")
print(" opt_setting median %.3f ms, pdf_update (no resample) median %.3f ms, with resample %.3f ms" % (
    1e3 * np.median([t[0] for t in ts]), 1e3 * np.median([t[1] for t in ts if not t[2]]),
    1e3 * np.median([t[1] for t in ts if t[2]] or [0])))
