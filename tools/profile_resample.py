"""Wall-clock pieces of one resample() at 1M particles (developer aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from optbayesexpt_amd import _devrng
settings, prior, cons, true, sigma = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
import optbayesexpt_amd as obe
pdf = obe.ParticlePDF(prior.copy(), scale=False)
g = np.random.default_rng(3)
w = g.exponential(1.0, prior.shape[1]); w /= w.sum()
pdf.rng = np.random.default_rng(5)
def sync(): torch.cuda.synchronize()
for rep in range(3):
    pdf.particle_weights = w
    pdf._weights.tensor(); sync()
    t0 = time.perf_counter(); pdf.resample(); sync(); print("resample total ms", 1e3 * (time.perf_counter() - t0))
n, d = pdf.n_particles, pdf.n_dims
pdf.particle_weights = w; pdf._weights.tensor(); sync()
T = {}
def lap(name, t0):
    sync(); T[name] = 1e3 * (time.perf_counter() - t0)
t0 = time.perf_counter(); rs = pdf._device_stream(n, n * d); lap("device_stream (state + raw kernel + alloc)", t0)
t0 = time.perf_counter(); cdf = pdf._cdf(); lap("cdf (3 kernels + D2H total)", t0)
t0 = time.perf_counter(); idx = pdf._draw_indices(n, rs); lap("uniforms + search", t0)
t0 = time.perf_counter(); m = pdf._moments(True); lap("moments + cov + D2H", t0)
cov = m[2 + 4 * d:2 + 4 * d + d * d].reshape((d, d))
t0 = time.perf_counter(); u, s, vh = np.linalg.svd(0.04 * cov); np.allclose(np.dot(vh.T * s, vh), 0.04 * cov); lap("svd + psd check (host)", t0)
t0 = time.perf_counter(); z = rs.normals(); lap("normals (classify/starts/scan/compact + D2H + advance)", t0)
st = _devrng.pcg64_state(pdf.rng)[0]
t0 = time.perf_counter(); _devrng.advance(pdf.rng, st, 4200000); lap("  of which: host advance()", t0)
t0 = time.perf_counter(); _devrng.pcg64_state(pdf.rng); lap("  pcg64_state()", t0)
for k, v in T.items(): print(f"{k:60s} {v:8.3f} ms")
