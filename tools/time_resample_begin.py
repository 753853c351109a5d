"""Wall time of the device side of a resample up to the gather (obe_resample_begin: three chains), unprofiled
(developer aid, GPU):  python tools/time_resample_begin.py [c5|c3]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                            # noqa: E402
import bench                                            # noqa: E402
from optbayesexpt_amd import _devrng, _lib              # noqa: E402
from optbayesexpt_amd.particlepdf import _P, _ptr       # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c5"
settings, prior, cons, true, sigma = bench.make_workload(cfg)
sv = (np.ascontiguousarray(settings[0][::64]),)
o = bench.build_obe(cfg, None, sv, prior.copy(), cons)
o.rng = np.random.default_rng(3)
g = np.random.default_rng(5)
w = g.exponential(1.0, o.n_particles)
o.particle_weights = w / w.sum()
n, d = o.n_particles, o.n_dims
p, wt = o._pw_tensors()
b = o._resample_buffers(n, d)
mlen = o._lib.moments_len(d)
stream = o._stream()
times = {}
for label, env in (("three chains", None),):
    ts = []
    for rep in range(30):
        st, h_state = _devrng.pcg64_state(o._rng)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o._lib.call("obe_resample_begin", _ptr(p), p.shape[1], d, n, _ptr(wt), _lib.host_ptr(h_state), 0, 0, 0, b["n_raw"],
                    _ptr(o._cdf_dev), _ptr(b["uni"]), _ptr(b["idx"][0]), _ptr(b["tables"]), _ptr(b["normals"]),
                    _ptr(b["zig_ws"]), b["zig_ws"].numel() * 8, _ptr(o._moments_dev), b["p_f"], b["p_i"],
                    None if b["aos"] is None else _ptr(b["aos"]), _ptr(o._ws), o._ws_bytes, stream)
        t1 = time.perf_counter()
        o._lib.call("obe_host_words_wait", _P(b["pin_f"].ctypes.data + 8), mlen, stream)      # the covariance has arrived
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0, t3 - t0))
    ts = np.array(ts[5:]) * 1e6
    print(f"{cfg} {n} x {d}: obe_resample_begin returns after {np.median(ts[:, 0]):6.1f} us, the covariance is on the host after "
          f"{np.median(ts[:, 1]):6.1f} us, all three chains have finished after {np.median(ts[:, 2]):6.1f} us (medians of 25)")
