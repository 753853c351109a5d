"""K2 call time vs particle count and weight state (developer aid)."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import optbayesexpt_amd as obe
from optbayesexpt_amd import _lib
from optbayesexpt_amd.particlepdf import _ptr
lib = _lib.load()
g = np.random.default_rng(0)
for n in (5000, 65536, 262144, 1048576, 4194304):
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    o = obe.OptBayesExpt(obe.models.lorentzian(), (np.linspace(1.5, 4.5, 64),), prior, (0.1,), auto_resample=False)
    p, w = o._pw_tensors()
    st, yy, ss = np.zeros(4), np.zeros(4), np.ones(4) * 500.0
    st[0], yy[0] = 3.0, 49500.0
    for label, reset in (("fresh weights each call", True), ("weights left to collapse", False)):
        w0 = w.clone()
        torch.cuda.synchronize()
        ts = []
        for rep in range(30):
            if reset:
                w.copy_(w0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            o._mlib.call("obe_bayes_update_model", o._model_struct, _ptr(p), p.shape[1], n, _ptr(w), _lib.host_ptr(st),
                         _lib.host_ptr(yy), _lib.host_ptr(ss), None, 1, float("nan"), _ptr(o._ws), o._ws_bytes, None, o._stream())
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"N={n:8d} {label:26s} median {1e6*np.median(ts):7.1f} us  min {1e6*min(ts):7.1f}  max {1e6*max(ts):7.1f}  sum w = {float(w.sum()):.3g}")
        w.copy_(w0)
