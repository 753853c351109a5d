"""K1 time of the hand-written Rabi and coil sweep forms vs the forms generated from the same
formulas (developer aid)."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import _expr_models
import optbayesexpt_amd as obe
from optbayesexpt_amd.particlepdf import _ptr
g = np.random.default_rng(0)
n = 262144
cases = {
    "rabi": ((np.linspace(0.02, 1, 128), np.linspace(-10, 10, 128)), np.array([g.uniform(1, 6, n), g.uniform(-4, 4, n)]),
             (100000.0, 0.01, 2.0), {}),
    "coil": ((np.logspace(-1, 1, 16384),), np.array([g.uniform(0.9, 1.1, n), g.uniform(0.08, 0.12, n), g.uniform(0.9, 1.1, n)]),
             (), {}),
}
gen = _expr_models.expression_models()
for name, (sv, prior, cons, kw) in cases.items():
    for label, model in (("hand-written", getattr(obe.models, name)()), ("generated", gen[name])):
        o = obe.OptBayesExpt(model, sv, prior.copy(), cons, scale=False, utility_method="variance_full",
                             default_noise_std=300.0, auto_resample=False)
        o.opt_setting()
        ms = ctypes.c_float()
        p, w = o._pw_tensors(); mom = o._moments_on_device()
        o._mlib.call("obe_sweep_kernel_time", o._model_struct, ctypes.c_void_p(o._settings_dev.data_ptr()), o._n_settings,
                     o._n_settings, _ptr(p), p.shape[1], p.shape[1], _ptr(w), _ptr(mom), 1, _ptr(o._ws), o._ws_bytes, 3,
                     ctypes.byref(ms), o._stream())
        print(f"{name:5s} {label:13s} {o._n_settings} settings x {n} particles (shifted): K1 {ms.value:8.3f} ms, "
              f"{o._n_settings * n / ms.value / 1e9:.1f} G evals/s, best idx {o.last_setting_index}")
