"""K1 time of the hand-tuned Lorentzian vs the same formula as an expression model (developer aid)."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench, _expr_models
import optbayesexpt_amd as obe
from optbayesexpt_amd.particlepdf import _ptr
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
settings, prior, cons, true, sigma = bench.make_workload(cfg)
for label, model in (("hand-tuned", obe.models.lorentzian()), ("expression", _expr_models.expression_models()["lorentzian"])):
    o = obe.OptBayesExpt(model, settings, prior.copy(), cons, scale=False, utility_method="variance_full", default_noise_std=500.0)
    o.rng = np.random.default_rng(1)
    for _ in range(2):
        x = o.opt_setting(); o.pdf_update((x, 49000.0, sigma))
    for shifted in (1, 0):
        ms = ctypes.c_float()
        p, w = o._pw_tensors(); mom = o._moments_on_device()
        o._mlib.call("obe_sweep_kernel_time", o._model_struct, ctypes.c_void_p(o._settings_dev.data_ptr()), o._n_settings, o._n_settings,
                     _ptr(p), p.shape[1], p.shape[1], _ptr(w), _ptr(mom), shifted, _ptr(o._ws), o._ws_bytes, 3, ctypes.byref(ms), o._stream())
        print(f"{cfg} {label:11s} shifted={shifted}  K1 {ms.value:8.3f} ms  last idx {o.last_setting_index}")
