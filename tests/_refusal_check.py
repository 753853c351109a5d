"""Run as a child process with OBE_CONTROL_SLOTS=0 (tests/test_gpu_speculative.py): the library then has no
arrival counter for any stream, so obe_bayes_update_model_moments_enqueue() refuses every call.  ADVICE r4 #1:
the refusal must come BEFORE anything is launched — the weights untouched — and the package's fallback to the
synchronous form must give the plain path's bits (a refusal after pass A would apply the likelihood twice)."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import torch                                      # noqa: E402
import optbayesexpt_amd as obe                    # noqa: E402
from optbayesexpt_amd import _lib                 # noqa: E402
from optbayesexpt_amd.particlepdf import _ptr     # noqa: E402
import test_gpu_speculative as t                  # noqa: E402

assert os.environ.get("OBE_CONTROL_SLOTS") == "0"
# (1) the C ABI: refused with -1, nothing launched, the weights bit for bit what they were
o = t.make(obe, True, n_particles=50000, n_settings=512)
par, w = o._particles.tensor(), o._weights.tensor()
before = w.clone()
hp = o._hargs.ptr
args = (o._model_struct, _ptr(par), par.shape[1], o.n_particles, _ptr(w), hp(o._setting_array((3.0,))),
        hp(o._rec_y), hp(o._rec_s), None, 1, float("nan"))
try:
    o._mlib.call("obe_bayes_update_model_moments_enqueue", *args, _ptr(o._moments_dev), _ptr(o._ws), o._ws_bytes,
                 o._hargs.ptr_keep(o._upd_host), 1, 0.5, o._stream())
except _lib.ObeHipError as exc:
    assert "no control words" in str(exc), exc
else:
    raise SystemExit("the enqueue form was not refused although the library has no arrival counters")
torch.cuda.synchronize()
assert torch.equal(w, before), "a refused call changed the weights"
# ... and a workspace without the spare tail word is refused the same way (nothing launched)
os.environ["OBE_CONTROL_SLOTS"] = "0"
# (2) the package: speculation asked for, refused, falls back to the synchronous form — same bits as the plain path
a = t.cycles(t.make(obe, True), 12)
b = t.cycles(t.make(obe, False), 12)
t.same(a, b)
assert sum(e["resampled"] for e in a) >= 1
spec = t.make(obe, True)
t.cycles(spec, 3)
assert spec.speculation_state()["unavailable"], spec.speculation_state()
print("REFUSAL OK")
