"""K1 launch duration: back-to-back vs isolated launches, optionally with an idle gap (developer aid).

    python tools/measure_sweep_launch.py [c3|c5|c2] [iters]

Environment: OBE_SWEEP_BLOCKS / OBE_SWEEP_SPT (grid tuning), OBE_TIME_GAP_US (host sleep before
each isolated launch).
"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, os.environ["OBE_AB_ROOT"]) if os.environ.get("OBE_AB_ROOT") else ROOT)   # A/B: another checkout
import torch
import bench
if os.environ.get("OBE_VARIANT"):      # a library built by tools/build_variant.py instead of the product one
    from optbayesexpt_amd import _lib as _l
    _l._LIB = _l.HipLib(os.path.join(ROOT, "tools", "_variants", f"libobe_hip_{os.environ['OBE_VARIANT']}.so"), allow_variant=True)
from optbayesexpt_amd.particlepdf import _ptr

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
settings, prior, cons, true, sigma = bench.make_workload(cfg)
obe = bench.build_obe(cfg, None, settings, prior.copy(), cons)
obe.rng = np.random.default_rng(1234)
sim = np.random.default_rng(4321)
for cyc in range(3):
    x = obe.opt_setting()
    y = float(np.atleast_1d(obe.model_function(x, true, cons))[0]) + sigma * sim.standard_normal()
    obe.pdf_update((x, y, sigma) if cfg != "c5" else (x, y))
mom = obe._moments_on_device()
p, w = obe._pw_tensors()
ns = obe.allsettings.shape[1]
n_local = int(os.environ.get("OBE_NS_LOCAL", ns))       # a settings shard: the first n_local settings
s_ptr = ctypes.c_void_p(obe._settings_dev.data_ptr())


def k1(n, shifted=0):
    ms = ctypes.c_float(0.0)
    obe._mlib.call("obe_sweep_kernel_time", obe._model_struct, s_ptr, ns, n_local, _ptr(p), p.shape[1], p.shape[1],
                   _ptr(w), _ptr(mom), shifted, _ptr(obe._ws), obe._ws_bytes, n, ctypes.byref(ms), obe._stream())
    return ms.value


tag = f"{os.environ.get('OBE_AB_ROOT', os.environ.get('OBE_VARIANT', 'tree'))} {cfg} ns_local={n_local} blocks={os.environ.get('OBE_SWEEP_BLOCKS', 'default')} spt={os.environ.get('OBE_SWEEP_SPT', 'default')} " \
      f"gap_us={os.environ.get('OBE_TIME_GAP_US', '0')}"
b2b = [k1(iters) for _ in range(3)]
iso = [k1(-iters) for _ in range(3)] if not os.environ.get("OBE_AB_ROOT") else [float("nan")]
print(f"{tag}: back-to-back {min(b2b):.3f} ms  isolated {min(iso):.3f} ms  (all: {b2b} {iso})")
if not os.environ.get("OBE_AB_ROOT"):
    sh = [k1(iters, 1) for _ in range(3)]
    print(f"{tag}: shifted variant back-to-back {min(sh):.3f} ms")
