"""The update as pdf_update() issues it (K2 + first moments: obe_bayes_update_model_moments) and the plain
K2 call, HIP events around back-to-back calls, at several cloud sizes (developer aid).

    python tools/measure_update.py [D=3|10]        OBE_VARIANT=<name>: a tools/build_variant.py library"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optbayesexpt_amd import _lib, models
if os.environ.get("OBE_VARIANT"):
    _lib._LIB = _lib.HipLib(os.path.join(ROOT, "tools", "_variants", f"libobe_hip_{os.environ['OBE_VARIANT']}.so"),
                            allow_variant=True)
from optbayesexpt_amd.particlepdf import _ptr
lib = _lib.load()
torch.cuda.set_device(0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 3
k = 1 if d == 3 else 7
model = models.lorentzian(k).struct(d, (0.1,))
g = np.random.default_rng(0)
timer = ctypes.c_void_p()
lib.call("obe_timer_create", ctypes.byref(timer))
st, yy, ss = np.zeros(4), np.zeros(4), np.ones(4) * 500.0
st[0], yy[0] = 3.0, 49500.0 if k == 1 else 1400.0
for n in (5000, 50000, 262144, 524288, 1 << 20, 1 << 22, 1 << 24):
    rows = [g.uniform(2, 4, (k, n)), g.uniform(400, 2000, (1, n)), g.normal(500, 1000, (1, n))]
    if d > k + 2:
        rows.append(g.exponential(500, (d - k - 2, n)) + 1.0)
    p = torch.from_numpy(np.vstack(rows)).cuda()
    w0 = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
    w = w0.clone()
    ws = torch.empty(lib.workspace_bytes(n, 64, 1, d) // 8 + 1, dtype=torch.float64, device="cuda")
    mom = torch.zeros(lib.moments_len(d), dtype=torch.float64, device="cuda")
    res = {}
    for fused in (True, False):
        def call():
            if fused:
                lib.call("obe_bayes_update_model_moments", model, _ptr(p), n, n, _ptr(w), _lib.host_ptr(st), _lib.host_ptr(yy),
                         _lib.host_ptr(ss), None, 1, float("nan"), _ptr(mom), _ptr(ws), ws.numel() * 8, None, None)
            else:
                lib.call("obe_bayes_update_model", model, _ptr(p), n, n, _ptr(w), _lib.host_ptr(st), _lib.host_ptr(yy),
                         _lib.host_ptr(ss), None, 1, float("nan"), _ptr(ws), ws.numel() * 8, None, None)
                lib.call("obe_moments", _ptr(p), n, d, n, _ptr(w), 0, _ptr(mom), None, _ptr(ws), ws.numel() * 8, None)
        us = []
        reps = 50 if n <= 1 << 20 else 10
        ms = ctypes.c_float(0.0)
        for rnd in range(7):
            lib.call("obe_timer_start", timer, None)
            for _ in range(reps):
                call()
            lib.call("obe_timer_stop", timer, None, ctypes.byref(ms))
            w.copy_(w0)
            if rnd >= 2:
                us.append(ms.value * 1e3 / reps)
        res[fused] = float(np.median(us))
    bytes_f = 8 * (k + 3) * n + 8 * n + 8 * (d + 1) * n + 8 * n          # A + B'
    bytes_u = 8 * (k + 3) * n + 8 * n + 16 * n + 8 * (d + 1) * n         # A + B + moments pass 1
    print(f"{os.environ.get('OBE_VARIANT', 'tree'):8s} D={d:2d} N={n:9d}: update+moments fused {res[True]:8.1f} us "
          f"({bytes_f / res[True] / 1e6:5.2f} TB/s)   update, then moments {res[False]:8.1f} us ({bytes_u / res[False] / 1e6:5.2f} TB/s)")
