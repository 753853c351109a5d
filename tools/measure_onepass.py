"""The fused update (K2 + first moments) in one launch against the two-launch form (developer aid, GPU):

    python tools/build_variant.py onepass -DOBE_ONE_PASS_UPDATE
    OBE_VARIANT=onepass OBE_FIRST_MOM_PER_CU=3 python tools/measure_onepass.py [D=3|10|4|11]

For several cloud sizes: (1) weights, the K3 block and the host block of both forms compared BIT FOR BIT
(obe_update_one_pass(0 / 1) on the same inputs; obe_update_one_pass(-1) tells which form actually ran),
(2) HIP events around back-to-back calls of each form, median of 7 rounds."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                          # noqa: E402
from optbayesexpt_amd import _lib, models             # noqa: E402
from optbayesexpt_amd.particlepdf import _ptr         # noqa: E402

if os.environ.get("OBE_VARIANT"):
    _lib._LIB = _lib.HipLib(os.path.join(ROOT, "tools", "_variants", f"libobe_hip_{os.environ['OBE_VARIANT']}.so"),
                            allow_variant=True)
lib = _lib.load()
torch.cuda.set_device(0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 3
k = {3: 1, 4: 1, 10: 7, 11: 7, 9: 7}[d]
noise = d == k + 3
model = models.lorentzian(k).struct(d, (0.1,))
g = np.random.default_rng(0)
timer = ctypes.c_void_p()
lib.call("obe_timer_create", ctypes.byref(timer))
st, yy, ss = np.zeros(4), np.zeros(4), np.ones(4) * 500.0
st[0], yy[0] = 3.0, 49500.0 if k == 1 else 1400.0
rows_arg = None
if noise:
    rows_arg = np.zeros(16, dtype=np.int32)
    rows_arg[0] = d - 1
print(f"first-moment grid per CU: {os.environ.get('OBE_FIRST_MOM_PER_CU', 'default')}")
for n in (5000, 50000, 262144, 524288, 1 << 20, 1179648, 1 << 21):
    rows = [g.uniform(2, 4, (k, n)), g.uniform(400, 2000, (1, n)), g.normal(500, 1000, (1, n))]
    if noise:
        rows.append(g.exponential(500, (1, n)) + 1.0)
    p = torch.from_numpy(np.vstack(rows)).cuda()
    w0 = torch.from_numpy(g.exponential(1.0, n)).cuda()
    w0 /= w0.sum()
    ws = torch.empty(lib.workspace_bytes(n, 64, 1, d) // 8 + 1, dtype=torch.float64, device="cuda")
    host = _lib.pinned_array(2 + lib.moments_len(d))
    res, out, forms = {}, {}, {}
    for one in (0, 1):
        lib.cdll.obe_update_one_pass(one)
        w = w0.clone()
        mom = torch.zeros(lib.moments_len(d), dtype=torch.float64, device="cuda")

        def call(h=None):
            lib.call("obe_bayes_update_model_moments", model, _ptr(p), n, n, _ptr(w), _lib.host_ptr(st), _lib.host_ptr(yy),
                     None if noise else _lib.host_ptr(ss), None if not noise else _lib.host_ptr(rows_arg), 1,
                     float("nan"), _ptr(mom), _ptr(ws), ws.numel() * 8, h, None)
        host[:] = 0.0
        call(_lib.host_ptr(host))
        forms[one] = lib.cdll.obe_update_one_pass(-1)
        torch.cuda.synchronize()
        out[one] = (w.cpu().numpy().copy(), mom.cpu().numpy()[:2 + 4 * d].copy(), host[:4 + 4 * d].copy())
        us, ms = [], ctypes.c_float(0.0)
        reps = 50
        for rnd in range(9):
            w.copy_(w0)
            lib.call("obe_timer_start", timer, None)
            for _ in range(reps):
                call()
            lib.call("obe_timer_stop", timer, None, ctypes.byref(ms))
            if rnd >= 2:
                us.append(ms.value * 1e3 / reps)
        res[one] = float(np.median(us))
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(out[0], out[1]))
    nbytes = 8 * (d + 1) * n + 8 * n
    print(f"D={d:2d} N={n:8d}: two launches {res[0]:7.2f} us   one launch {res[1]:7.2f} us (form {forms[1]}; "
          f"{nbytes / res[1] / 1e6:5.2f} TB/s on {nbytes / 1e6:.1f} MB)   bits identical: {same}")
    assert same, "the one-launch form differs from the two-launch form"
lib.cdll.obe_update_one_pass(0)
