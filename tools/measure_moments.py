"""Duration of obe_moments (pass 1, and pass 1 + 2) by HIP events (developer aid).
    python tools/measure_moments.py   [OBE_VARIANT=<name> for a tools/build_variant.py library]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optbayesexpt_amd import _lib
if os.environ.get("OBE_VARIANT"):
    _lib._LIB = _lib.HipLib(os.path.join(ROOT, "tools", "_variants", f"libobe_hip_{os.environ['OBE_VARIANT']}.so"), allow_variant=True)
lib = _lib.load()
P = ctypes.c_void_p
g = torch.Generator(device="cuda").manual_seed(1)
for d, n in ((3, 1048576), (10, 524288), (3, 262144), (16, 1048576)):
    x = torch.randn((d, n), dtype=torch.float64, device="cuda", generator=g)
    w = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    w /= w.sum()
    out = torch.zeros(lib.moments_len(d), dtype=torch.float64, device="cuda")
    ws = torch.empty(lib.workspace_bytes(n, 1, 1, d) // 8 + 1, dtype=torch.float64, device="cuda")
    st = P(torch.cuda.current_stream().cuda_stream)
    res = []
    for cov in (0, 1):
        def run():
            lib.call("obe_moments", P(x.data_ptr()), n, d, n, P(w.data_ptr()), cov, P(out.data_ptr()), None,
                     P(ws.data_ptr()), ws.numel() * 8, st)
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        e1.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"{os.environ.get('OBE_VARIANT', 'tree'):6s} D={d:2d} N={n:8d}: pass1 {res[0]:7.1f} us ({8 * (d + 1) * n / res[0] / 1e6:5.2f} TB/s)   "
          f"pass1+2 {res[1]:7.1f} us   checksum {float(out.sum()):.6e}")
