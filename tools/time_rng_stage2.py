"""Times obe_ziggurat_finish (start flags, scan, compaction) after one classification (developer aid)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from optbayesexpt_amd import _lib, _devrng
lib = _lib.load(); dev = torch.device("cuda", 0)
n, d = 524288, 10
n_u, n_normal = n, n * d
n_rel = n_normal + n_normal // 24 + 4096
rng = np.random.default_rng(1); st, h = _devrng.pcg64_state(rng)
u = torch.empty(n_u, dtype=torch.float64, device=dev); z = torch.empty(n_normal, dtype=torch.float64, device=dev)
ws = torch.empty(int(lib.cdll.obe_ziggurat_workspace_bytes(n_rel)) // 8 + 1, dtype=torch.float64, device=dev)
tb = _devrng._tables(dev); P = _lib.c_void_p
cons = np.zeros(2, dtype=np.int64)
lib.call("obe_pcg64_uniforms_classify", _lib.host_ptr(h), n_u, n_rel, P(u.data_ptr()), P(tb.data_ptr()), P(ws.data_ptr()), ws.numel() * 8, None)
lib.cdll.obe_defer_host_sync(1)
pin = _lib.pinned_array(2, np.int64)
timer = ctypes.c_void_p(); lib.call("obe_timer_create", ctypes.byref(timer)); ms = ctypes.c_float()
for rnd in range(4):
    lib.call("obe_timer_start", timer, None)
    for _ in range(20):
        lib.call("obe_ziggurat_finish", n_rel, n_normal, P(z.data_ptr()), _lib.host_ptr(pin), P(ws.data_ptr()), ws.numel() * 8, None)
    lib.call("obe_timer_stop", timer, None, ctypes.byref(ms))
print("dbg", os.environ.get("OBE_ZS_DBG", "0"), ":", ms.value * 1e3 / 20, "us per finish (starts + scan + compact)", pin)
