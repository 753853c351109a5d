"""`pdf_update()`'s library call (obe_bayes_update_model_moments) N times back to back on one cloud, for a
kernel trace:   rocprofv3 --kernel-trace --stats -d gpurun_out/upd -o upd -- python3 tools/trace_update.py [D] [N]
(developer aid; per-kernel durations of the update at c3 / c5 size)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from optbayesexpt_amd import _lib, models
from optbayesexpt_amd.particlepdf import _ptr
lib = _lib.load()
torch.cuda.set_device(0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 20 if d == 3 else 1 << 19)
k = 1 if d == 3 else 7
model = models.lorentzian(k).struct(d, (0.1,))
g = np.random.default_rng(0)
rows = [g.uniform(2, 4, (k, n)), g.uniform(400, 2000, (1, n)), g.normal(500, 1000, (1, n))]
if d > k + 2:
    rows.append(g.exponential(500, (d - k - 2, n)) + 1.0)
p = torch.from_numpy(np.vstack(rows)).cuda()
w0 = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
w = w0.clone()
ws = torch.zeros(lib.workspace_bytes(n, 64, 1, d) // 8 + 1, dtype=torch.float64, device="cuda")
mom = torch.zeros(lib.moments_len(d), dtype=torch.float64, device="cuda")
st, yy, ss = np.zeros(4), np.zeros(4), np.ones(4) * 500.0
st[0], yy[0] = 3.0, 49500.0 if k == 1 else 1400.0
host = _lib.pinned_array(2 + lib.moments_len(d))
for rnd in range(6):
    for _ in range(50):
        lib.call("obe_bayes_update_model_moments", model, _ptr(p), n, n, _ptr(w), _lib.host_ptr(st), _lib.host_ptr(yy),
                 _lib.host_ptr(ss), None, 1, float("nan"), _ptr(mom), _ptr(ws), ws.numel() * 8, _lib.host_ptr(host), None)
    w.copy_(w0)
torch.cuda.synchronize()
print("done", host[:2])
