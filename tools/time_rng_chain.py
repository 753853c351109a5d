"""The random-number chain of one c5 resample (524 288 uniforms + 5 242 880 normals), alone on one stream, 60 times:
run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations without the other chains of a resample beside it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                        # noqa: E402
from optbayesexpt_amd import _devrng, _lib          # noqa: E402
from optbayesexpt_amd.particlepdf import _ptr       # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
n, d = 524288, 10
n_normal = n * d
n_rel = n_normal + n_normal // 24 + 4096
_, h = _devrng.pcg64_state(np.random.default_rng(1))
u = torch.empty(n, dtype=torch.float64, device=dev)
z = torch.empty(n_normal, dtype=torch.float64, device=dev)
zws = torch.empty(int(lib.cdll.obe_ziggurat_workspace_bytes(n_rel)) // 8 + 1, dtype=torch.float64, device=dev)
tb = _devrng._tables(dev)
pin_i = _lib.pinned_array(2, np.int64)
lib.cdll.obe_defer_host_sync(1)
for _ in range(60):
    lib.call("obe_pcg64_uniforms_classify", _lib.host_ptr(h), n, n_rel, _ptr(u), _ptr(tb), _ptr(zws), zws.numel() * 8, None)
    lib.call("obe_ziggurat_finish", n_rel, n_normal, _ptr(z), _lib.host_ptr(pin_i), _ptr(zws), zws.numel() * 8, None)
torch.cuda.synchronize()
print("consumed, found:", pin_i.tolist())
