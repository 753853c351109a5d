"""Where the fixed cost of opt_setting() goes on a 1/8 settings shard (developer aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from optbayesexpt_amd.dist import SettingsShard

class _Solo(SettingsShard):
    def _gather_records(self, record):          # no communication: only this rank's record counts
        g = torch.full((self.world_size, 4), float("-inf"), dtype=torch.float64)
        g[self.rank] = record.cpu()
        return g

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
settings, prior, cons, true, sigma = bench.make_workload(cfg)
obe = bench.build_obe(cfg, _Solo(rank=0, world_size=world) if world > 1 else None, settings, prior.copy(), cons)
obe.rng = np.random.default_rng(1)
for _ in range(3):
    x = obe.opt_setting(); obe.pdf_update((x, 49000.0, sigma))
def t(f, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
def touch():
    obe._weights.mark_device_written()
print("moments_on_device (stale each time) us:", t(lambda: (touch(), obe._moments_on_device())))
print("noise_var_device us:", t(lambda: obe._noise_var_device()))
print("cost_device us:", t(lambda: obe._cost_device()))
print("sweep_device(True) with fresh moments us:", t(lambda: obe._sweep_device(True)))
print("sweep_device(True) with stale moments us:", t(lambda: (touch(), obe._sweep_device(True))))
print("opt_setting us:", t(lambda: (touch(), obe.opt_setting())))
print("pdf_update (auto_resample off) us:", end=" ")
obe.tuning_parameters["auto_resample"] = False
print(t(lambda: obe.pdf_update(((3.0,), 49000.0, sigma))))
import ctypes
from optbayesexpt_amd import _lib
from optbayesexpt_amd.particlepdf import _ptr
lib = _lib.load(); ms = ctypes.c_float()
p, w = obe._pw_tensors(); mom = obe._moments_on_device()
n_local = obe._s_end - obe._s_begin
lib.call("obe_sweep_kernel_time", obe._model_struct, ctypes.c_void_p(obe._settings_dev.data_ptr()), obe._n_settings, n_local, _ptr(p), p.shape[1], p.shape[1], _ptr(w), _ptr(mom), 0, _ptr(obe._ws), obe._ws_bytes, 10, ctypes.byref(ms), obe._stream())
print("K1 kernel alone (unshifted) us:", 1e3 * ms.value)
