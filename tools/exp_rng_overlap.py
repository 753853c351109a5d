"""Experiment (round 6): what does the random-number chain of a resample cost the sweep when it runs
BESIDE it on a lower-priority stream?  (Would generating the next resample's uniforms and normals ahead of time —
during the sweep and the host round trips — hide them, or just move them?)

    python tools/exp_rng_overlap.py

One rank's c5 slice (2048 settings x 524 288 particles x 10 parameters).  Timed with host clocks around
torch.cuda.synchronize(): (a) 4 sweeps back to back; (b) the random chain alone (uniforms + classification, ziggurat
finish: 524 288 + 5.24 M numbers); (c) the chain enqueued on a LOW-priority stream, then the 4 sweeps on the default
stream; (d) the same with a default-priority side stream.  (c) - (a) is what the sweeps pay for the company."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                        # noqa: E402
import bench                                        # noqa: E402
from optbayesexpt_amd import _devrng, _lib          # noqa: E402
from optbayesexpt_amd.dist import SettingsShard     # noqa: E402
from optbayesexpt_amd.particlepdf import _ptr       # noqa: E402


def main():
    hip = ctypes.CDLL("libamdhip64.so")
    lo, hi = ctypes.c_int(), ctypes.c_int()
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    print(f"stream priorities: least {lo.value}, greatest {hi.value}")

    def make_stream(priority):
        s = ctypes.c_void_p()
        rc = hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, priority)      # hipStreamNonBlocking
        assert rc == 0, rc
        return s

    settings, prior, cons, true, sigma = bench.make_workload("c5")
    obe = bench.build_obe("c5", SettingsShard(rank=0, world_size=8), settings, prior.copy(), cons)
    lib = obe._lib
    mom = obe._moments_on_device()
    p, w = obe._pw_tensors()
    s_ptr = ctypes.c_void_p(obe._settings_dev.data_ptr() + 8 * obe._s_begin)
    ms = ctypes.c_float(0.0)

    def sweeps(iters=4):
        obe._mlib.call("obe_sweep_kernel_time", obe._model_struct, s_ptr, obe._n_settings, obe._s_end - obe._s_begin,
                       _ptr(p), p.shape[1], obe.n_particles, _ptr(w), _ptr(mom), _lib.OBE_SWEEP_SHIFTED, _ptr(obe._ws),
                       obe._ws_bytes, iters, ctypes.byref(ms), obe._stream())
        return ms.value

    n, d = obe.n_particles, obe.n_dims
    n_normal = n * d
    n_rel = n_normal + n_normal // 24 + 4096
    rng = np.random.default_rng(1)
    _, h = _devrng.pcg64_state(rng)
    dev = obe._device
    u = torch.empty(n, dtype=torch.float64, device=dev)
    z = torch.empty(n_normal, dtype=torch.float64, device=dev)
    zws = torch.empty(int(lib.cdll.obe_ziggurat_workspace_bytes(n_rel)) // 8 + 1, dtype=torch.float64, device=dev)
    tb = _devrng._tables(dev)
    pin_i = _lib.pinned_array(2, np.int64)

    def chain(stream):
        lib.call("obe_pcg64_uniforms_classify", _lib.host_ptr(h), n, n_rel, _ptr(u), _ptr(tb), _ptr(zws), zws.numel() * 8, stream)
        lib.call("obe_ziggurat_finish", n_rel, n_normal, _ptr(z), _lib.host_ptr(pin_i), _ptr(zws), zws.numel() * 8, stream)

    bench.warm_clocks(obe)
    low, same = make_stream(lo.value), make_stream(0)
    chain(low)
    torch.cuda.synchronize()

    def once(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0)

    # interleaved, so that clock drift hits all variants alike; (a) is pack + one warm launch + 2 timed launches
    runs = {"a": [], "b": [], "c": [], "d": []}
    for _ in range(40):
        runs["a"].append(once(lambda: sweeps(2)))
        runs["c"].append(once(lambda: (chain(low), sweeps(2))))
        runs["d"].append(once(lambda: (chain(same), sweeps(2))))
        runs["b"].append(once(lambda: chain(low)))
    med = {k: float(np.median(v)) for k, v in runs.items()}
    q = {k: (float(np.percentile(v, 25)), float(np.percentile(v, 75))) for k, v in runs.items()}
    print(f"(a) pack + 3 sweeps                                      : {med['a']:8.1f} us  (quartiles {q['a'][0]:.1f} .. {q['a'][1]:.1f})")
    print(f"(b) the random chain alone                               : {med['b']:8.1f} us  (quartiles {q['b'][0]:.1f} .. {q['b'][1]:.1f})")
    print(f"(c) chain on a LOW-priority stream, then (a)            : {med['c']:8.1f} us  (quartiles {q['c'][0]:.1f} .. {q['c'][1]:.1f})"
          f"   -> the sweeps pay {med['c'] - med['a']:6.1f} us for {med['b']:.0f} us of company")
    print(f"(d) chain on a default-priority side stream, then (a)   : {med['d']:8.1f} us  (quartiles {q['d'][0]:.1f} .. {q['d'][1]:.1f})"
          f"   -> {med['d'] - med['a']:6.1f} us")


if __name__ == "__main__":
    main()
