"""Continue the object's NumPy random stream on the device.

``resample()`` needs N uniforms and N*D standard normals from ``self.rng``.  When that is
a ``numpy.random.Generator`` over ``PCG64`` (what ``default_rng()`` gives) the very same
numbers are produced by the kernels of csrc/obe_rng.hip — bit-identical uniforms and
normals, and the exact count of raw 64-bit values consumed — and the host generator is
then moved to the state numpy itself would have reached.  Any other generator (legacy
``np.random``, MT19937, a user wrapper) falls back to calling the generator on the host,
which is what the reference does; either way the numbers come from the caller's stream.
"""
import os

import numpy as np
import torch

from . import _lib

_P = _lib.c_void_p
_MASK64 = (1 << 64) - 1
_MASK128 = (1 << 128) - 1
_PCG_MULT = 0x2360ed051fc65da44385df649fccf645
MIN_DEVICE_DRAWS = 4096          # below this (~1000 particles x 4) the host call beats the ~25 launches (pipelined: 0.36 vs 0.46 ms per resample cycle at 5000 particles)

_TABLES = {}                     # device -> uint8 tensor: ki[256] u64 | wi[256] f64 | fi[256] f64


def _tables(device):
    key = str(device)
    if key not in _TABLES:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ziggurat_tables.npz")
        z = np.load(path)
        blob = np.concatenate([z["ki"].astype(np.uint64).view(np.uint8),
                               z["wi"].astype(np.float64).view(np.uint8),
                               z["fi"].astype(np.float64).view(np.uint8)])
        _TABLES[key] = torch.from_numpy(blob.copy()).to(device)
    return _TABLES[key]


def pcg64_state(rng):
    """(state dict, uint64[4] = state_hi, state_lo, inc_hi, inc_lo) or None if ``rng`` is
    not a Generator over PCG64."""
    if not isinstance(rng, np.random.Generator):
        return None
    bg = rng.bit_generator
    if type(bg).__name__ != "PCG64":
        return None
    st = bg.state
    s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    return st, np.array([s >> 64, s & _MASK64, inc >> 64, inc & _MASK64], dtype=np.uint64)


def advance(rng, st, n_consumed):
    """Move the host generator ``n_consumed`` raw draws forward (LCG jump-ahead on Python
    integers), keeping the buffered 32-bit half-draw exactly as numpy's own calls would."""
    state, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    acc_mult, acc_plus, cur_mult, cur_plus, delta = 1, 0, _PCG_MULT, inc, int(n_consumed)
    while delta > 0:
        if delta & 1:
            acc_mult = (acc_mult * cur_mult) & _MASK128
            acc_plus = (acc_plus * cur_mult + cur_plus) & _MASK128
        cur_plus = ((cur_mult + 1) * cur_plus) & _MASK128
        cur_mult = (cur_mult * cur_mult) & _MASK128
        delta >>= 1
    st["state"]["state"] = (acc_mult * state + acc_plus) & _MASK128
    rng.bit_generator.state = st


class DeviceStream:
    """Raw PCG64 values generated on the device for one resample: ``n_uniform`` uniforms
    followed by ``n_normal`` ziggurat normals, exactly as numpy would draw them."""

    def __init__(self, lib, device, stream, rng, n_uniform, n_normal):
        got = pcg64_state(rng)
        assert got is not None
        if not isinstance(lib, _lib.DeviceBound):
            lib = _lib.DeviceBound(lib, torch.device(device))
        self.lib, self.device, self.stream, self.rng = lib, device, stream, rng
        self.st, self.h_state = got
        self.n_uniform, self.n_normal = int(n_uniform), int(n_normal)
        self.margin = self.n_normal // 24 + 4096 if n_normal else 0     # ~2.2 % is consumed extra
        self._generate()

    def _generate(self):
        self.n_raw = self.n_uniform + self.n_normal + self.margin
        self.raw = torch.empty(self.n_raw, dtype=torch.int64, device=self.device)
        self.lib.call("obe_pcg64_raw", _lib.host_ptr(self.h_state), self.n_raw, _P(self.raw.data_ptr()),
                      self.stream)

    def uniforms(self):
        out = torch.empty(self.n_uniform, dtype=torch.float64, device=self.device)
        self.lib.call("obe_pcg64_uniform", _P(self.raw.data_ptr()), self.n_uniform, _P(out.data_ptr()),
                      self.stream)
        return out

    def normals(self):
        """(n_normal,) device tensor; also advances the host generator past everything
        this stream handed out (uniforms + the raw values the normals consumed)."""
        consumed = np.zeros(1, dtype=np.int64)
        out = torch.empty(self.n_normal, dtype=torch.float64, device=self.device)
        tables = _tables(self.device)
        while True:
            n_tail = self.n_raw - self.n_uniform
            ws_bytes = int(self.lib.cdll.obe_ziggurat_workspace_bytes(n_tail))
            ws = torch.empty(ws_bytes // 8 + 1, dtype=torch.float64, device=self.device)
            with self.lib.guard():
                rc = self.lib.cdll.obe_ziggurat_normal(_P(self.raw.data_ptr() + 8 * self.n_uniform), n_tail, 0,
                                                       _P(tables.data_ptr()), self.n_normal, _P(out.data_ptr()),
                                                       _lib.host_ptr(consumed), _P(ws.data_ptr()),
                                                       ws.numel() * 8, self.stream)
            if rc == 0:
                break
            if rc != 1:
                raise _lib.ObeHipError(f"obe_ziggurat_normal failed ({rc}): {self.lib.last_error()}")
            self.margin = 2 * self.margin + 65536          # unlucky stream: regenerate with more head-room
            self._generate()
        advance(self.rng, self.st, self.n_uniform + int(consumed[0]))
        return out

    def normals_deferred(self, pinned_i64):
        """Launch the normals without waiting for the device: returns the (n_normal,) device tensor;
        ``pinned_i64[0:2]`` receive {raw values consumed, normals found} once the stream has been
        synchronised, and ``finish_normals()`` then validates them and advances the host generator.
        (obe_defer_host_sync must be on.)"""
        out = torch.empty(self.n_normal, dtype=torch.float64, device=self.device)
        tables = _tables(self.device)
        n_tail = self.n_raw - self.n_uniform
        ws_bytes = int(self.lib.cdll.obe_ziggurat_workspace_bytes(n_tail))
        self._zig_ws = torch.empty(ws_bytes // 8 + 1, dtype=torch.float64, device=self.device)
        self.lib.call("obe_ziggurat_normal", _P(self.raw.data_ptr() + 8 * self.n_uniform), n_tail, 0,
                      _P(tables.data_ptr()), self.n_normal, _P(out.data_ptr()), _P(pinned_i64.data_ptr()),
                      _P(self._zig_ws.data_ptr()), self._zig_ws.numel() * 8, self.stream)
        return out

    def finish_normals(self, pinned_i64):
        """After the stream synchronisation that follows ``normals_deferred()``: True and the host
        generator advanced if the raw buffer was long enough; False if the caller has to draw the
        normals again with ``normals()`` (which regenerates a longer buffer)."""
        consumed, found = int(pinned_i64[0]), int(pinned_i64[1])
        n_tail = self.n_raw - self.n_uniform
        if self.lib.cdll.obe_ziggurat_check(consumed, found, self.n_normal, n_tail, 0) != 0:
            self.margin = 2 * self.margin + 65536
            self._generate()
            return False
        advance(self.rng, self.st, self.n_uniform + consumed)
        return True

    def finish_uniform_only(self):
        advance(self.rng, self.st, self.n_uniform)
