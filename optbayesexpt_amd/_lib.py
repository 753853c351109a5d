"""ctypes binding of libobe_hip.so (the C ABI declared in include/obe_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded the
import of the product classes fails with an explicit error.  PyTorch is imported first
so that the process ends up with a single HIP runtime (torch bundles libamdhip64.so.7,
the library's NEEDED entry resolves to the already-loaded copy).
"""
import ctypes
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libobe_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "obe_hip.h")



def _header_constants(path=HEADER_PATH):
    """The #define'd integers of include/obe_hip.h (one source of truth for limits and the ABI version)."""
    text = open(path).read()
    return {k: int(v) for k, v in re.findall(r"^#define\s+(OBE_[A-Z_]+)\s+(-?\d+)\s*(?:/\*.*)?$", text, flags=re.M)}


_H = _header_constants()
OBE_ABI_VERSION = _H["OBE_ABI_VERSION"]
OBE_MAX_CONSTS = _H["OBE_MAX_CONSTS"]
OBE_MAX_CHANNELS = _H["OBE_MAX_CHANNELS"]
OBE_MAX_SETDIMS = _H["OBE_MAX_SETDIMS"]
OBE_MAX_DIMS = _H["OBE_MAX_DIMS"]
OBE_FAST_DIMS = _H["OBE_FAST_DIMS"]                  # cloud kernels compiled for the exact row count up to here
OBE_CLOUD_MAX_DIMS = _H["OBE_CLOUD_MAX_DIMS"]        # what the tiled cloud kernels take (ParticlePDF alone)
OBE_WS_RESULT_OFFSET = 2      # doubles; include/obe_hip.h
HOST_SENTINEL = 0x7ff8c0dec0dec0de     # the value of an armed host result word (csrc/obe_common.h: kHostSentinel)
OBE_SWEEP_SHIFTED, OBE_SWEEP_SAFE, OBE_SWEEP_SPECULATIVE, OBE_SWEEP_NOWAIT = 1, 2, 8, 16      # bits of obe_sweep_utility's `shifted` argument

c_void_p, c_int, c_int32, c_int64, c_double = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int32,
                                               ctypes.c_int64, ctypes.c_double)


class ObeHipError(RuntimeError):
    """A libobe_hip call returned a non-zero status (``status``: -1 = the call was refused on its arguments
    BEFORE anything was launched, include/obe_hip.h; anything else is a hipError_t of a launch or a wait)."""

    def __init__(self, message, status=None):
        RuntimeError.__init__(self, message)
        self.status = status

    @property
    def refused_before_launch(self):
        return self.status == -1


class ObeModelStruct(ctypes.Structure):
    """Mirror of ``struct obe_model`` (include/obe_hip.h)."""
    _fields_ = [("id", c_int32), ("aux", c_int32), ("n_params", c_int32),
                ("n_setdims", c_int32), ("n_channels", c_int32), ("n_consts", c_int32),
                ("consts", c_double * OBE_MAX_CONSTS)]


def declared_symbols(header_path=HEADER_PATH):
    """Every function name the header declares (used by the symbol-export test)."""
    text = open(header_path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(obe_[a-z0-9_]+)\s*\(", text)))


_P = c_void_p   # any pointer (device or host) is passed as an integer address

_SIGNATURES = {
    "obe_abi_version": (c_int, []),
    "obe_last_error": (ctypes.c_char_p, []),
    "obe_defer_host_sync": (c_int, [c_int32]),
    "obe_update_one_pass": (c_int, [c_int32]),
    "obe_strict_sums": (c_int, [c_int32]),
    "obe_source_fingerprint": (ctypes.c_char_p, []),
    "obe_model_validate": (c_int, [ctypes.POINTER(ObeModelStruct)]),
    "obe_device_info": (c_int, [ctypes.c_char_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int64)]),
    "obe_workspace_bytes": (c_int64, [c_int64, c_int64, c_int32, c_int32]),
    "obe_bayes_update_model": (c_int, [ctypes.POINTER(ObeModelStruct), _P, c_int64, c_int64, _P, _P, _P, _P, _P,
                                       c_int32, c_double, _P, c_int64, _P, _P]),
    "obe_bayes_update_model_moments": (c_int, [ctypes.POINTER(ObeModelStruct), _P, c_int64, c_int64, _P, _P, _P, _P, _P,
                                               c_int32, c_double, _P, _P, c_int64, _P, _P]),
    "obe_bayes_update_model_moments_enqueue": (c_int, [ctypes.POINTER(ObeModelStruct), _P, c_int64, c_int64, _P, _P, _P,
                                                       _P, _P, c_int32, c_double, _P, _P, c_int64, _P, c_int32,
                                                       c_double, _P]),
    "obe_bayes_update_sweep": (c_int, [ctypes.POINTER(ObeModelStruct), _P, c_int64, c_int64, _P, _P, _P, _P, _P,
                                       c_int32, c_double, c_int64, c_int32, c_double, _P, c_int64, _P, _P]),
    "obe_bayes_update_y": (c_int, [_P, c_int64, c_int32, _P, c_int64, c_int64, _P, _P, _P, _P, c_int32,
                                   c_double, _P, c_int64, _P, _P]),
    "obe_bayes_update_lik": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P]),
    "obe_likelihood_y": (c_int, [_P, c_int64, c_int32, _P, c_int64, c_int64, _P, _P, _P, c_int32, c_double,
                                 _P, _P]),
    "obe_weight_sums": (c_int, [_P, c_int64, _P, c_int64, _P, _P]),
    "obe_eval_over_particles": (c_int, [ctypes.POINTER(ObeModelStruct), _P, c_int64, c_int64, _P, _P, c_int64, _P]),
    "obe_eval_over_settings": (c_int, [ctypes.POINTER(ObeModelStruct), _P, c_int64, c_int64, _P, _P, c_int64, _P]),
    "obe_moments_len": (c_int64, [c_int32]),
    "obe_moments": (c_int, [_P, c_int64, c_int32, c_int64, _P, c_int32, _P, _P, _P, c_int64, _P]),
    "obe_weight_cdf": (c_int, [_P, c_int64, c_int32, _P, _P, _P, c_int64, _P]),
    "obe_draw_indices": (c_int, [_P, c_int64, c_int32, c_int32, _P, _P, c_int32, _P, _P, _P, c_int64, _P]),
    "obe_systematic_indices": (c_int, [_P, c_int64, c_double, c_int64, _P, _P]),
    "obe_cdf_search": (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_int64, _P]),
    "obe_gather_columns": (c_int, [_P, c_int64, c_int32, c_int64, _P, c_int64, _P, c_int64, _P]),
    "obe_resample_particles": (c_int, [_P, c_int64, c_int32, c_int64, _P, _P, _P, _P, c_double, c_int32,
                                       _P, c_int64, _P, _P, c_int64, _P]),
    "obe_mask_nonpositive": (c_int, [_P, c_int64, c_int64, _P, c_int32, _P, _P, _P, c_int64, _P]),
    "obe_mask_nonpositive_moments": (c_int, [_P, c_int64, c_int32, c_int64, _P, c_int32, _P, _P, _P, _P, _P, c_int64,
                                             _P]),
    "obe_resample_begin": (c_int, [_P, c_int64, c_int32, c_int64, _P, _P, c_int32, c_int32, c_int32, c_int64,
                                   _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P, c_int64, _P]),
    "obe_resample_randoms_enqueue": (c_int, [_P, c_int64, c_int32, c_int64, _P, _P, _P, _P, c_int64, _P, _P]),
    "obe_resample_particles_aos": (c_int, [_P, c_int32, c_int64, _P, _P, _P, _P, c_double, c_int32, _P, c_int64, _P, _P]),
    "obe_resample_particles_aos_masked": (c_int, [_P, c_int32, c_int64, _P, _P, _P, _P, c_double, c_int32, _P, c_int64, _P,
                                                  _P, c_int32, _P, _P]),
    "obe_mask_renorm_moments": (c_int, [_P, c_int64, c_int32, c_int64, _P, _P, _P, _P, _P, _P, c_int64, _P]),
    "obe_pcg64_uniforms_classify": (c_int, [_P, c_int64, c_int64, _P, _P, _P, c_int64, _P]),
    "obe_ziggurat_finish": (c_int, [c_int64, c_int64, _P, _P, _P, c_int64, _P]),
    "obe_noise_var_from_moments": (c_int, [_P, c_int32, _P, c_int32, _P, _P]),
    "obe_cumsum": (c_int, [_P, c_int64, c_int32, _P, _P, c_int64, _P]),
    "obe_interval_utility": (c_int, [_P, c_int64, _P, c_int64, _P, c_double, _P, _P]),
    "obe_power_normalize": (c_int, [_P, c_int64, c_double, _P, _P, c_int64, _P]),
    "obe_sweep_utility": (c_int, [ctypes.POINTER(ObeModelStruct), _P, c_int64, c_int64, _P, c_int64, c_int64,
                                  _P, _P, c_int64, _P, c_int32, _P, c_int64, _P, c_double, _P, _P, _P, _P, _P,
                                  _P, c_int64, _P]),
    "obe_yspace_variance": (c_int, [_P, c_int64, c_int32, c_int64, _P, _P]),
    "obe_utility_argmax": (c_int, [_P, c_int32, c_int64, _P, c_int64, _P, c_double, _P, _P, _P, _P, c_int64, _P]),
    "obe_argmax": (c_int, [_P, c_int64, _P, _P, _P, c_int64, _P]),
    "obe_eval_draws": (c_int, [ctypes.POINTER(ObeModelStruct), _P, c_int64, c_int64, _P, c_int64, c_int64, _P,
                               c_int64, _P, _P]),
    "obe_yspace_add_noise": (c_int, [_P, c_int64, c_int32, c_int64, _P, _P]),
    "obe_yspace_maxmin": (c_int, [_P, c_int64, c_int64, _P, _P]),
    "obe_yspace_entropy": (c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P]),
    "obe_kld_utility": (c_int, [_P, c_int32, c_int64, _P, _P, _P]),
    "obe_pcg64_raw": (c_int, [_P, c_int64, _P, _P]),
    "obe_pcg64_uniform": (c_int, [_P, c_int64, _P, _P]),
    "obe_ziggurat_workspace_bytes": (c_int64, [c_int64]),
    "obe_ziggurat_normal": (c_int, [_P, c_int64, c_int64, _P, c_int64, _P, _P, _P, c_int64, _P]),
    "obe_ziggurat_check": (c_int, [c_int64, c_int64, c_int64, c_int64, c_int64]),
    "obe_timer_create": (c_int, [ctypes.POINTER(c_void_p)]),
    "obe_timer_start": (c_int, [_P, _P]),
    "obe_timer_stop": (c_int, [_P, _P, ctypes.POINTER(ctypes.c_float)]),
    "obe_timer_destroy": (c_int, [_P]),
    "obe_sweep_settings_per_lane": (c_int, [c_int64]),
    "obe_sweep_settings_per_lane_for": (c_int, [c_int64, c_int64]),
    "obe_host_device_pointer": (c_int, [_P, ctypes.POINTER(c_void_p)]),
    "obe_host_words_arm": (c_int, [_P, c_int64]),
    "obe_host_words_wait": (c_int, [_P, c_int64, _P]),
    "obe_host_word_arm": (c_int, [_P]),
    "obe_host_word_wait": (c_int, [_P, _P]),
    "obe_sweep_timing": (c_int, [c_int32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64)]),
    "obe_sweep_kernel_time": (c_int, [ctypes.POINTER(ObeModelStruct), _P, c_int64, c_int64, _P, c_int64, c_int64,
                                      _P, _P, c_int32, _P, c_int64, c_int32, ctypes.POINTER(ctypes.c_float), _P]),
}


# entry points whose code depends on the model: a plugin library serves these
MODEL_ENTRY_POINTS = ("obe_model_validate", "obe_workspace_bytes", "obe_sweep_settings_per_lane",
                      "obe_sweep_settings_per_lane_for", "obe_bayes_update_model",
                      "obe_bayes_update_model_moments", "obe_bayes_update_model_moments_enqueue", "obe_update_one_pass",
                      "obe_strict_sums",
                      "obe_bayes_update_sweep",
                      "obe_eval_over_particles",
                      "obe_eval_over_settings", "obe_sweep_utility", "obe_sweep_kernel_time", "obe_sweep_timing",
                      "obe_eval_draws")


class HipLib:
    """Loaded library with typed entry points; ``call(name, *args)`` raises on error."""

    def __init__(self, path=LIB_PATH, plugin=False, allow_variant=False):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: the HIP library has not been built.  Run "
                "`python -m optbayesexpt_amd.build` (needs hipcc).  optbayesexpt_amd has "
                "no CPU fallback.")
        import torch  # noqa: F401  (loads the HIP runtime the library will bind to)
        self.path = path
        self.cdll = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL if plugin else ctypes.RTLD_GLOBAL)
        names = MODEL_ENTRY_POINTS + ("obe_abi_version", "obe_last_error", "obe_source_fingerprint") if plugin \
            else tuple(_SIGNATURES)
        for name in names:
            restype, argtypes = _SIGNATURES[name]
            fn = getattr(self.cdll, name)
            fn.restype = restype
            fn.argtypes = argtypes
        abi = self.cdll.obe_abi_version()
        if abi != OBE_ABI_VERSION:
            raise ImportError(f"libobe_hip ABI {abi} does not match this package ({OBE_ABI_VERSION})")
        from . import build
        built_from = self.cdll.obe_source_fingerprint().decode()
        #: extra compiler flags of a tools/build_variant.py library ("" for the product build); such a
        #: library is only accepted when the caller asks for it by path with allow_variant=True
        self.variant_flags = built_from.partition("+")[2]
        if allow_variant:
            built_from = built_from.partition("+")[0]
        if built_from != build._source_fingerprint():
            raise ImportError(f"{path} was built from other kernel sources ({built_from}) than the ones next to it "
                              f"({build._source_fingerprint()}): run `python -m optbayesexpt_amd.build`")

    def last_error(self):
        msg = self.cdll.obe_last_error()
        return msg.decode() if msg else ""

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise ObeHipError(f"{name} failed (status {rc}): {self.last_error()}", rc)
        if _AUDIT_ON:
            audit.after_call(name, args)          # OBE_CHECK_DELIVERY=1 (_audit.py): which host words are armed now
        return rc

    def workspace_bytes(self, n_particles, n_settings, n_channels, n_dims):
        return int(self.cdll.obe_workspace_bytes(n_particles, n_settings, n_channels, n_dims))

    def moments_len(self, n_dims):
        return int(self.cdll.obe_moments_len(n_dims))

    def device_info(self):
        buf = ctypes.create_string_buffer(256)
        ncu, mem = c_int(0), c_int64(0)
        rc = self.cdll.obe_device_info(buf, 256, ctypes.byref(ncu), ctypes.byref(mem))
        return dict(name=buf.value.decode(), n_cu=ncu.value, hbm_bytes=mem.value, is_gfx950=(rc == 0))


class DeviceBound:
    """A HipLib (or plugin) bound to one GPU.  The C ABI launches on the calling thread's *current*
    HIP device with the stream and pointers it is given; an object created with ``device='cuda:1'``
    while device 0 is current would otherwise launch on the wrong GPU.  Every call through this
    wrapper runs with the object's device current (a no-op test when it already is)."""

    def __init__(self, lib, device):
        import torch
        self._lib = lib._lib if isinstance(lib, DeviceBound) else lib
        self._torch = torch
        self.index = device.index if device.index is not None else torch.cuda.current_device()
        # one visible GPU: the current device cannot be another one
        self._single = torch.cuda.device_count() == 1 and self.index == 0

    def guard(self):
        if self._torch.cuda.current_device() == self.index:
            return _NO_GUARD
        return self._torch.cuda.device(self.index)

    def call(self, name, *args):
        if self._single:
            return self._lib.call(name, *args)
        with self.guard():
            return self._lib.call(name, *args)

    def device_info(self):
        with self.guard():
            return self._lib.device_info()

    def __getattr__(self, name):          # cdll, path, workspace_bytes, moments_len, last_error
        return getattr(self._lib, name)


import contextlib                         # noqa: E402
from ._audit import audit                 # noqa: E402
_AUDIT_ON = audit.on
_NO_GUARD = contextlib.nullcontext()

_LIB = None
_PLUGINS = {}


def load_plugin(path):
    """A per-model plugin library (optbayesexpt_amd.build.build_plugin), loaded once."""
    if path not in _PLUGINS:
        _PLUGINS[path] = HipLib(path, plugin=True)
    return _PLUGINS[path]


def load():
    """The process-wide library instance (loaded on first use).  If the in-tree library has
    not been built yet and hipcc is present it is built now; otherwise HipLib raises —
    there is no CPU fallback."""
    global _LIB
    if _LIB is None:
        from . import build
        if os.path.exists(build.HIPCC) and build.library_is_stale():
            # before the first dlopen (a loaded library cannot be replaced); serialised across the
            # ranks of a multi-GPU job by build()'s file lock, silent on stdout
            build.build(only_if_stale=True)
        _LIB = HipLib()
    return _LIB


def host_ptr(arr):
    """Address of a C-contiguous NumPy array (kept alive by the caller)."""
    return c_void_p(arr.ctypes.data)          # (data_as() costs twice as much: 2.2 vs 1.2 us per call)


class HostArgs:
    """Small host arrays that an object passes to the library call after call (record values,
    settings, result scalars), with their addresses made once: ``arr.ctypes`` builds a helper
    object every time it is touched (~1 us), and a cycle passes five such arrays."""

    def __init__(self):
        self._known = {}

    def keep(self, arr):
        self._known[id(arr)] = (arr, host_ptr(arr))          # (the array is kept alive with its address)
        return arr

    def ptr(self, arr):
        hit = self._known.get(id(arr))
        return hit[1] if hit is not None else host_ptr(arr)

    def ptr_keep(self, arr):
        """ptr() that remembers a long-lived array the first time it sees it."""
        hit = self._known.get(id(arr))
        if hit is None:
            self.keep(arr)
            hit = self._known[id(arr)]
        return hit[1]


# Page-locked landing zones outlive the objects that own them until the device has drained.  Kernels write their
# results straight into such memory, and some of them are deliberately not waited for (the sweep enqueued behind an
# update, the sum(w) of a small draw, the moments behind a constraint mask): when their object is garbage-collected
# with such a kernel still in flight, torch's caching host allocator — which knows nothing of kernels that write host
# memory themselves — would hand the block to the NEXT pin_memory() call at once, i.e. the next object's landing
# zone, and the late kernel would write into it (found by tools/soak_ranks.py once it mixed object classes: a
# spurious "Probabilities do not sum to 1").  So a released block goes to a limbo list instead, with the device
# whose kernels write it, and the list is emptied only behind a synchronisation of exactly those devices.
# (Device memory needs no such care: torch's device allocator reuses a block on the stream it was used on, and every
# kernel of this package runs on that stream or on a side stream that is joined back before the call returns.)
_LIMBO = []                # (keeper, device index)
_LIMBO_MAX = 64
_NO_LIMBO = os.environ.get("OBE_NO_LIMBO") == "1"


def _current_device_index():
    import torch
    try:
        return torch.cuda.current_device() if torch.cuda.is_available() else None
    except Exception:
        return None


def _retire_pinned(keeper, device):
    audit.zone_released(keeper)
    if _NO_LIMBO:            # (test hook: the behaviour before round 5's fix — the block goes straight back to the allocator)
        audit.zone_freed(keeper, drained=False)
        return
    _LIMBO.append((keeper, device))


def _purge_limbo(force=False):
    """Empties the limbo list once it holds _LIMBO_MAX blocks: synchronises the devices those blocks belong to —
    and no others: a rank of a multi-GPU job sees every GPU of the node, and touching one it does not use would
    create a context there — and only then lets the allocator have them."""
    if len(_LIMBO) < _LIMBO_MAX and not force:
        return
    import torch
    try:
        for d in sorted({d for _, d in _LIMBO if d is not None}):
            torch.cuda.synchronize(d)
    except Exception:        # (interpreter shutdown, a device that is gone: keep the blocks)
        return
    for keeper, _ in _LIMBO:
        audit.zone_freed(keeper, drained=True)
    del _LIMBO[:]


def pinned_array(n, dtype=np.float64, device=None):
    """A zeroed page-locked host array (NumPy view of a pinned torch tensor, which it keeps alive).
    The kernels that end a call write their few result scalars straight into such memory; a pageable
    array works too, through a small device-to-host copy."""
    import torch
    import weakref
    _purge_limbo()
    t = torch.zeros(n, dtype=torch.from_numpy(np.zeros(0, dtype=dtype)).dtype).pin_memory()
    raw = t.numpy()                       # the array every view of the zone has as its base: it keeps the storage alive ...
    dev = _current_device_index() if device is None else device
    weakref.finalize(raw, _retire_pinned, t, dev)          # ... and when the last view has gone, the limbo list does
    a = audit.wrap(raw)
    audit.zone_created(t.data_ptr(), t.numel() * t.element_size(), t)
    return a


def device_ptr_of_pinned(lib, arr):
    """The address kernels use for a ``pinned_array`` (obe_host_device_pointer): what to pass where
    the C ABI takes a *device* pointer and the result should land in host memory directly."""
    d = c_void_p()
    lib.call("obe_host_device_pointer", host_ptr(arr), ctypes.byref(d))
    return c_void_p(d.value)


def f64(values, n=None):
    """Small host array of float64 for by-value style arguments."""
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a
