"""TEST INFRASTRUCTURE — build and bind oracle/csweep.c (plain C + OpenMP restatement of the
Lorentzian full sweep and update): a second oracle for the NumPy one and the all-cores CPU
baseline of bench.py.  Never imported by the product package."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csweep.c")
OUT = os.path.join(HERE, "_build", "libcsweep.so")


def build(verbose=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        cmd = ["gcc", "-O3", "-mavx2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", SRC, "-o", OUT, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"gcc failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print("built", OUT)
    return OUT


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        P, c_long, c_int, c_double = ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_double
        L.csweep_threads.restype = c_int
        L.csweep_lorentz_yvar.argtypes = [P, c_long, P, c_long, c_long, P, c_int, c_double, P]
        L.csweep_lorentz_yvar.restype = None
        L.csweep_lorentz_update.argtypes = [c_double, c_double, c_double, P, c_long, c_long, P, c_int, c_double, P]
        L.csweep_lorentz_update.restype = c_double
        _LIB = L
    return _LIB


def threads():
    return int(lib().csweep_threads())


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def lorentz_yvar(settings, particles, weights, d, n_peaks=1):
    """(N_s,) weighted variance of the K-peak Lorentzian over the particle cloud."""
    x = np.ascontiguousarray(settings, dtype=np.float64)
    par = np.ascontiguousarray(particles, dtype=np.float64)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    out = np.empty(x.size)
    lib().csweep_lorentz_yvar(_p(x), x.size, _p(par), par.shape[1], par.shape[1], _p(w), n_peaks, float(d), _p(out))
    return out


def lorentz_update(x, y_meas, sigma, particles, weights, d, n_peaks=1):
    """Normalised posterior weights and sum(w'^2) after one measurement with known sigma."""
    par = np.ascontiguousarray(particles, dtype=np.float64)
    w = np.array(weights, dtype=np.float64)
    s2 = ctypes.c_double(0.0)
    lib().csweep_lorentz_update(float(x), float(y_meas), float(sigma), _p(par), par.shape[1], par.shape[1], _p(w),
                                n_peaks, float(d), ctypes.byref(s2))
    return w, s2.value
