"""The C ABI used from plain C (examples/c_abi_cycle.c: no Python, no torch in the process):
compiled with gcc against include/obe_hip.h, run on the GPU, checked against the oracle."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from oracle import models as omodels

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_cycle_from_plain_c(hip, tmp_path):
    rocm = "/opt/rocm"
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc or the HIP runtime headers are not installed")
    libdir = os.path.join(ROOT, "optbayesexpt_amd", "lib")
    exe = str(tmp_path / "c_abi_cycle")
    subprocess.run(["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{ROOT}/include",
                    os.path.join(ROOT, "examples", "c_abi_cycle.c"), f"-L{libdir}", "-lobe_hip", f"-L{rocm}/lib",
                    "-lamdhip64", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm}/lib", "-o", exe], check=True)
    g = np.random.default_rng(31)
    ns, n = 1500, 40000
    d, y_meas, sigma = 0.1, 49200.0, 500.0
    settings = np.linspace(1.5, 4.5, ns)
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    w = g.exponential(1.0, n)
    w /= w.sum()
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(src, "wb") as f:
        f.write(struct.pack("<qq", ns, n))
        f.write(struct.pack("<ddd", d, y_meas, sigma))
        f.write(settings.tobytes() + np.ascontiguousarray(prior).tobytes() + w.tobytes())
    r = subprocess.run([exe, src, dst], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(dst, "rb").read()
    best_idx = struct.unpack_from("<q", raw, 0)[0]
    best, kappa, sum_t, sum_w2 = struct.unpack_from("<dddd", raw, 8)
    body = np.frombuffer(raw, dtype=np.float64, offset=40)
    utility, w_new, mean, std = body[:ns], body[ns:ns + n], body[ns + n:ns + n + 3], body[ns + n + 3:ns + n + 6]

    allsettings = oracle.flatten_settings((settings,))
    yvar = oracle.yvar_full_sweep(omodels.lorentzian, allsettings, prior, w, (d,))
    ref_u = oracle.utility_from_yvar(yvar, sigma ** 2, 1.0)
    assert best_idx == int(np.argmax(ref_u)) and best == utility[best_idx]
    assert_allclose(utility, ref_u, rtol=1e-10)
    y_model = omodels.lorentzian((settings[best_idx],), prior, (d,))
    lik = oracle.gauss_likelihood(np.asarray(y_model, dtype=np.float64).reshape(-1), y_meas, sigma)
    ref_w = oracle.normalized_product(w, lik)
    assert_allclose(w_new, ref_w, rtol=1e-10, atol=1e-13 * ref_w.max())
    assert_allclose(sum_t, np.sum(np.nan_to_num(w * lik)), rtol=1e-12)
    assert_allclose(1.0 / sum_w2, oracle.effective_particles(ref_w), rtol=1e-10)
    assert_allclose(mean, oracle.weighted_mean(prior, ref_w), rtol=1e-10)
    assert_allclose(std, oracle.weighted_std(prior, ref_w), rtol=1e-7)
    assert kappa >= 0.0


def test_enqueued_cycle_from_plain_c(hip, tmp_path):
    """examples/c_abi_pipelined.c: the update enqueued without waiting and the sweep behind it
    (OBE_SWEEP_SPECULATIVE), from C — the program itself compares with the synchronous calls bit for bit and
    checks that a sweep behind a resampling update does nothing."""
    rocm = "/opt/rocm"
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("gcc or the HIP runtime headers are not installed")
    libdir = os.path.join(ROOT, "optbayesexpt_amd", "lib")
    exe = str(tmp_path / "c_abi_pipelined")
    subprocess.run(["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", f"-I{ROOT}/include",
                    os.path.join(ROOT, "examples", "c_abi_pipelined.c"), f"-L{libdir}", "-lobe_hip", f"-L{rocm}/lib",
                    "-lamdhip64", "-lm", f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm}/lib", "-o", exe], check=True)
    r = subprocess.run([exe, "2500", "150000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical to the synchronous one" in r.stdout and "did nothing" in r.stdout
