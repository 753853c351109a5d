"""The two example scripts (the reference demos' measurement loops against this package)
run end to end on the GPU and recover the simulated parameters."""
import importlib.util
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("selection", ["optimal", "good"])
def test_find_peak_example(hip, selection):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        true, mean, std = load("find_peak").main(n_measure=150, n_samples=20000, selection=selection, seed=3, quiet=True)
    assert np.all(np.abs(mean - np.array(true)) < 5 * std + 1e-9)
    assert std[0] < 0.01                                  # the peak position is pinned down


def test_sweeper_example(hip):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        true, mean, std = load("sweeper").main(n_measure=1200, n_samples=20000, selection="good", seed=4, quiet=True)
    assert np.all(np.abs(mean - np.array(true)) < 6 * std + 1e-9)
    assert std[0] < 0.05


@pytest.mark.parametrize("auto", ["1", "0"])
def test_reference_style_script_runs_through_the_optbayesexpt_alias(hip, auto):
    """A script written for the reference (plain Python model function, ``from optbayesexpt import
    OptBayesExpt, MeasurementSimulator``) run unchanged with compat/ on the path: with
    OBE_AUTO_DEVICE_MODEL=1 its model is translated into the HIP kernels, without it it stays a
    host-callable model; either way the experiment finds the peak."""
    import subprocess
    import sys
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "compat")]),
               OBE_AUTO_DEVICE_MODEL=auto)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "reference_style_script.py"), "150", "20000"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"model on the device: {auto == '1'}" in r.stdout
