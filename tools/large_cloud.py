"""The hot path on a cloud far beyond the BASELINE sizes (developer aid / robustness check):

    python tools/large_cloud.py [log2_particles=26] [n_settings=4096]

N = 2**26 particles x D = 3 is 1.6 GB of parameters + 2.1 GB of packed draws; 2**28 is 6.4 + 8.6 GB
(the card has 288 GB).  One opt_setting (full sweep) + pdf_update + moments + resample, each checked
against NumPy on the host: utility at sampled settings (oracle two-pass variance), posterior
weights, N_eff, mean / covariance / std, the resample indices (np.cumsum + searchsorted) and the
moved particles.  Exercises 64-bit indexing in every kernel and prints the time of each piece."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import optbayesexpt_amd as obe  # noqa: E402
import oracle  # noqa: E402
from oracle import models as om  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = 1 << lg
g = np.random.default_rng(20240424)
t0 = time.perf_counter()
prior = np.empty((3, n))
prior[0] = g.uniform(2, 4, n)
prior[1] = g.uniform(-2000, -400, n)
prior[2] = g.normal(50000, 1000, n)
print(f"N = 2^{lg} = {n} particles, {ns} settings; prior generated in {time.perf_counter() - t0:.1f} s", flush=True)
sv = (np.linspace(1.5, 4.5, ns),)
cons = (0.1,)


def sync():
    torch.cuda.synchronize()


def timed(label, fn):
    sync()
    t = time.perf_counter()
    out = fn()
    sync()
    print(f"  {label:44s} {1e3 * (time.perf_counter() - t):10.2f} ms", flush=True)
    return out


o = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior, cons, scale=False, utility_method="variance_full",
                     default_noise_std=500.0)
o.rng = np.random.default_rng(7)
o._particles.tensor()
o._weights.tensor()
ok = True


def check(label, got, ref, rtol, atol=0.0):
    global ok
    got, ref = np.asarray(got), np.asarray(ref)
    err = np.max(np.abs(got - ref) / (atol + rtol * np.abs(ref))) if got.size else 0.0
    good = bool(err <= 1.0)
    ok &= good
    print(f"  check {label:38s} {'ok' if good else 'FAIL'}  (worst error / tolerance = {err:.3g})", flush=True)


# ---- cycle 1: uniform weights
x = timed("opt_setting (full sweep, uniform weights)", o.opt_setting)
timed("opt_setting again", o.opt_setting)
util = o._utility_dev.cpu().numpy()
sample = np.unique(np.r_[0, ns - 1, o.last_setting_index, g.integers(0, ns, 3)])
w0 = np.full(n, 1.0 / n)
ref = oracle.yvar_full_sweep(om.lorentzian, oracle.flatten_settings(sv)[:, sample], prior, w0, cons, chunk=1 << 16)
check("utility at sampled settings", util[sample], ref[0] / 500.0 ** 2, 1e-10)

y = float(om.lorentzian(x, (3.0, -1000.0, 50000.0), cons)) + 300.0
timed("pdf_update (first call: loads the kernels)", lambda: o.pdf_update((x, y, 500.0)))
lik = oracle.gauss_likelihood(om.lorentzian(x, prior, cons), y, 500.0)
w1 = oracle.normalized_product(w0, lik)
wd = o._weights.tensor().cpu().numpy()
check("posterior weights", wd, w1, 1e-10, 1e-13 * w1.max())
check("N_eff", o.last_n_eff, oracle.effective_particles(w1), 1e-10)
resampled_1 = bool(o.just_resampled)
print(f"  (resampled in cycle 1: {resampled_1}, N_eff / N = {o.last_n_eff / n:.3f})")

# ---- moments on non-uniform weights
wts = g.exponential(1.0, n)
wts /= wts.sum()
o.set_pdf(prior, wts)
o._particles.tensor()
o._weights.tensor()
wd = o._weights.tensor().cpu().numpy()
check("set_pdf weights", wd, wts, 1e-12)
mean = timed("mean()", o.mean)
cov = timed("covariance()", o.covariance)
std = timed("std()", o.std)
check("mean", mean, oracle.weighted_mean(prior, wd), 1e-11)
check("covariance", cov, oracle.weighted_covariance(prior, wd), 1e-9, 1e-12 * np.abs(cov).max())
check("std", std, oracle.weighted_std(prior, wd), 1e-7)

# ---- sweep with non-uniform weights, then a resample
timed("opt_setting (non-uniform weights)", o.opt_setting)
util = o._utility_dev.cpu().numpy()
ref = oracle.yvar_full_sweep(om.lorentzian, oracle.flatten_settings(sv)[:, sample], prior, wd, cons, chunk=1 << 16)
check("utility (non-uniform weights)", util[sample], ref[0] / 500.0 ** 2, 1e-10)

ref_rng = np.random.default_rng(7)
ref_rng.bit_generator.state = o.rng.bit_generator.state
timed("resample (device generator, pipelined)", o.resample)
idx = o.last_resample_indices_device.cpu().numpy()
u = ref_rng.random(n)
ref_idx = oracle.choice_indices(wd, u)
bad = int(np.sum(idx != ref_idx))
# np.cumsum adds serially: its CDF is off by ~eps * sqrt(N) by the end, the blocked device scan by
# ~eps * log N; a uniform that falls between the two CDFs picks the neighbouring particle.  The
# arbiter is the same CDF accumulated in extended precision.
cdf_x = np.cumsum(wd.astype(np.longdouble))
cdf_x /= cdf_x[-1]
exact_idx = cdf_x.astype(np.float64).searchsorted(u, side="right")
bad_x, ref_bad_x = int(np.sum(idx != exact_idx)), int(np.sum(ref_idx != exact_idx))
print(f"  resample indices: {bad} of {n} differ from float64 np.cumsum + searchsorted; against the CDF summed in "
      f"extended precision: device scan {bad_x}, np.cumsum {ref_bad_x}")
step = np.abs(idx - ref_idx)
ok &= bool(step.max() <= 1) and bad <= max(8, int(n * 1e-16 * np.sqrt(n) * n * 4)) and bad_x <= max(8, ref_bad_x)
z = ref_rng.standard_normal((n, 3))
del cdf_x, exact_idx, step
assert ref_rng.bit_generator.state == o.rng.bit_generator.state, "generator state after the resample"
a = o.tuning_parameters["a_param"]
f = oracle.nudge_factor((1 - a * a) * np.asarray(cov))
new_ref = prior[:, idx] + (z @ f.T).T
new = o._particles.tensor().cpu().numpy()
scale = np.sqrt(np.max(np.linalg.eigvalsh((1 - a * a) * np.asarray(cov))))
body = np.all(np.abs(z) <= 3.6541528853610088, axis=1)       # ziggurat tail draws may differ in the last bit
check("particles after the resample", new[:, body], new_ref[:, body], 1e-10, 512 * 2.3e-16 * scale)
check("weights after the resample", o._weights.tensor().cpu().numpy(), np.full(n, 1.0 / n), 1e-15)
# ---- steady state: a second update and a second resample (kernels loaded, buffers allocated)
x = o.opt_setting()
timed("pdf_update (steady state)", lambda: o.pdf_update((x, y, 500.0)))
o.particle_weights = wts
o._weights.tensor()
timed("resample (steady state)", o.resample)
print("LARGE CLOUD", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
