"""OBE_CHECK_DELIVERY=1 (optbayesexpt_amd/_audit.py): the watched-word delivery of results to the host, audited by
construction instead of by soak runs.

(i)  The GPU suite — units, trajectories, sweeper, the fuzz / soak slices of test_gpu_speculative.py, the two-rank
     slices — runs once more in a child pytest with the audit on: no read of a word that was armed and not waited
     for, no landing zone released with armed words.
(ii) The three races of rounds 4-5 as DETERMINISTIC failures of the mode.  Each script below issues the call pattern of
     the code BEFORE its fix and the audit must refuse it (the fixed pattern is what (i) runs):
       * round 4, fixed in ba933f8 (the code before it: ba933f8^ = 394e0ce): a result block that spans several 128-byte lines was
         waited for by watching ONE flag word stored behind a fence; about once in 3000 resamples the host read
         covariance entries of another line that had not arrived;
       * round 5, fixed in 6bc2b8f (the code before it: 6bc2b8f^): good_setting()'s sum(p) and a small draw's sum(w)
         were read once ANOTHER word had arrived ("Probabilities do not sum to 1" on one rank, then a hung peer);
       * round 5, fixed in 3e013a4 (the code before it: 3e013a4^): a landing zone went back to the allocator while a kernel of its dead
         owner (the sweep enqueued behind an update) had not delivered yet, and the next object got the block.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OBE_CHECK_DELIVERY="1", PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"), **extra)
    return env


def _script(body, **extra):
    pre = ("import ctypes, gc, warnings\nimport numpy as np\nimport torch\nimport optbayesexpt_amd as obe\n"
           "from optbayesexpt_amd import _lib\nfrom optbayesexpt_amd._audit import DeliveryError, audit\n"
           "from optbayesexpt_amd.particlepdf import _ptr\nassert audit.on\n")
    r = subprocess.run([sys.executable, "-c", pre + body], env=_env(**extra), capture_output=True, text=True, timeout=600)
    return r


@pytest.mark.parametrize("files", [
    ["tests/test_gpu_units.py"],
    ["tests/test_gpu_speculative.py", "tests/test_gpu_sweeper.py"],
    ["tests/test_gpu_trajectories.py", "tests/test_gpu_c_abi.py", "tests/test_gpu_examples.py"],
    ["tests/test_gpu_two_ranks.py"],
    ["tests/test_gpu_scale.py", "-k", "not whole_grid and not large_cloud"],
])
def test_gpu_suite_under_the_delivery_audit(hip, files, tmp_path):
    report = tmp_path / "audit.jsonl"
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + files,
                       cwd=ROOT, env=_env(OBE_AUDIT_REPORT=str(report)), capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "DeliveryError" not in r.stdout
    rows = [json.loads(l) for l in report.read_text().splitlines()]
    total = {k: sum(row[k] for row in rows) for k in ("zones", "armed", "waited", "reads")}
    assert not any(row["pending_violations"] for row in rows), rows
    # the mode was really on, in every process of the run, and saw the protocol at work
    assert total["zones"] > 10 and total["armed"] > 20 and total["waited"] > 20 and total["reads"] > 100, total
    print(f"{files}: {len(rows)} audited process(es): {total}")


def test_round5_unwatched_sum_of_p_is_refused(hip):
    """6bc2b8f^: obe_draw_indices delivers the chosen index AND sum(p); only the index was armed and waited for."""
    r = _script('''
g = np.random.default_rng(0)
o = obe.ParticlePDF(g.normal(size=(2, 3000)))
lib, st = o._lib, o._stream()
w = o._weights.tensor()
cdf = torch.empty(o.n_particles, dtype=torch.float64, device=o._device)
idx_host = _lib.pinned_array(1, np.int64)
idx_dev = _lib.device_ptr_of_pinned(lib, idx_host)
total = o._total_pinned
u = np.array([0.3])
lib.call("obe_host_word_arm", _lib.host_ptr(idx_host))
lib.call("obe_draw_indices", _ptr(w), o.n_particles, 0, 0, _ptr(cdf), _lib.host_ptr(u), 1, idx_dev, o._total_ptrs[1],
         _ptr(o._ws), o._ws_bytes, st)
lib.call("obe_host_word_wait", _lib.host_ptr(idx_host), st)
assert 0 <= int(idx_host[0]) < o.n_particles          # the watched word: fine
try:
    float(total[1])
    print("NOT CAUGHT")
except DeliveryError as exc:
    print("CAUGHT:", exc)
lib.call("obe_host_word_wait", o._total_ptrs[1], st)  # the fix: wait for that word too
assert abs(float(total[1]) - 1.0) < 1e-12
print("FIXED PATTERN OK")
''')
    assert r.returncode == 0 and "CAUGHT:" in r.stdout and "FIXED PATTERN OK" in r.stdout and "NOT CAUGHT" not in r.stdout, \
        r.stdout[-2000:] + r.stderr[-2000:]


def test_round5_unwatched_sum_of_w_of_a_small_draw_is_refused(hip):
    """6bc2b8f^: the sum(w) of a small draw was read after the NEXT call's own words had arrived (a later kernel's
    stores do not prove an earlier kernel's)."""
    r = _script('''
g = np.random.default_rng(1)
o = obe.ParticlePDF(g.normal(size=(2, 3000)))
lib, st = o._lib, o._stream()
w = o._weights.tensor()
cdf = torch.empty(o.n_particles, dtype=torch.float64, device=o._device)
idx = torch.empty(5, dtype=torch.int64, device=o._device)
u = g.random(5)
lib.call("obe_draw_indices", _ptr(w), o.n_particles, 0, 0, _ptr(cdf), _lib.host_ptr(u), 5, _ptr(idx), o._total_ptrs[0],
         _ptr(o._ws), o._ws_bytes, st)
lib.call("obe_weight_sums", _ptr(w), o.n_particles, _ptr(o._ws), o._ws_bytes, _lib.host_ptr(o._host_out), st)
assert abs(float(o._host_out[1]) - 1.0) < 1e-12       # the later call's own words: waited for by that call
try:
    float(o._total_pinned[0])
    print("NOT CAUGHT")
except DeliveryError as exc:
    print("CAUGHT:", exc)
''')
    assert r.returncode == 0 and "CAUGHT:" in r.stdout and "NOT CAUGHT" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_round4_block_watched_by_one_flag_word_is_refused(hip):
    """ba933f8^: the covariance block of obe_resample_begin (3 + 4 D + D^2 words, several 128-byte lines) was
    waited for through its last word only."""
    r = _script('''
from optbayesexpt_amd import _devrng
g = np.random.default_rng(2)
n, d = 70000, 3
o = obe.ParticlePDF(g.normal(size=(d, n)))
o.rng = np.random.default_rng(3)
lib, st = o._lib, o._stream()
o.bayesian_update(np.exp(-g.random(n)))
state, h_state = _devrng.pcg64_state(o.rng)
b = o._resample_buffers(n, d)
p, w = o._pw_tensors()
mlen = lib.moments_len(d)
lib.call("obe_resample_begin", _ptr(p), p.shape[1], d, n, _ptr(w), _lib.host_ptr(h_state), 0, 0, 0, b["n_raw"],
         _ptr(o._cdf_dev), _ptr(b["uni"]), _ptr(b["idx"][0]), _ptr(b["tables"]), _ptr(b["normals"]), _ptr(b["zig_ws"]),
         b["zig_ws"].numel() * 8, _ptr(o._moments_dev), b["p_f"], b["p_i"], None if b["aos"] is None else _ptr(b["aos"]),
         _ptr(o._ws), o._ws_bytes, st)
pin_f = b["pin_f"]
lib.call("obe_host_word_wait", ctypes.c_void_p(pin_f.ctypes.data + 8 * mlen), st)      # the "flag": the last word
float(pin_f[mlen])                                                                     # fine
try:
    pin_f[1 + 2 + 4 * d:1 + mlen].copy()                                               # the covariance: never waited for
    print("NOT CAUGHT")
except DeliveryError as exc:
    print("CAUGHT:", exc)
lib.call("obe_host_words_wait", ctypes.c_void_p(pin_f.ctypes.data + 8), mlen, st)      # the fix: every word
lib.call("obe_host_words_wait", b["p_f"], 1, st)
lib.call("obe_host_words_wait", b["p_i"], 2, st)
cov = pin_f[1 + 2 + 4 * d:1 + mlen].reshape(d, d)
assert np.allclose(cov, cov.T) and np.all(np.diag(cov) > 0)
print("FIXED PATTERN OK")
''')
    assert r.returncode == 0 and "CAUGHT:" in r.stdout and "FIXED PATTERN OK" in r.stdout and "NOT CAUGHT" not in r.stdout, \
        r.stdout[-2000:] + r.stderr[-2000:]


_DEAD_OWNER = '''
import bench
settings, prior, cons, true, sigma = bench.make_workload("c2")
sv = (np.ascontiguousarray(settings[0][::8]),)
def experiment(seed):
    o = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior[:, :65536].copy(), cons, scale=False,
                         utility_method="variance_full", default_noise_std=sigma)
    o.tuning_parameters["speculative_sweep"] = True
    o.tuning_parameters["auto_resample"] = False
    o.rng = np.random.default_rng(seed)
    for _ in range(3):
        x = o.opt_setting()
        o.pdf_update((x, 49500.0, sigma))         # ... enqueues the next sweep behind the update: armed words, not waited for
    assert o.sweep_state()["pending"] != "none", o.sweep_state()
    return None                                   # the object dies with that sweep's result undelivered
try:
    for seed in range(4):
        experiment(seed)
        gc.collect()
    print("NO VIOLATION")
except DeliveryError as exc:
    print("CAUGHT:", exc)
'''


def test_round5_landing_zone_freed_under_a_dead_owners_kernel_is_refused(hip):
    """3e013a4^: with the limbo list switched off (OBE_NO_LIMBO=1, the behaviour before the fix) the landing zone of
    an object that dies with a sweep enqueued ahead goes straight back to the allocator: the audit reports it at the
    next library call.  With the list (the product) the same script is clean."""
    r = _script(_DEAD_OWNER, OBE_NO_LIMBO="1")
    assert r.returncode == 0 and "CAUGHT:" in r.stdout and "armed word" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    r = _script(_DEAD_OWNER)
    assert r.returncode == 0 and "NO VIOLATION" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
