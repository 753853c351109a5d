"""The sweep that pdf_update() enqueues behind its update without waiting (obe_base.py: _speculation_wanted)
changes WHEN the kernels run, never what they compute: every cycle of a run with it must equal, bit for bit,
the cycle of a run without it — whatever the caller does between the update and the next sweep (GPU)."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def lorentz(x, x0, a, b, d):
    return b + a / (((x - x0) / d) ** 2 + 1.0)


def make(obe, mode, n_particles=20000, n_settings=3000, noise_param=False, shard=None, seed=3, threshold=0.5):
    rng = np.random.default_rng(seed)
    x = np.linspace(1.5, 4.5, n_settings)
    prior = [rng.uniform(2.0, 4.0, n_particles), rng.uniform(1.0, 3.0, n_particles), rng.normal(0.5, 0.3, n_particles)]
    if noise_param:
        prior.append(rng.uniform(0.05, 0.6, n_particles))
        o = obe.OptBayesExptNoiseParameter(obe.models.lorentzian(1), (x,), np.array(prior), (0.1,),
                                           noise_parameter_index=3, utility_method="variance_full",
                                           resample_threshold=threshold, settings_shard=shard)
    else:
        o = obe.OptBayesExpt(obe.models.lorentzian(1), (x,), np.array(prior), (0.1,), utility_method="variance_full",
                             resample_threshold=threshold, settings_shard=shard)
    o.tuning_parameters["speculative_sweep"] = mode
    o.rng = np.random.default_rng(11)
    return o


def snapshot(o):
    return dict(w=o.particle_weights.copy(), p=o.particles.copy(), resampled=bool(o.just_resampled),
                rng=o.rng.bit_generator.state["state"]["state"], mean=o.mean().copy())


def cycles(o, n, between=None, sigma=0.3, seed=5):
    meas = np.random.default_rng(seed)
    log = []
    for c in range(n):
        s = o.opt_setting()
        swept = dict(u=o._utility_dev.cpu().numpy().copy(), yv=o._yvar_dev.cpu().numpy().copy(),
                     sweep=dict(o.last_sweep))
        y = lorentz(s[0], 3.1, 2.2, 0.4, 0.1) + sigma * meas.standard_normal()
        o.pdf_update((s, y, sigma))
        entry = dict(snapshot(o), **swept)
        entry["setting"] = tuple(float(v) for v in s)
        if between is not None:
            entry["between"] = between(o, c)
        log.append(entry)
    return log


def same(a, b):
    assert len(a) == len(b)
    for c, (x, y) in enumerate(zip(a, b)):
        assert x["setting"] == y["setting"], c
        assert x["resampled"] == y["resampled"], c
        assert x["rng"] == y["rng"], c
        kx, ky = x["sweep"].pop("kappa"), y["sweep"].pop("kappa")
        assert x["sweep"] == y["sweep"], c
        assert kx == ky or (np.isnan(kx) and np.isnan(ky)), c
        for k in ("w", "p", "u", "yv", "mean"):
            assert np.array_equal(x[k], y[k], equal_nan=True), (c, k)
        if "between" in x:
            assert np.array_equal(np.asarray(x["between"]), np.asarray(y["between"]), equal_nan=True), c


def counted(o):
    """Counts the speculative sweeps that were used / launched / aborted by the update's resample."""
    n = dict(taken=0, launched=0, aborted=0, after_resample=0)
    take, sweep = o._take_speculative_sweep, o._sweep_device

    def counting_take(shifted):
        if o.sweep_state()["pending"] == "aborted":
            n["aborted"] += 1
        got = take(shifted)
        n["taken"] += got is not None
        return got

    def counting_sweep(want_best, speculate=False):
        n["launched"] += bool(speculate)
        n["after_resample"] += speculate == "after_resample"
        return sweep(want_best, speculate=speculate)

    o._take_speculative_sweep, o._sweep_device = counting_take, counting_sweep
    return n


@pytest.mark.parametrize("noise_param", [False, True])
def test_cycles_with_and_without_the_speculative_sweep_are_the_same_cycles(hip, noise_param):
    import optbayesexpt_amd as obe
    plain = cycles(make(obe, False, noise_param=noise_param), 40)
    assert 3 <= sum(e["resampled"] for e in plain) <= 30
    for mode in (True, "auto"):
        o = make(obe, mode, noise_param=noise_param)
        n = counted(o)
        same(cycles(o, 40), [dict(e, sweep=dict(e["sweep"])) for e in plain])
        resamples = sum(e["resampled"] for e in plain)
        # True: every sweep but the first was enqueued by the update before it — behind the update itself, or,
        # when that update resampled (and the sweep behind it did nothing), behind the resample.
        # 'auto' waits for two plain cycles, and does not guess while more than half of the recent updates resample
        if mode is True:
            assert n["taken"] == 39 and n["after_resample"] == resamples and n["launched"] == 40 + resamples
        else:
            assert n["taken"] >= 30 and n["after_resample"] >= resamples - 3


def test_whatever_happens_between_the_update_and_the_sweep(hip):
    """mean()/covariance() (workspace reuse), an extra opt_setting(), utility(), an extra update, new weights,
    a changed noise level, a cost hook that appears: each leaves the cycles equal to the plain ones."""
    import optbayesexpt_amd as obe

    def between(o, c):
        k = c % 8
        if k == 0:
            return np.concatenate([o.covariance().ravel(), o.std()])
        if k == 1:
            return np.asarray(o.opt_setting())              # the speculative result is this sweep's; the cycle's is fresh
        if k == 2:
            return o.utility().copy()
        if k == 3:
            o.pdf_update(((2.9,), 1.7, 0.3))               # a second update: the first one's sweep is never asked for
            return o.mean()
        if k == 4:
            w = o.particle_weights.copy()
            w[::2] *= 0.5
            o.particle_weights = w / w.sum()
            return o.mean()
        if k == 5:
            o.default_noise_std = o.default_noise_std * 1.5
            return o.default_noise_std.ravel()
        if k == 6 and c == 22:
            o.cost_estimate = lambda: 2.0 + np.cos(o.allsettings[0])
            return np.zeros(1)
        return np.asarray(o.good_setting(pickiness=9))      # draws from self.rng on the utility of the sweep

    plain = cycles(make(obe, False), 32, between)
    o = make(obe, True)
    n = counted(o)
    same(cycles(o, 32, between), plain)
    assert n["taken"] >= 8 and n["launched"] >= 20


def test_one_rank_of_a_sharded_object_speculates_too(hip):
    """A shard keeps no host words: the record of the speculative sweep is copied on the device before the
    workspace is reused and gathered when the sweep is asked for."""
    import optbayesexpt_amd as obe
    from optbayesexpt_amd.dist import SettingsShard

    def between(o, c):
        return o.covariance().ravel() if c % 2 else o.std()

    plain = cycles(make(obe, False), 24, between)
    o = make(obe, True, shard=SettingsShard(0, 1))
    n = counted(o)
    same(cycles(o, 24, between), plain)
    assert n["taken"] >= 12


def test_resample_decision_of_the_device_is_the_hosts(hip):
    """The flag the update kernel leaves for the sweep and the host's resample_test() agree cycle by cycle
    (thresholds that make almost every / almost no update resample), and an overridden resample_test() that
    disagrees with the device costs a sweep, never a wrong one."""
    import optbayesexpt_amd as obe
    for thr in (0.98, 0.02):
        plain = cycles(make(obe, False, threshold=thr, n_particles=6000, n_settings=700), 25)
        o = make(obe, True, threshold=thr, n_particles=6000, n_settings=700)
        flags = []
        test = o.resample_test

        def checking():
            flags.append(float(o._upd_host[4 + 4 * o.n_dims]))
            test()
            assert flags[-1] == float(o.just_resampled)

        o.resample_test = checking
        same(cycles(o, 25), plain)
        assert len(flags) == 25 and (np.mean(flags) > 0.7 if thr > 0.5 else np.mean(flags) < 0.3)

    class Stubborn(obe.OptBayesExpt):
        def resample_test(self):                   # resamples every third update, whatever N_eff is
            self._n = getattr(self, "_n", 0) + 1
            if self._n % 3 == 0:
                self.resample()
                self.just_resampled = True
            else:
                self.just_resampled = False

    def stubborn(mode):
        rng = np.random.default_rng(3)
        x = np.linspace(1.5, 4.5, 900)
        prior = np.array([rng.uniform(2.0, 4.0, 8000), rng.uniform(1.0, 3.0, 8000), rng.normal(0.5, 0.3, 8000)])
        o = Stubborn(obe.models.lorentzian(1), (x,), prior, (0.1,), utility_method="variance_full")
        o.tuning_parameters["speculative_sweep"] = mode
        o.rng = np.random.default_rng(11)
        return o

    same(cycles(stubborn(True), 20), cycles(stubborn(False), 20))


def test_enqueue_form_and_sweep_flags_through_the_abi(hip):
    """obe_bayes_update_model_moments_enqueue + obe_sweep_utility(OBE_SWEEP_SPECULATIVE / OBE_SWEEP_NOWAIT) called
    the way include/obe_hip.h describes them, against the synchronous calls: same host block, same weights, same
    sweep result; a sweep behind an update that resamples leaves its words armed and its outputs untouched; the
    refusals (draws mode, pageable result words) are errors, not silent synchronous calls."""
    import torch
    import optbayesexpt_amd as obe
    from optbayesexpt_amd import _lib
    from optbayesexpt_amd.particlepdf import _ptr, _P
    g = np.random.default_rng(8)
    n, ns, d = 30000, 2500, 3
    prior = np.array([g.uniform(2, 4, n), g.uniform(1, 3, n), g.normal(0.5, 0.3, n)])
    w0 = g.exponential(1.0, n)
    w0 /= w0.sum()
    o = obe.OptBayesExpt(obe.models.lorentzian(1), (np.linspace(1.5, 4.5, ns),), prior, (0.1,),
                         utility_method="variance_full")
    lib, st = o._mlib, o._stream()
    par = o._particles.tensor()
    w = o._weights.tensor()
    mom = torch.zeros_like(o._moments_dev)
    noise = torch.full((1,), 0.09, dtype=torch.float64, device=w.device)
    yvar, util = torch.zeros((1, ns), dtype=torch.float64, device=w.device), torch.zeros(ns, dtype=torch.float64, device=w.device)
    hp = _lib.host_ptr
    # (a mild measurement: N_eff stays above 0.1 N, the other clause of particlepdf.py:236-258's test)
    setting, y_meas, sigma = np.array([3.05, 0, 0, 0]), np.array([1.9, 0, 0, 0]), np.array([2.0, 1, 1, 1])
    upd = (o._model_struct, _ptr(par), par.shape[1], n, _ptr(w), hp(setting), hp(y_meas), hp(sigma), None, 1, float("nan"))

    def sweep(flags, best, idx, kappa):
        return lib.cdll.obe_sweep_utility(o._model_struct, _P(o._settings_dev.data_ptr()), ns, ns, _ptr(par), par.shape[1], n,
                                          _ptr(w), None, 0, _ptr(mom), _lib.OBE_SWEEP_SHIFTED | flags, _ptr(noise), 0, None, 1.0,
                                          _ptr(yvar), _ptr(util), best, idx, kappa, _ptr(o._ws), o._ws_bytes, st)

    def reset():
        w.copy_(torch.from_numpy(w0))
        util.fill_(-1.0)
        torch.cuda.synchronize()

    # the synchronous pair
    reset()
    out = _lib.pinned_array(5 + 4 * d)
    lib.call("obe_bayes_update_model_moments", *upd, _ptr(mom), _ptr(o._ws), o._ws_bytes, hp(out), st)
    res = _lib.pinned_array(4)
    best, idx, kappa = res[0:1], res.view(np.int64)[1:2], res[2:3]
    assert sweep(0, hp(best), hp(idx), hp(kappa)) == 0
    torch.cuda.synchronize()
    ref = dict(out=out[:4 + 4 * d].copy(), w=w.cpu().numpy(), res=res[:3].copy(), util=util.cpu().numpy(), mom=mom.cpu().numpy())

    # enqueued update + speculative sweep, no resample (threshold 0: N_eff / N is never below it)
    reset()
    out2, res2 = _lib.pinned_array(5 + 4 * d), _lib.pinned_array(4)
    b2, i2, k2 = res2[0:1], res2.view(np.int64)[1:2], res2[2:3]
    lib.call("obe_bayes_update_model_moments_enqueue", *upd, _ptr(mom), _ptr(o._ws), o._ws_bytes, hp(out2), 1, 0.0, st)
    assert sweep(_lib.OBE_SWEEP_SPECULATIVE, hp(b2), hp(i2), hp(k2)) == 0
    lib.call("obe_host_words_wait", hp(out2), 5 + 4 * d, st)
    assert 1.0 / ref["out"][1] > 0.1 * n
    assert np.array_equal(out2[:4 + 4 * d], ref["out"]) and out2[4 + 4 * d] == 0.0
    lib.call("obe_host_words_wait", hp(res2), 3, st)
    assert np.array_equal(res2[:3].view(np.uint64), ref["res"].view(np.uint64))
    torch.cuda.synchronize()
    assert np.array_equal(w.cpu().numpy(), ref["w"]) and np.array_equal(util.cpu().numpy(), ref["util"])
    assert np.array_equal(mom.cpu().numpy()[:2 + 4 * d], ref["mom"][:2 + 4 * d])

    # ... and behind an update that resamples (threshold 1.1: N_eff / N is always below it): nothing runs
    reset()
    lib.call("obe_bayes_update_model_moments_enqueue", *upd, _ptr(mom), _ptr(o._ws), o._ws_bytes, hp(out2), 1, 1.1, st)
    assert sweep(_lib.OBE_SWEEP_SPECULATIVE, hp(b2), hp(i2), hp(k2)) == 0
    lib.call("obe_host_words_wait", hp(out2), 5 + 4 * d, st)
    assert out2[4 + 4 * d] == 1.0 and np.array_equal(out2[:4 + 4 * d], ref["out"])
    torch.cuda.synchronize()
    assert np.all(res2[:3].view(np.uint64) == _lib.HOST_SENTINEL)          # still armed: the sweep did nothing
    assert np.all(util.cpu().numpy() == -1.0) and np.array_equal(w.cpu().numpy(), ref["w"])
    # auto_resample = 0: the same sums never say "resample"
    reset()
    lib.call("obe_bayes_update_model_moments_enqueue", *upd, _ptr(mom), _ptr(o._ws), o._ws_bytes, hp(out2), 0, 1.1, st)
    lib.call("obe_host_words_wait", hp(out2), 5 + 4 * d, st)
    assert out2[4 + 4 * d] == 0.0

    # OBE_SWEEP_NOWAIT does not look at the decision (the word still says what the last enqueued update left)
    lib.call("obe_bayes_update_model_moments_enqueue", *upd, _ptr(mom), _ptr(o._ws), o._ws_bytes, hp(out2), 1, 1.1, st)
    lib.call("obe_host_words_wait", hp(out2), 5 + 4 * d, st)
    reset()
    lib.call("obe_bayes_update_model_moments", *upd, _ptr(mom), _ptr(o._ws), o._ws_bytes, hp(out), st)
    assert sweep(_lib.OBE_SWEEP_NOWAIT, hp(b2), hp(i2), hp(k2)) == 0
    lib.call("obe_host_words_wait", hp(res2), 3, st)
    torch.cuda.synchronize()
    assert np.array_equal(res2[:3].view(np.uint64), ref["res"].view(np.uint64))
    assert np.array_equal(util.cpu().numpy(), ref["util"])

    # refusals
    pageable = np.zeros(1)
    assert sweep(_lib.OBE_SWEEP_SPECULATIVE, hp(pageable), hp(i2), hp(k2)) != 0
    assert "page-locked" in lib.last_error()
    draws = torch.zeros(16, dtype=torch.int64, device=w.device)
    rc = lib.cdll.obe_sweep_utility(o._model_struct, _P(o._settings_dev.data_ptr()), ns, ns, _ptr(par), par.shape[1], n, _ptr(w),
                                    _ptr(draws), 16, _ptr(mom), _lib.OBE_SWEEP_SHIFTED | _lib.OBE_SWEEP_NOWAIT, _ptr(noise), 0,
                                    None, 1.0, _ptr(yvar), _ptr(util), hp(b2), hp(i2), hp(k2), _ptr(o._ws), o._ws_bytes, st)
    assert rc != 0 and "full sweeps" in lib.last_error()
    with pytest.raises(_lib.ObeHipError, match="page-locked"):
        lib.call("obe_bayes_update_model_moments_enqueue", *upd, _ptr(mom), _ptr(o._ws), o._ws_bytes, hp(np.zeros(5 + 4 * d)),
                 1, 0.5, st)


@pytest.mark.parametrize("sharded", [False, True])
def test_cycles_on_alternating_streams(hip, sharded):
    """A caller that moves to another torch stream between pdf_update() and opt_setting() (ordering the streams
    the torch way): the sweep enqueued on the stream it left is waited for there, or — a shard, whose record
    lives on the device — not used."""
    import torch
    import optbayesexpt_amd as obe
    from optbayesexpt_amd.dist import SettingsShard

    def run(mode):
        o = make(obe, mode, n_particles=30000, n_settings=2000, shard=SettingsShard(0, 1) if sharded else None)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        meas = np.random.default_rng(5)
        log = []
        torch.cuda.synchronize()
        for c in range(16):
            cur, prev = streams[c % 2], streams[(c + 1) % 2]
            cur.wait_stream(prev)
            with torch.cuda.stream(cur):
                s = o.opt_setting()
                u = o._utility_dev.clone()
                y = lorentz(s[0], 3.1, 2.2, 0.4, 0.1) + 0.3 * meas.standard_normal()
                o.pdf_update((s, y, 0.3))
                log.append((tuple(s), u.cpu().numpy(), o.particle_weights.copy(), bool(o.just_resampled)))
        torch.cuda.synchronize()
        return log

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert x[0] == y[0] and x[3] == y[3]
        assert np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2])


def test_the_constraint_mask_inside_the_gather_is_the_mask_after_it(hip):
    """Round 5: inside pdf_update() the gather of a resample zeroes the weights of new particles with sigma <= 0 itself and
    leaves mask_kernel's partial sums, so that enforce_parameter_constraints() (obe_noiseparam.py:57-79) is one launch
    instead of two.  Same particles masked, same weights, moments, counts, settings, generator — bit for bit — as with
    tuning_parameters['mask_in_gather'] = False; a resample() called on its own still leaves uniform weights."""
    import optbayesexpt_amd as obe
    logs, counts = {}, {}
    for flag in (True, False):
        o = make(obe, "auto", n_particles=70000, n_settings=1500, noise_param=True, threshold=0.9)
        o.tuning_parameters["mask_in_gather"] = flag
        used = []
        call = o._lib.call

        class Spy:                        # which of the two routes the constraint took
            def __getattr__(self, name):
                return getattr(o_lib, name)

            def call(self, name, *a):
                if name in ("obe_mask_renorm_moments", "obe_mask_nonpositive_moments", "obe_resample_particles_aos_masked"):
                    used.append(name)
                return call(name, *a)
        o_lib = o._lib
        o._lib = Spy()
        log = cycles(o, 25, between=lambda obj, c: np.array([obj.last_constraint_count], dtype=np.float64))
        logs[flag], counts[flag] = log, used
        assert sum(e["resampled"] for e in log) >= 5 and max(e["between"][0] for e in log) > 0
    same(logs[True], logs[False])
    assert "obe_mask_renorm_moments" in counts[True] and "obe_resample_particles_aos_masked" in counts[True]
    assert "obe_mask_renorm_moments" not in counts[False] and "obe_resample_particles_aos_masked" not in counts[False]
    # outside pdf_update() nothing follows the resample: uniform weights, as the reference's resample() leaves them
    o = make(obe, False, n_particles=70000, n_settings=300, noise_param=True)
    o.resample()
    w = o.particle_weights
    assert np.all(w == 1.0 / w.size)


def _run_tool(args, timeout):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable] + [os.path.join(root, args[0])] + args[1:], capture_output=True, text=True,
                          timeout=timeout, cwd=root)


def test_a_slice_of_the_randomised_parity_runs(hip):
    """150 seeded recipes of tools/fuzz_parity.py inside the suite (random models, cloud and grid
    sizes, zero weights, utility and selection methods, noise-parameter and sweeper objects — the product classes
    against the oracle classes step by step).  The long runs stay in the tool; their totals are in DESIGN.md."""
    r = _run_tool(["tools/fuzz_parity.py", "150", "20261003"], 900)
    assert r.returncode == 0 and "fuzz: 150 cases, 0 failures" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    print(r.stdout[-1500:])
    assert "complete: every cycle compared" in r.stdout          # (how the cases ended is part of the report)


def test_a_slice_of_the_speculative_soak(hip):
    """... and 2 000 cycles of tools/soak_speculative.py: random experiments with and without the sweep enqueued
    behind the update, random things done between pdf_update() and the next opt_setting(), bit for bit."""
    r = _run_tool(["tools/soak_speculative.py", "5", "77", "2000"], 900)
    assert r.returncode == 0 and "all equal" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    print(r.stdout[-300:])


def test_a_refused_enqueue_leaves_the_weights_alone(hip):
    """with no arrival counters in the library (OBE_CONTROL_SLOTS=0, a child process)
    obe_bayes_update_model_moments_enqueue() must refuse BEFORE its first launch, and the package's fallback to
    the synchronous form must reproduce the plain path bit for bit (tests/_refusal_check.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, OBE_CONTROL_SLOTS="0")
    r = subprocess.run([sys.executable, os.path.join(here, "_refusal_check.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "REFUSAL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_control_slots_change_hands_when_the_table_is_full(hip):
    """with room for TWO streams' arrival counters (OBE_CONTROL_SLOTS=2, a child process) an object
    driven from five streams in turn keeps the fused fold and the speculation on every one of them — the least
    recently used slot is handed on — and stays bit-identical to a run on one stream."""
    import os
    import subprocess
    import sys
    code = """
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import optbayesexpt_amd as obe
import test_gpu_speculative as t
ref = t.cycles(t.make(obe, True), 15)
streams = [torch.cuda.Stream() for _ in range(5)]
meas = np.random.default_rng(5)
o = t.make(obe, True)
log = []
for k in range(15):
    with torch.cuda.stream(streams[k %% 5]):
        s = o.opt_setting()
        y = t.lorentz(s[0], 3.1, 2.2, 0.4, 0.1) + 0.3 * meas.standard_normal()
        o.pdf_update((s, y, 0.3))
        log.append((tuple(float(v) for v in s), o.particle_weights.copy(), bool(o.just_resampled)))
    torch.cuda.synchronize()
for c, (e, r) in enumerate(zip(log, ref)):
    assert e[0] == r['setting'] and e[2] == r['resampled'] and np.array_equal(e[1], r['w']), c
assert not o.sweep_state()['unavailable']
print('SLOTS OK')
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OBE_CONTROL_SLOTS="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SLOTS OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_randoms_enqueued_ahead_are_the_resamples_own(hip):
    """Round 6: pdf_update() enqueues the uniforms and normals of the NEXT resample beside the update
    (tuning_parameters['randoms_ahead']; obe_resample_randoms_enqueue) and a resample that finds the generator where
    that chain started from uses them.  Same kernels on the same generator state: two objects, one with the chain
    enqueued ahead from the first update on, one never, through 40 cycles of a resample-heavy experiment — chosen
    settings, resample decisions, resample indices, particles, weights and the generator state BIT FOR BIT in every
    cycle; numbers kept through cycles that do not resample; a generator moved by the caller in between is a miss
    (thrown away, regenerated), two misses in a row end the habit."""
    import bench
    import optbayesexpt_amd as obe
    settings, prior, cons, true, sigma = bench.make_workload("c2")
    sv = (np.ascontiguousarray(settings[0][::16]),)

    def build(ahead):
        o = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior.copy(), cons, scale=False, utility_method="variance_full",
                             default_noise_std=sigma)
        o.tuning_parameters["randoms_ahead"] = ahead
        o.tuning_parameters["resample_threshold"] = 0.9        # resample in most cycles, not in all
        o.rng = np.random.default_rng(2026)
        return o

    a, b = build(True), build(False)
    sim = np.random.default_rng(7)
    resamples, kept = 0, 0
    for cyc in range(40):
        xa, xb = a.opt_setting(), b.opt_setting()
        assert a.last_setting_index == b.last_setting_index, cyc
        y = float(a.model_function(xa, true, cons)) + sigma * sim.standard_normal()
        if cyc in (25, 26, 31):
            for o in (a, b):
                o.rng.random()                       # the caller draws from the generator between two cycles
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            a.pdf_update((xa, y, sigma))
            b.pdf_update((xb, y, sigma))
        assert a.just_resampled == b.just_resampled, cyc
        resamples += a.just_resampled
        kept += (not a.just_resampled) and a.__dict__.get("_ahead") is not None
        if a.just_resampled:
            np.testing.assert_array_equal(a.last_resample_indices_device.cpu().numpy(),
                                          b.last_resample_indices_device.cpu().numpy(), err_msg=f"cycle {cyc}")
            np.testing.assert_array_equal(np.array(a.particles), np.array(b.particles), err_msg=f"cycle {cyc}")
        np.testing.assert_array_equal(np.array(a.particle_weights), np.array(b.particle_weights), err_msg=f"cycle {cyc}")
        assert a.rng.bit_generator.state == b.rng.bit_generator.state, cyc
    hits = a.__dict__.get("_ahead_hits", 0)
    print(f"randoms ahead: {resamples} resamples in 40 cycles, {hits} of them took the numbers enqueued ahead, "
          f"{kept} cycles kept them for later, misses in a row at the end: {a.__dict__.get('_ahead_misses', 0)}")
    assert resamples >= 10 and hits >= resamples - 6 and kept >= 1
    assert b.__dict__.get("_ahead_hits", 0) == 0 and b.__dict__.get("_ahead") is None
    # 'auto' (the default): only once the experiment has resampled, and only for the full sweep + opt_setting
    c = build("auto")
    assert not c._randoms_ahead_wanted()
    c._sweeps.resample_rate = 0.3
    assert c._randoms_ahead_wanted()
    d = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior.copy(), cons, scale=False, default_noise_std=sigma)
    d.tuning_parameters["randoms_ahead"] = "auto"
    d._sweeps.resample_rate = 0.3
    assert not d._randoms_ahead_wanted()             # reference semantics: every sweep draws from the generator
    # ... and the default is OFF (measured: no gain at the BASELINE sizes, profiles/r06_randoms_ahead.txt)
    e = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior.copy(), cons, scale=False, utility_method="variance_full",
                         default_noise_std=sigma)
    e._sweeps.resample_rate = 0.3
    import os
    assert not e._randoms_ahead_wanted() or os.environ.get("OBE_RANDOMS_AHEAD", "0") != "0"
