"""Host-side logic that needs no GPU: the model registry's NumPy forms, the in-place
write tracking of the host mirrors, settings sharding arithmetic, demo helpers, and the
"fail loudly without a GPU" contract."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import optbayesexpt_amd as obe
from optbayesexpt_amd import dist as obe_dist
from optbayesexpt_amd._mirror import TrackedArray
from oracle import models as omodels


def test_public_names_match_reference_package():
    # optbayesexpt/__init__.py:1-6 minus the TCP server/socket (out of scope)
    for name in ("ParticlePDF", "OptBayesExpt", "OptBayesExptNoiseParameter",
                 "MeasurementSimulator", "trace_sort"):
        assert hasattr(obe, name)


def test_device_model_numpy_forms_match_demo_formulas():
    g = np.random.default_rng(0)
    x = np.linspace(1.5, 4.5, 50)
    cases = [(obe.models.lorentzian(), omodels.lorentzian, (x,), g.normal(3, 1, (3, 1)), (0.1,)),
             (obe.models.lorentzian(7), omodels.multi_lorentzian(7), (x,), g.normal(3, 1, (10, 1)), (0.1,)),
             (obe.models.line_ab(), omodels.line_ab, (x,), g.normal(0, 1, (2, 1)), ()),
             (obe.models.line_mb(), omodels.line_mb, (x,), g.normal(0, 1, (3, 1)), ()),
             (obe.models.rabi(), omodels.rabi, (x[:, None], x[None, :] - 3), (2.0, 0.5), (1e5, 0.01, 2.0)),
             (obe.models.coil(), omodels.coil, (x,), (1.0, 0.1, 1.0, 0.3), ())]
    for dm, fn, sets, pars, cons in cases:
        assert_array_equal(dm(sets, pars, cons), fn(sets, pars, cons), err_msg=dm.name)
        assert dm.n_setdims == len(sets)
    # both broadcasting directions of the reference's calling convention
    dm = obe.models.lorentzian()
    pars = (g.uniform(2, 4, 100), g.uniform(-2000, -400, 100), g.normal(5e4, 1e3, 100))
    assert dm((3.0,), pars, (0.1,)).shape == (100,)
    assert dm((x,), (3.0, -1000.0, 5e4), (0.1,)).shape == (50,)


def test_tracked_array_reports_in_place_writes():
    from optbayesexpt_amd._mirror import Mirror
    m = Mirror("cpu", host=np.arange(12, dtype=float).reshape(3, 4))
    m._dev_valid = True              # pretend the device copy is current
    base, stamps = m._host, [m.version]

    def writes():                    # a tracked write invalidates the device copy and bumps the version
        seen = (not m._dev_valid) and m.version != stamps[-1]
        stamps.append(m.version)
        m._dev_valid = True
        return seen
    v = m.host()
    assert isinstance(v, TrackedArray) and not v.flags.writeable
    v[1, 2] = 5.0
    assert writes() and base[1, 2] == 5.0
    row = v[1]                       # views stay linked (self.parameters[1][idx] = 0)
    row[0] = 3.0
    assert writes() and base[1, 0] == 3.0
    for i in np.argwhere(v[2] > 0):  # the reference's loop-of-index-writes idiom
        v[2][i] = 0
    assert writes() and not base[2].any()
    c = v.copy()                     # copies are plain, writable data
    c[0, 0] = -1
    c.fill(7.0)
    assert not writes()
    assert isinstance(v / v.sum(), np.ndarray) and not isinstance(v / 2, TrackedArray)
    v /= 2.0                         # in-place ufunc
    assert writes()
    np.multiply(v, 2.0, out=v)
    assert writes()
    assert v.tolist()[0][1] == 1.0 and np.sum(v) == base.sum()
    # every untracked way of writing in place is refused loudly instead of being lost
    before = base.copy()
    for bad in (lambda: np.copyto(v, 0.0), lambda: v.fill(0.0), lambda: v.sort(), lambda: np.put(v, [0], 9.0),
                lambda: np.putmask(v, v > 1, 0.0), lambda: v.flat.__setitem__(0, 9.0),
                lambda: np.nan_to_num(v, copy=False), lambda: v[0].partition(1)):
        with pytest.raises(ValueError, match="read-only"):
            bad()
    assert not writes() and np.array_equal(base, before)
    # a view taken before the mirror replaced its buffer is a snapshot (an array kept across
    # pdf_update in the reference): writable through the same idioms, without effect on the mirror
    m._host = base.copy()
    v[0, 0] = 123.0
    assert not writes() and m._host[0, 0] != 123.0 and v[0, 0] == 123.0


def test_shard_bounds_partition():
    for n in (1, 7, 8, 9, 201, 4096, 65536):
        for world in (1, 2, 3, 8):
            edges = [obe_dist.shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for (b0, e0), (b1, e1) in zip(edges, edges[1:]):
                assert e0 == b1 and e0 >= b0
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_first_max_is_numpy_argmax_over_rank_winners():
    g = np.random.default_rng(1)
    for trial in range(200):
        n, world = int(g.integers(4, 40)), int(g.integers(1, 5))
        u = g.normal(size=n).round(1)            # coarse values => ties
        if trial % 7 == 0:
            u[g.integers(0, n)] = np.nan
        vals, idxs = [], []
        for r in range(world):
            b, e = obe_dist.shard_bounds(n, r, world)
            if e > b:
                k = int(np.argmax(u[b:e]))
                vals.append(u[b + k])
                idxs.append(b + k)
        k = obe_dist.first_max(np.array(vals), np.array(idxs))
        assert idxs[k] == int(np.argmax(u))


def test_trace_sort_and_simulator():
    s, m, e, n = obe.trace_sort(np.array([3., 1., 2., 1., 3., 3.]), np.array([1., 2., 3., 4., 5., 9.]))
    assert s == [1., 2., 3.] and n == [2, 1, 3]
    assert_allclose(m, [3., 3., 5.])
    assert_allclose(e, [np.std([2., 4.]) / np.sqrt(2), 0.0, np.std([1., 5., 9.]) / np.sqrt(3)])
    sim = obe.MeasurementSimulator(obe.models.lorentzian(), (3.0, -1000.0, 5e4), (0.1,), noise_level=0.0)
    assert_allclose(sim.simdata((3.0,)), 49000.0)
    assert sim.simdata((np.array([1.0, 3.0]),)).shape == (2,)


def test_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        obe.ParticlePDF(np.zeros((2, 8)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        obe.OptBayesExpt(obe.models.line_ab(), (np.arange(3.),), np.zeros((2, 8)), ())


def test_product_package_never_imports_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.abspath(obe.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_expression_translator():
    """models.from_expression's translator (no compilation here): the NumPy form reproduces
    the demo formulas bit for bit, the generated header has the model interface, and
    anything outside the small expression language is rejected."""
    from optbayesexpt_amd import _exprmodel
    g = np.random.default_rng(3)
    x = np.linspace(1.5, 4.5, 40)
    header, form, digest = _exprmodel.translate("b + a / (((x - x0) / d)**2 + 1)", ("x",), ("x0", "a", "b"), ("d",))
    pars = (g.uniform(2, 4, 1), g.uniform(-2000, -400, 1), g.normal(5e4, 1e3, 1))
    assert_array_equal(form((x,), pars, (0.1,)), omodels.lorentzian((x,), pars, (0.1,)))
    assert "struct PluginModel" in header and "NS = 1, NC = 1, NREAD = 3, NCONST = 1" in header
    assert "sq(" in header and len(digest) == 16
    # the sweep form: x/d hoisted per setting, x0/d and sqrt(w)*a, sqrt(w)*b per particle, the
    # division batched over the settings of a lane, the sums as FMAs
    assert "NXS = 1, NPK = 3" in header and "xs[0] = (x[0] * (1.0 / m.consts[0]));" in header
    assert "pk[1] = (th(1) * sw);" in header and "batch_div_poisoned<SPT>(den0, pk[1], ip0);" in header
    assert "v[j][0] = fma(ip0[j / 2], sibling_of<SPT>(den0, j), pk[2]);" in header
    # two particles through one inversion tree
    assert "batch_div_poisoned2<SPT>(den0a, den0b, pa[1], pb[1], ip0a, ip0b);" in header
    assert "vb[j][0] = fma(ip0b[j / 2], sibling_of<SPT>(den0b, j), pb[2]);" in header
    # and its always-IEEE twin for the repeat after a poisoned sweep
    assert "sweep_eval_safe" in header and "r0[j] = guarded_rcp(den0[j]);" in header
    assert _exprmodel.translate("b + a / (((x - x0) / d)**2 + 1)", ("x",), ("x0", "a", "b"), ("d",))[2] == digest
    rabi = ("baseline*(1 - exp(-t/T1)*contrast/2*(1 - cos(pi*2*hypot(df - fc, B1)*t))/(((df - fc)/B1)**2 + 1))")
    _, form2, _ = _exprmodel.translate(rabi, ("t", "df"), ("B1", "fc"), ("baseline", "contrast", "T1"))
    sets = (x[:, None] / 5, x[None, :] - 3)
    assert_array_equal(form2(sets, (2.0, 0.5), (1e5, 0.01, 2.0)), omodels.rabi(sets, (2.0, 0.5), (1e5, 0.01, 2.0)))
    _, form3, _ = _exprmodel.translate(("cos(w*t)", "sin(w*t) + c0"), ("t",), ("w",), ("c0",))
    assert form3((x,), (2.0,), (1.0,)).shape == (2, 40)
    for bad in ("__import__('os')", "x if a else b", "a[0]", "lambda: 1", "foo(x)", "x @ a", "exp(x, a)", "y + 1"):
        with pytest.raises((ValueError, SyntaxError)):
            _exprmodel.translate(bad, ("x",), ("a", "b"), ())
    with pytest.raises(ValueError):
        _exprmodel.translate("x + exp", ("x",), ("exp",), ())      # a parameter may not shadow a function


def test_model_function_source_translation():
    """models.from_function's front end (no compilation here): reference-style functions become
    expressions over generated names, the NumPy form of the result reproduces the function bit
    for bit, and anything that is not straight-line arithmetic is refused."""
    import _fn_models
    from optbayesexpt_amd import _exprmodel, _fnmodel
    e = _fnmodel.expressions_from_function(_fn_models.lorentzian)
    assert e == (("p2 + p1 / (((s0 - p0) / c0) ** 2 + 1)",), ("s0",), ("p0", "p1", "p2"), ("c0",))
    assert _fnmodel.expressions_from_function(_fn_models.lorentzian_via_helper)[0] == \
        ("p2 + p1 / (((s0 - p0) * 2 / c0) ** 2 + 1)",)                # the helper's body, inlined
    for fn, shape in ((_fn_models.lorentzian, (1, 1, 3, 1)), (_fn_models.rabi, (1, 2, 2, 3)),
                      (_fn_models.lorentzian_via_helper, (1, 1, 3, 1)),
                      (_fn_models.two_channels, (2, 1, 2, 1))):
        exprs, sets, pars, cons = _fnmodel.expressions_from_function(fn)
        assert (len(exprs), len(sets), len(pars), len(cons)) == shape
        _, numpy_form, _ = _exprmodel.translate(exprs, sets, pars, cons)
        _fnmodel.check_against_function(fn, numpy_form, len(sets), len(pars), len(cons))
    for fn in (_fn_models.with_branch, _fn_models.with_complex, _fn_models.with_global, _fn_models.with_loop,
               lambda s, p, c: p[0], np.sin):
        with pytest.raises(ValueError):
            _fnmodel.expressions_from_function(fn)
    # a wrong translation would be caught by the bitwise check
    _, wrong, _ = _exprmodel.translate(("p2 + p1 / (((s0 - p0) / c0) ** 2 + 1.0000001)",), ("s0",), ("p0", "p1", "p2"), ("c0",))
    with pytest.raises(ValueError):
        _fnmodel.check_against_function(_fn_models.lorentzian, wrong, 1, 3, 1)


def test_plugin_cleanup_only_touches_our_own_files(tmp_path, monkeypatch):
    """A shared OBE_PLUGIN_DIR may hold files that are not ours: only regular files that carry the
    package's plugin naming pattern *and* an outdated source fingerprint are removed."""
    from optbayesexpt_amd import build
    monkeypatch.setattr(build, "PLUGIN_DIR", str(tmp_path))
    fp = build._source_fingerprint()
    old = "0" * 12 if fp != "0" * 12 else "1" * 12
    keep = ["notes.txt", "libfoo.so", f"libobe_model_{'a' * 16}_{fp}.so", f"obe_model_{'a' * 16}_{fp}.h",
            f"libobe_model_{'a' * 16}_{old}.so.bak", f"xlibobe_model_{'a' * 16}_{old}.so",
            f"obe_model_{'a' * 16}_{old}_obe_sweep.o"]
    gone = [f"libobe_model_{'b' * 16}_{old}.so", f"obe_model_{'b' * 16}_{old}.h"]
    for f in keep + gone:
        (tmp_path / f).write_text("x")
    (tmp_path / "subdir").mkdir()
    (tmp_path / f"libobe_model_{'c' * 16}_{old}.so").mkdir()        # a directory with a plugin's name
    build._remove_outdated_plugins(fp)
    left = set(os.listdir(tmp_path))
    assert left == set(keep) | {"subdir", f"libobe_model_{'c' * 16}_{old}.so"}


_RACE = r"""
import os, sys, time
sys.path.insert(0, {root!r})
from optbayesexpt_amd import build
build.PLUGIN_DIR = {pdir!r}
build.HIPCC = {hipcc!r}
build.FLAGS = []
build.PLUGIN_SOURCES = ["obe_capi.hip"]
build.PLUGIN_COMMON_SOURCES = []
lib = build.build_plugin("// header\n", "d" * 16)
assert os.path.getsize(lib) > 0
print(open(lib).read().count("linked"))
"""


def test_concurrent_plugin_builds_are_serialised(tmp_path):
    """One process per GPU: every rank asks for the same plugin at once.  With a slow stand-in
    for hipcc, exactly one of three racing processes links; the others wait on the lock and find the
    finished library (never a half-written one), and nobody deletes anybody's objects."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "fake_hipcc"
    log = tmp_path / "calls.log"
    fake.write_text(f"""#!/usr/bin/env python3
import sys, time
out = sys.argv[sys.argv.index("-o") + 1]
kind = "linked" if "-shared" in sys.argv else "compiled"
open({str(log)!r}, "a").write(kind + "\\n")
with open(out, "w") as f:          # a slow, non-atomic writer
    f.write(kind[:3]); f.flush(); time.sleep(0.4); f.write(kind[3:] + "\\n")
""")
    fake.chmod(0o755)
    pdir = tmp_path / "plugins"
    (pdir).mkdir()
    (pdir / "notes.txt").write_text("mine")
    script = _RACE.format(root=root, pdir=str(pdir), hipcc=str(fake))
    procs = [subprocess.Popen([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for _ in range(3)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert [o[0].strip() for o in outs] == ["1", "1", "1"]          # everyone read a complete library
    calls = log.read_text().split()
    assert calls.count("linked") == 1 and calls.count("compiled") == 1
    assert (pdir / "notes.txt").read_text() == "mine"
    assert not [f for f in os.listdir(pdir) if f.startswith(".build_")]


def test_device_bound_library_switches_device_around_calls(monkeypatch):
    """An object created for another GPU than the current one launches with ITS device current:
    DeviceBound wraps every library call in torch.cuda.device(index) unless it already is."""
    import contextlib
    import torch
    from optbayesexpt_amd import _lib
    log = []

    class FakeLib:
        def call(self, name, *args):
            log.append(("call", name, current[0]))
            return 0

        def workspace_bytes(self, *a):
            return 64

    current = [0]

    @contextlib.contextmanager
    def fake_device(idx):
        prev, current[0] = current[0], idx
        log.append(("enter", idx))
        try:
            yield
        finally:
            current[0] = prev
            log.append(("exit", idx))
    monkeypatch.setattr(torch.cuda, "current_device", lambda: current[0])
    monkeypatch.setattr(torch.cuda, "device", fake_device)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    same = _lib.DeviceBound(FakeLib(), torch.device("cuda", 0))
    other = _lib.DeviceBound(FakeLib(), torch.device("cuda", 1))
    same.call("obe_x")
    assert log == [("call", "obe_x", 0)]
    del log[:]
    other.call("obe_y")
    assert log == [("enter", 1), ("call", "obe_y", 1), ("exit", 1)] and current[0] == 0
    assert other.workspace_bytes(1, 1, 1, 1) == 64                     # plain delegation
    assert _lib.DeviceBound(other, torch.device("cuda", 1))._lib is other._lib      # no double wrapping


def test_classes_carry_every_public_method_of_the_reference_classes():
    """The public (non-underscore) members of the reference's three classes at v1.2.0 (recorded by
    introspection of optbayesexpt.ParticlePDF / OptBayesExpt / OptBayesExptNoiseParameter): every one
    exists here under the same name (the class surface is the drop-in boundary)."""
    import optbayesexpt_amd as obe
    pdf = ["bayesian_update", "covariance", "mean", "randdraw", "resample", "resample_test", "set_pdf", "std"]
    base = pdf + ["cost_estimate", "enforce_parameter_constraints", "eval_over_all_parameters",
                  "eval_over_all_settings", "get_setting", "good_setting", "likelihood", "opt_setting",
                  "pdf_update", "random_setting", "set_n_draws", "utility", "utility_full_kld",
                  "utility_max_min", "utility_pseudo", "utility_variance", "y_var_noise_model",
                  "yvar_from_entropy", "yvar_from_parameter_draws", "yvar_max_min", "yvar_noise_model"]
    for cls, names in ((obe.ParticlePDF, pdf), (obe.OptBayesExpt, base), (obe.OptBayesExptNoiseParameter, base)):
        missing = [n for n in names if not callable(getattr(cls, n, None))]
        assert not missing, (cls.__name__, missing)


def test_bench_launches_its_own_ranks_and_reports_failure(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks itself (fresh
    processes, before anything touches the GPU).  Without a GPU both ranks fail: the launcher must
    come back with a non-zero exit code and no JSON line, not hang."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("the failure path needs a box without a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "c1", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "ranks failed" in r.stderr


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_bench_boundary_update_runs_without_the_enqueued_sweep():
    """bench.update_at_boundary: the last update before a timed region starts or ends must not enqueue the next
    cycle's sweep (a timed region of K cycles then holds exactly K sweeps), and leaves the object's tuning
    parameters as it found them — also when the update raises."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class Stub:
        def __init__(self, tp, fail=False):
            self.tuning_parameters, self.seen, self.fail = tp, [], fail

        def pdf_update(self, record):
            self.seen.append((record, self.tuning_parameters.get("speculative_sweep", "absent")))
            if self.fail:
                raise RuntimeError("boom")

    for before in ({}, {"speculative_sweep": "auto"}, {"speculative_sweep": True, "a_param": 0.98}):
        o = Stub(dict(before))
        bench.update_at_boundary(o, ("x", 1.0))
        assert o.seen == [(("x", 1.0), False)] and o.tuning_parameters == before
    o = Stub({"speculative_sweep": True}, fail=True)
    with pytest.raises(RuntimeError):
        bench.update_at_boundary(o, ("x", 1.0))
    assert o.tuning_parameters == {"speculative_sweep": True}


# ------------------------------------------------------------------------------------------------
# The sweep-path state machine (optbayesexpt_amd/_sweepstate.py) driven with a recording I/O object:
# no device, no library.
def _load_sweepstate():
    from optbayesexpt_amd import _sweepstate
    return _sweepstate


class _FakeIO:
    def __init__(self):
        self.log, self.armed = [], False

    def wait_words(self, words, n, stream):
        self.log.append(("wait", words, n, stream))

    def still_armed(self, block):
        return self.armed

    def synchronize(self):
        self.log.append(("sync",))


def _state(ss, io):
    import types
    return ss.SweepState(io, types.SimpleNamespace(KAPPA_ENTER=1000.0, KAPPA_LEAVE=3000.0, SAFE_STREAK=3, SAFE_RETRY=64))


def _inputs(cloud=(1, 1), shifted=True, noise="n", settings=(0, 100), alias=True, cost_hook=False):
    return dict(cloud=cloud, shifted=shifted, noise=noise, settings=settings, alias=alias, cost_hook=cost_hook)


def test_sweepstate_pattern_and_auto_policy():
    ss = _load_sweepstate()
    st = _state(ss, _FakeIO())
    assert st.pattern is ss.Pattern.COLD and not st.speculation_wanted("auto")
    assert st.speculation_wanted(True) and not st.speculation_wanted(False) and not st.speculation_wanted("never")
    for k, want in ((1, ss.Pattern.WARM), (2, ss.Pattern.STEADY), (3, ss.Pattern.STEADY)):
        st.update_finished((k, k), resampled=False)
        st.full_sweep_requested((k, k))
        assert st.pattern is want
    assert st.speculation_wanted("auto")
    # a sweep of some OTHER cloud (the caller wrote new weights in between) breaks the pattern
    st.update_finished((4, 4), resampled=False)
    st.full_sweep_requested((4, 5))
    assert st.pattern is ss.Pattern.COLD and not st.speculation_wanted("auto")
    # ... as does a sweep nobody's update preceded
    st.full_sweep_requested((4, 5))
    assert st.pattern is ss.Pattern.COLD
    # mostly-resampling updates: 'auto' stops guessing behind the update, but still sends the sweep of a resampled cloud
    for k in range(6, 12):
        st.update_finished((k, k), resampled=True)
        st.full_sweep_requested((k, k))
    assert st.pattern is ss.Pattern.STEADY and st.resample_rate > 0.5
    assert not st.speculation_wanted("auto") and st.speculation_wanted("auto", after_resample=True)
    st.library_refused()
    assert not st.speculation_wanted(True) and st.describe()["unavailable"]


def test_sweepstate_resample_behind_a_speculative_sweep():
    ss = _load_sweepstate()
    io = _FakeIO()
    st = _state(ss, io)
    st.streak = 5
    st.enqueued(ss.Ticket(_inputs(), "words", "block", None, "s1", 1))
    assert st.pending is ss.Pending.ENQUEUED
    st.update_delivered(resampled=True)
    assert st.pending is ss.Pending.ABORTED
    # the sweep did nothing: not taken, nothing to wait for (its words stay armed), the pattern is not blamed
    assert st.take(_inputs(), 1) is None
    assert io.log == [] and st.pending is ss.Pending.NONE and st.ticket is None and st.streak == 5
    # the sweep of the resampled cloud is certain
    st.enqueued(ss.Ticket(_inputs(cloud=(2, 2)), "words", "block", None, "s1", 1), certain=True)
    assert st.pending is ss.Pending.RAN
    st.update_delivered(resampled=True)            # (a late delivery changes nothing: not ENQUEUED)
    assert st.pending is ss.Pending.RAN
    t = st.take(_inputs(cloud=(2, 2)), 1)
    assert t is not None and io.log == [("wait", "words", 3, "s1")] and st.pending is ss.Pending.NONE


def test_sweepstate_stream_change_between_the_calls():
    ss = _load_sweepstate()
    # unsharded (page-locked result words): collected on the stream it was launched on, whatever is current now
    io = _FakeIO()
    st = _state(ss, io)
    st.enqueued(ss.Ticket(_inputs(), "words", "block", None, "s1", 1))
    st.update_delivered(resampled=False)
    assert st.take(_inputs(), 2) is not None and io.log == [("wait", "words", 3, "s1")]
    # sharded (the record stays on the device): only usable on the same stream; dropped with a device sync otherwise
    io = _FakeIO()
    st = _state(ss, io)
    st.streak = 4
    st.enqueued(ss.Ticket(_inputs(), None, "block", "record", "s1", 1))
    st.update_delivered(resampled=False)
    assert st.take(_inputs(), 2) is None and io.log == [("sync",)] and st.streak == 0
    st.enqueued(ss.Ticket(_inputs(), None, "block", "record", "s1", 1))
    st.update_delivered(resampled=False)
    t = st.take(_inputs(), 1)
    assert t is not None and t.record == "record" and io.log == [("sync",)]      # (no wait: the collective reads it)


def test_sweepstate_hook_replaced_or_set_pdf_between_the_calls():
    ss = _load_sweepstate()
    for changed in (dict(cost_hook=True), dict(cloud=(9, 9)), dict(noise="other"), dict(alias=False),
                    dict(shifted=False), dict(settings=(0, 50))):
        io = _FakeIO()
        st = _state(ss, io)
        st.streak = 3
        st.enqueued(ss.Ticket(_inputs(), "words", "block", None, "s1", 1))
        st.update_delivered(resampled=False)
        # not the sweep being asked for: forgotten — but it RAN and will write its words: waited for before
        # anything arms them again; two plain cycles before the next attempt
        assert st.take(_inputs(**changed), 1) is None, changed
        assert io.log == [("wait", "words", 3, "s1")] and st.streak == 0 and st.ticket is None
    # an undelivered result (the stream drained, the words still armed): not run
    io = _FakeIO()
    io.armed = True
    st = _state(ss, io)
    st.enqueued(ss.Ticket(_inputs(), "words", "block", None, "s1", 1))
    st.update_delivered(resampled=False)
    assert st.take(_inputs(), 1) is None and st.pending is ss.Pending.NONE
    # drop() of nothing is nothing
    st.drop(1)
    assert io.log == [("wait", "words", 3, "s1")]


def test_sweepstate_kappa_crossing_both_thresholds():
    ss = _load_sweepstate()
    st = _state(ss, _FakeIO())
    assert st.shifted_for_next_sweep("auto", full=True)                       # starts shifted
    assert not st.shifted_for_next_sweep("never", True) and st.shifted_for_next_sweep("always", True)
    assert st.shifted_for_next_sweep("never", full=False)                     # draws mode: always shifted
    assert not st.sweep_reported_kappa("auto", True, True, 500.0) and st.unshifted        # below ENTER: go unshifted
    assert not st.shifted_for_next_sweep("auto", True)
    assert not st.sweep_reported_kappa("auto", True, False, 2000.0) and st.unshifted      # between: kept (hysteresis)
    assert st.sweep_reported_kappa("auto", True, False, 5000.0) and not st.unshifted      # above LEAVE: repeat shifted
    assert not st.sweep_reported_kappa("auto", True, True, 2000.0) and not st.unshifted   # shifted, above ENTER: stays
    st.unshifted = True
    assert st.sweep_reported_kappa("auto", True, False, float("nan")) and not st.unshifted  # NaN: never kept
    assert not st.sweep_reported_kappa("auto", True, True, float("nan")) and not st.unshifted
    st.unshifted = True                  # the other modes and draws-mode sweeps leave the hysteresis alone
    assert not st.sweep_reported_kappa("never", True, False, 1e9) and st.unshifted
    assert not st.sweep_reported_kappa("auto", False, True, 1.0) and st.unshifted


def test_sweepstate_form_policy():
    ss = _load_sweepstate()
    st = _state(ss, _FakeIO())
    assert st.form_for_next_sweep() is ss.Form.FAST
    for k in range(3):
        assert st.form is ss.Form.FAST
        st.fast_form_left_its_range()
    assert st.form is ss.Form.SAFE
    for k in range(63):                                   # pinned: 63 sweeps with the twin ...
        assert st.form_for_next_sweep() is ss.Form.SAFE
    assert st.form_for_next_sweep() is ss.Form.FAST       # ... the 64th probes the fast form
    st.fast_form_left_its_range()                         # a failure pins it again at once
    assert st.form is ss.Form.SAFE and st.safe_run == 0
    for k in range(63):
        st.form_for_next_sweep()
    assert st.form_for_next_sweep() is ss.Form.FAST
    st.fast_form_held()
    assert st.form is ss.Form.FAST and st.safe_streak == 0
    st.range_hint(False)
    assert st.form is ss.Form.SAFE
    st.range_hint(None)
    assert st.form is ss.Form.SAFE
    st.range_hint(True)
    assert st.form is ss.Form.FAST and st.safe_streak == 2          # one fast attempt; a failure pins it again
    st.fast_form_left_its_range()
    assert st.form is ss.Form.SAFE


def test_released_landing_zones_wait_in_limbo(monkeypatch):
    """Page-locked landing zones that an object releases are not handed back to the allocator while a kernel of
    that object may still write to them (_lib.pinned_array): the storage moves to a limbo list — with the device
    whose kernels write it — when its array is collected, and the list is emptied only when it has grown to its
    limit, behind a synchronisation of exactly the devices named in it (a rank of an 8-GPU job must not
    create a context on the seven GPUs it does not use).  Here the page-locking itself is replaced by a no-op."""
    import gc
    import torch
    from optbayesexpt_amd import _lib
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)
    monkeypatch.setattr(_lib, "_LIMBO", [])
    synced = []
    monkeypatch.setattr(torch.cuda, "synchronize", lambda d=None: synced.append(d))
    a = _lib.pinned_array(4, device=3)
    view = a[1:3]
    b = _lib.pinned_array(3, np.int64, device=3)
    assert a.shape == (4,) and a.dtype == np.float64 and not a.any() and b.dtype == np.int64
    del a
    gc.collect()
    assert _lib._LIMBO == []                   # a view keeps the array, and with it the storage, with its owner
    del view, b
    gc.collect()
    assert len(_lib._LIMBO) == 2               # both storages are parked, not freed
    assert [d for _, d in _lib._LIMBO] == [3, 3]
    monkeypatch.setattr(_lib, "_LIMBO_MAX", 5)
    for _ in range(3):                         # ... until the list reaches its limit: the next allocation empties it
        _lib.pinned_array(1, device=5)         # (dropped at once: parked)
    gc.collect()
    assert len(_lib._LIMBO) == 5 and synced == []
    keep = _lib.pinned_array(2, device=5)
    assert len(_lib._LIMBO) == 0 and keep.shape == (2,)
    assert synced == [3, 5]                    # the devices of the parked blocks, each once — and no other


def test_bench_finds_the_cores_of_a_gpus_numa_node_from_sysfs(tmp_path, monkeypatch):
    """bench.py pins every rank of an N > 1 run to the cores of its GPU's NUMA node before torch is imported; the
    lookup is sysfs only (KFD topology order = HIP device order -> DRM render node -> local_cpulist).  A fake tree:
    two CPU nodes, three GPUs on two NUMA nodes; visibility lists remap the device index; anything missing -> None."""
    import bench
    sysfs = tmp_path / "sys"
    nodes = sysfs / "class/kfd/kfd/topology/nodes"
    layout = {0: (0, None), 1: (0, None), 2: (256, 128), 3: (256, 130), 4: (256, 129)}     # node -> (simd_count, render minor)
    for k, (simd, minor) in layout.items():
        (nodes / str(k)).mkdir(parents=True)
        text = f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\n"
        if minor is not None:
            text += f"drm_render_minor {minor}\nlocation_id 1234\n"
        (nodes / str(k) / "properties").write_text(text)
    for minor, (cpus, numa) in {128: ("0-31,128-159", 0), 130: ("32-63,160-191", 1), 129: ("0-31,128-159", 0)}.items():
        dev = sysfs / f"class/drm/renderD{minor}/device"
        dev.mkdir(parents=True)
        (dev / "local_cpulist").write_text(cpus + "\n")
        (dev / "numa_node").write_text(f"{numa}\n")
    # a GPU of the node that this container was not given: its properties cannot be read (EPERM on the pool's boxes;
    # here: a directory in the file's place) — skipped, as the runtime skips it
    (nodes / "9").mkdir()
    (nodes / "9" / "properties").mkdir()
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    cpus, numa = bench.gpu_local_cpus(0, sysfs=str(sysfs))
    assert numa == 0 and cpus == list(range(0, 32)) + list(range(128, 160))
    assert bench.gpu_local_cpus(1, sysfs=str(sysfs))[1] == 1                       # KFD order, not minor order
    assert bench.gpu_local_cpus(2, sysfs=str(sysfs))[1] == 0
    assert bench.gpu_local_cpus(3, sysfs=str(sysfs)) is None                       # no such device
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,2")
    assert bench.gpu_local_cpus(0, sysfs=str(sysfs))[1] == 1                       # device 0 of this process is GPU 1
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2,1,0")
    assert bench.gpu_local_cpus(0, sysfs=str(sysfs))[1] == 1                       # ROCR first ([2,1,0]), then HIP picks [1,2] of it -> GPU 1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-abcdef")
    assert bench.gpu_local_cpus(0, sysfs=str(sysfs)) is None                       # UUID lists: nothing is pinned
    assert bench.gpu_local_cpus(0, sysfs=str(tmp_path / "nowhere")) is None
    assert bench.format_cpulist(bench.parse_cpulist("0-3,8,10-11\n")) == "0-3,8,10-11"
    assert bench.parse_cpulist("") == []
