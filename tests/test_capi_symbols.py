"""The C-ABI library (CPU-only checks, no kernel launches): it loads, exports exactly the
functions include/obe_hip.h declares with the argument counts the ctypes binding uses,
and its host-side argument validation reports errors without touching a GPU."""
import ctypes
import re

import numpy as np
import pytest

from optbayesexpt_amd import _lib, models


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def _header_prototypes():
    text = open(_lib.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(obe_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return protos


def test_library_exports_every_declared_symbol(lib):
    declared = _lib.declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib.cdll, name), f"{name} declared in obe_hip.h but not exported"
    assert set(declared) == set(_lib._SIGNATURES), "ctypes table and header disagree"
    assert lib.cdll.obe_abi_version() == _lib.OBE_ABI_VERSION == 2


def _dynamic_symbols(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_library_exports_nothing_but_the_declared_entry_points(lib):
    """`nm -D --defined-only` of libobe_hip.so == the names include/obe_hip.h declares (OBE_API), no more: the
    C++ helpers (obe::wait_host_words, obe::stream_control_words, ...), the kernels' host stubs and handle objects
    are internal (-fvisibility=hidden + the link-time export list, optbayesexpt_amd/build.py)."""
    import shutil
    if shutil.which("nm") is None:
        pytest.skip("no nm on this box")
    declared = _lib.declared_symbols()
    assert _dynamic_symbols(lib.path) == declared


def test_plugin_exports_nothing_but_the_declared_entry_points(lib):
    """... and the same for a per-model plugin library: it defines a SUBSET of the header's names (the
    model-dependent entry points + the three library queries) and nothing else, so that two plugins and the library
    in one process cannot interpose one another's internals."""
    import os
    import shutil
    from optbayesexpt_amd import build
    if shutil.which("nm") is None:
        pytest.skip("no nm on this box")
    model = None
    try:
        model = models.from_expression("b + a / (((x - x0) / d)**2 + 1)", settings=("x",),
                                       parameters=("x0", "a", "b"), constants=("d",))
    except RuntimeError:
        if os.path.exists(build.HIPCC):
            raise
        pytest.skip("plugin not prebuilt and no hipcc here")
    syms = _dynamic_symbols(model.plugin_path)
    assert set(_lib.MODEL_ENTRY_POINTS) | {"obe_abi_version", "obe_last_error", "obe_source_fingerprint"} <= set(syms)
    assert set(syms) <= set(_lib.declared_symbols()), sorted(set(syms) - set(_lib.declared_symbols()))


def _header_parameter_names():
    text = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER_PATH).read(), flags=re.S)
    names = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(obe_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        names[m.group(1)] = [] if args in ("", "void") else [a.strip().split()[-1].lstrip("*") for a in args.split(",")]
    return names


def test_audit_rules_address_the_parameters_they_name():
    """OBE_CHECK_DELIVERY's rules (optbayesexpt_amd/_audit.py) pick host-word pointers and counts out of a call's
    argument tuple by POSITION; the positions are pinned here to the parameter names of include/obe_hip.h, and every
    entry point that has a page-locked result parameter has a rule (or is listed as waiting for nothing new)."""
    from optbayesexpt_amd import _audit
    names = _header_parameter_names()
    assert set(_audit._RULES) == set(_audit.RULE_PARAMETERS)
    for fn, expect in _audit.RULE_PARAMETERS.items():
        for index, name in expect.items():
            assert names[fn][index] == name, (fn, index, names[fn][index], name)
    # every entry point with a host result parameter is covered by a rule
    host_results = ("h_out", "h_pinned_out", "h_pinned_word", "h_pinned_words", "h_best", "h_best_idx", "h_kappa", "h_f64",
                    "h_i64", "h_total", "h_total_pinned", "h_moments", "h_changed")
    unruled = {fn for fn, ps in names.items() if any(p in host_results for p in ps)} - set(_audit._RULES)
    # (these deliver by a synchronous copy or wait before they return, into memory the audit does not track as armed)
    assert unruled <= {"obe_moments", "obe_bayes_update_sweep", "obe_likelihood_y"}, sorted(unruled)


def test_ctypes_argument_counts_match_header():
    protos = _header_prototypes()
    assert set(protos) == set(_lib._SIGNATURES)
    for name, n_args in protos.items():
        assert len(_lib._SIGNATURES[name][1]) == n_args, name


def test_model_struct_layout_and_validation(lib):
    assert ctypes.sizeof(_lib.ObeModelStruct) == 6 * 4 + 8 * 8
    m = models.lorentzian(1).struct(3, (0.1,))
    assert lib.cdll.obe_model_validate(m) == 0
    assert (m.n_setdims, m.n_channels) == (1, 1)
    m7 = models.lorentzian(7).struct(10, (0.1,))
    assert lib.cdll.obe_model_validate(m7) == 0
    bad = models.lorentzian(1).struct(3, (0.1,))
    bad.n_params = 2                                   # fewer rows than the model reads
    assert lib.cdll.obe_model_validate(bad) == -1
    assert "n_params" in lib.last_error()
    bad = models.lorentzian(1).struct(3, (0.1,))
    bad.aux = 9
    assert lib.cdll.obe_model_validate(bad) == -1
    bad = models.coil().struct(4, ())
    bad.n_channels = 1
    assert lib.cdll.obe_model_validate(bad) == -1
    bad.id = 99
    assert lib.cdll.obe_model_validate(bad) == -1 and "unknown model" in lib.last_error()
    with pytest.raises(ValueError):
        models.lorentzian(1).struct(3, ())             # missing constant d
    with pytest.raises(ValueError):
        models.rabi().struct(1, (1.0, 0.1, 2.0))       # too few parameter rows


def test_workspace_and_moment_sizes(lib):
    assert lib.moments_len(3) == 2 + 12 + 9
    small = lib.workspace_bytes(1000, 10, 1, 3)
    big = lib.workspace_bytes(1 << 20, 65536, 1, 3)
    assert 0 < small < big < 1 << 30
    assert lib.workspace_bytes(1 << 20, 65536, 2, 3) > big


def test_workspace_covers_every_smaller_draw_count(lib):
    """A sweep of N_DRAWS <= n draws runs in the workspace sized for n (the public N_DRAWS attribute
    can be set to anything): the size must not shrink when the draw count grows, although the
    number of particle chunks of the plan is not monotone (the chunk length is rounded up to whole
    waves after the count is chosen)."""
    g = np.random.default_rng(12)
    for _ in range(4000):
        n = int(g.integers(300, 3_000_000))
        ns = int(g.integers(1, 70_000))
        nd = int(g.integers(max(1, n // 3), n + 1))
        assert lib.workspace_bytes(nd, ns, 1, 3) <= lib.workspace_bytes(n, ns, 1, 3), (n, ns, nd)


def test_argument_errors_are_reported_not_crashed(lib):
    """NULL pointers / bad sizes come back as status -1 with a message (no launch)."""
    out = np.zeros(4)
    rc = lib.cdll.obe_weight_sums(None, 10, None, 0, _lib.host_ptr(out), None)
    assert rc == -1 and lib.last_error()
    rc = lib.cdll.obe_moments(None, 0, 3, 0, None, 0, None, None, None, 0, None)
    assert rc == -1
    rc = lib.cdll.obe_cdf_search(None, 0, None, 0, None, None, 0, None)
    assert rc == -1
    with pytest.raises(_lib.ObeHipError):
        lib.call("obe_argmax", None, 0, None, None, None, 0, None)


def test_sharded_objects_decide_the_range_check_from_all_slices(lib):
    """How many sweeps (and all-gathers) an opt_setting() of a sharded object takes hangs on whether kappa can
    report "the fast form left its range", which hangs on the settings a lane owns, which hangs on the LENGTH of a
    slice: 1023 settings over two ranks are 512 + 511 settings, 2 and 1 per lane (obe_sweep_settings_per_lane_for,
    a host function).  Every rank must answer with the same figure — the largest over all slices."""
    import types
    from optbayesexpt_amd.dist import shard_bounds
    from optbayesexpt_amd.obe_base import OptBayesExpt
    per_lane = lib.cdll.obe_sweep_settings_per_lane_for
    assert (per_lane(512, 0), per_lane(511, 0)) == (2, 1)
    dm = models.lorentzian(1)
    assert dm.safe_sweep and dm.safe_sweep_min_spt == 2
    for n_settings, world in ((1023, 2), (4100, 4), (2047, 2), (65536, 8), (16384, 8), (7, 3)):
        answers = []
        for rank in range(world):
            b, e = shard_bounds(n_settings, rank, world)
            fake = types.SimpleNamespace(_s_begin=b, _s_end=e, _n_settings=n_settings, _mlib=lib, _device_model=dm,
                                         _shard=types.SimpleNamespace(rank=rank, world_size=world))
            fake._settings_per_lane = types.MethodType(OptBayesExpt._settings_per_lane, fake)
            answers.append((OptBayesExpt._settings_per_lane(fake), OptBayesExpt._sweep_needs_range_check(fake)))
        assert len(set(answers)) == 1, (n_settings, world, answers)
        lengths = [shard_bounds(n_settings, r, world) for r in range(world)]
        assert answers[0][0] == max(per_lane(e - b, 0) for b, e in lengths)
    # an unsharded object answers for its own grid
    fake = types.SimpleNamespace(_s_begin=0, _s_end=511, _n_settings=511, _mlib=lib, _device_model=dm, _shard=None)
    fake._settings_per_lane = types.MethodType(OptBayesExpt._settings_per_lane, fake)
    assert OptBayesExpt._settings_per_lane(fake) == 1 and not OptBayesExpt._sweep_needs_range_check(fake)
