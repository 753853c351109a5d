"""HIP path unit parity (GPU): every C-ABI family against the golden unit vectors, the
oracle on seeded random inputs, and the literal expectations of the reference's own
unit tests (same toy inputs).  Integer results exact, float64 results to 1e-10."""
import warnings

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import _replay
from _replay import assert_rel, conditioning_rtol
import oracle
from oracle import models as omodels

pytestmark = pytest.mark.gpu

RTOL = 1e-10


@pytest.fixture(scope="module")
def obe(hip):
    import optbayesexpt_amd
    return optbayesexpt_amd


@pytest.fixture(scope="module")
def unit():
    return _replay.load("unit_cases.npz")


# ----------------------------------------------------------------- K3 moments
@pytest.mark.parametrize("d", [1, 3, 10])
def test_moments_match_reference(obe, unit, d):
    x, w = unit[f"mom{d}_x"], unit[f"mom{d}_w"]
    pdf = obe.ParticlePDF(x)
    pdf.particle_weights = w
    assert_allclose(pdf.mean(), unit[f"mom{d}_mean"], rtol=1e-12)
    cov = pdf.covariance()
    assert cov.shape == (d, d)
    assert_allclose(cov, unit[f"mom{d}_cov"], rtol=1e-12, atol=1e-12 * np.abs(unit[f"mom{d}_cov"]).max())
    # std() is a one-pass formula: accuracy eps*<x^2>/var
    sd = unit[f"mom{d}_std"]
    tol = 1e-12 * sd + 64 * 2.3e-16 * unit[f"mom{d}_mean"] ** 2 / sd
    assert np.all(np.abs(pdf.std() - sd) <= tol)


def test_moments_large_ragged(obe):
    g = np.random.default_rng(5)
    for n in (1, 63, 64, 65, 255, 257, 100003):
        x = g.normal(3, 2, (4, n))
        w = g.exponential(1, n)
        w /= w.sum()
        pdf = obe.ParticlePDF(x)
        pdf.particle_weights = w
        assert_allclose(pdf.mean(), oracle.weighted_mean(x, w), rtol=1e-12)
        if n > 1:
            ref = oracle.weighted_covariance(x, w)
            assert_allclose(pdf.covariance(), ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("d", [1, 2, 7, 16])
def test_moments_every_dimension_class(obe, d):
    """The block partials are reduced by a wave reduce-scatter in groups of up to 64 values and
    folded by one 16-wave workgroup: 2 + 2 D values in pass 1, D (D + 1) / 2 in pass 2 — 4 ... 136,
    i.e. one partly filled group up to three groups.  Against the oracle, several workgroups."""
    g = np.random.default_rng(40 + d)
    n = 300007
    x = g.normal(0.0, 1.0, (d, n)) * g.uniform(0.5, 20.0, (d, 1)) + g.normal(0.0, 5.0, (d, 1))
    w = g.exponential(1.0, n) ** 2
    w /= w.sum()
    pdf = obe.ParticlePDF(x)
    pdf.particle_weights = w
    assert_allclose(pdf.mean(), oracle.weighted_mean(x, w), rtol=1e-12)
    ref = oracle.weighted_covariance(x, w).reshape(d, d)
    assert_allclose(pdf.covariance(), ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
    assert_allclose(pdf.std(), np.sqrt(np.maximum(np.sum(w * x * x, axis=1) - np.sum(w * x, axis=1) ** 2, 0.0)),
                    rtol=1e-9)


def test_deferred_host_results_at_the_abi(obe, hip):
    """obe_defer_host_sync: moments and the CDF total land in page-locked memory without a stream
    synchronisation inside the call; after the caller's own synchronisation they equal what the
    synchronous calls deliver."""
    import ctypes
    import torch
    from optbayesexpt_amd import _lib
    P = ctypes.c_void_p
    g = np.random.default_rng(9)
    d, n = 4, 123457
    x = torch.from_numpy(g.normal(1.0, 2.0, (d, n))).cuda()
    wn = g.exponential(1.0, n)
    w = torch.from_numpy(wn / wn.sum()).cuda()
    mlen = hip.moments_len(d)
    out = torch.zeros(mlen, dtype=torch.float64, device="cuda")
    cdf = torch.empty(n, dtype=torch.float64, device="cuda")
    ws = torch.empty(hip.workspace_bytes(n, 1, 1, d) // 8 + 1, dtype=torch.float64, device="cuda")
    st = P(torch.cuda.current_stream().cuda_stream)
    sync_m, sync_t = np.zeros(mlen), np.zeros(1)
    hip.call("obe_moments", P(x.data_ptr()), n, d, n, P(w.data_ptr()), 1, P(out.data_ptr()), _lib.host_ptr(sync_m),
             P(ws.data_ptr()), ws.numel() * 8, st)
    hip.call("obe_weight_cdf", P(w.data_ptr()), n, 0, P(cdf.data_ptr()), _lib.host_ptr(sync_t), P(ws.data_ptr()),
             ws.numel() * 8, st)
    pinned = torch.full((mlen + 1,), -7.0, dtype=torch.float64).pin_memory()
    assert hip.cdll.obe_defer_host_sync(1) == 0
    try:
        hip.call("obe_moments", P(x.data_ptr()), n, d, n, P(w.data_ptr()), 1, P(out.data_ptr()),
                 P(pinned.data_ptr() + 8), P(ws.data_ptr()), ws.numel() * 8, st)
        hip.call("obe_weight_cdf", P(w.data_ptr()), n, 0, P(cdf.data_ptr()), P(pinned.data_ptr()), P(ws.data_ptr()),
                 ws.numel() * 8, st)
    finally:
        assert hip.cdll.obe_defer_host_sync(0) == 1
    torch.cuda.synchronize()
    assert_array_equal(pinned[1:].numpy(), sync_m)
    assert pinned[0].item() == sync_t[0] and abs(sync_t[0] - 1.0) < 1e-12
    assert hip.cdll.obe_ziggurat_check(100, 50, 50, 1000, 0) == 0          # enough raw values
    assert hip.cdll.obe_ziggurat_check(100, 49, 50, 1000, 0) == 1          # too few normals found
    assert hip.cdll.obe_ziggurat_check(990, 50, 50, 1000, 0) == 1          # ended too close to the buffer's end


# ------------------------------------------------------- K4 draws and resample
@pytest.mark.parametrize("tag,scale", [("s0", False), ("s1", True)])
@pytest.mark.parametrize("strict", [False, True])
def test_randdraw_and_resample_match_reference(obe, unit, tag, scale, strict):
    pdf = obe.ParticlePDF(unit[f"rs_{tag}_x"].copy(), scale=scale)
    pdf.tuning_parameters["strict_cdf"] = strict
    pdf.particle_weights = unit[f"rs_{tag}_w"].copy()
    pdf.rng = np.random.default_rng(4242)
    draws = pdf.randdraw(30)
    assert_array_equal(pdf.last_draw_indices, unit[f"rs_{tag}_draw_idx"])       # bit-exact indices
    assert_array_equal(draws, unit[f"rs_{tag}_draws"])
    pdf.resample()
    assert_array_equal(pdf.last_draw_indices, unit[f"rs_{tag}_resample_idx"])   # bit-exact indices
    ref = unit[f"rs_{tag}_particles"]
    got = pdf.particles
    assert got.shape == ref.shape
    for i in range(ref.shape[0]):
        assert_allclose(got[i], ref[i], rtol=RTOL)
    assert_array_equal(pdf.particle_weights, unit[f"rs_{tag}_weights"])


@pytest.mark.parametrize("n", [65536, 65537, 300001])
def test_small_draws_on_large_clouds(obe, n):
    """randdraw(30)-sized draws go through obe_draw_indices: one launch up to 65 536 particles, the
    three-kernel scan + by-value search beyond; fresh-CDF draws skip the scan.  Indices against
    the oracle (strict CDF: exact by construction; blocked scan: exact on these seeds), and an
    invalid weight vector raises numpy's ValueError with the generator left where it was."""
    g = np.random.default_rng(n)
    x = g.normal(size=(2, n))
    w = g.exponential(1.0, n) ** 2
    w /= w.sum()
    for strict in (True, False):
        pdf = obe.ParticlePDF(x.copy())
        pdf.tuning_parameters["strict_cdf"] = strict
        pdf.particle_weights = w.copy()
        pdf.rng = np.random.default_rng(2)
        state = pdf.rng.bit_generator.state
        empty = pdf.randdraw(0)              # rng.choice(size=0): an (n_dims, 0) array, nothing consumed
        assert empty.shape == (2, 0) and pdf.rng.bit_generator.state == state
        pdf.rng = np.random.default_rng(11)
        ref = np.random.default_rng(11)
        for n_draws in (30, 1, 64):                      # the second and third draw reuse the CDF
            got = pdf.randdraw(n_draws)
            idx = oracle.choice_indices(w, ref.random(n_draws))
            assert_array_equal(pdf.last_draw_indices, idx)
            assert_array_equal(got, x[:, idx])
        assert pdf.rng.bit_generator.state == ref.bit_generator.state
    bad = w.copy()
    bad[5] = np.nan
    pdf.particle_weights = bad
    state = pdf.rng.bit_generator.state
    with pytest.raises(ValueError, match="NaN"):
        pdf.randdraw(30)
    assert pdf.rng.bit_generator.state == state
    pdf.particle_weights = 2.0 * w
    with pytest.raises(ValueError, match="sum to 1"):
        pdf.randdraw(30)
    assert pdf.rng.bit_generator.state == state


def test_systematic_resampling_extension(obe):
    """tuning_parameters['resample_method'] = 'systematic' (the scheme BASELINE.json's north_star
    names; the reference is multinomial): one uniform, indices at the stratified CDF points —
    against the oracle's restatement, plus the defining property of the scheme: particle i is
    copied floor(N w_i) or ceil(N w_i) times."""
    g = np.random.default_rng(99)
    for n, scale in ((5000, False), (70001, True)):
        x = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
        w = g.exponential(1.0, n) ** 2
        w /= w.sum()
        a = obe.ParticlePDF(x.copy(), scale=scale)
        b = oracle.OracleParticlePDF(x.copy(), scale=scale)
        for o in (a, b):
            o.tuning_parameters["resample_method"] = "systematic"
            o.particle_weights = w.copy()
            o.rng = np.random.default_rng(7)
        a.tuning_parameters["strict_cdf"] = True              # same CDF bits as the oracle's np.cumsum
        a.resample()
        b.resample()
        idx = a.last_resample_indices_device.cpu().numpy()
        assert_array_equal(idx, b.last_draw_indices)
        counts = np.bincount(idx, minlength=n)
        assert np.all(counts >= np.floor(n * w - 1e-9)) and np.all(counts <= np.ceil(n * w + 1e-9))
        assert np.all(np.diff(idx) >= 0)
        assert_allclose(a.particles, b.particles, rtol=1e-10, atol=1e-9)
        assert_allclose(a.particle_weights, 1.0 / n)
        assert a.rng.bit_generator.state == b.rng.bit_generator.state or n > 30000   # device normals: same count
    with pytest.raises(ValueError):
        a.tuning_parameters["resample_method"] = "stratified"
        a.particle_weights = w.copy()
        a.resample()


def test_strict_cdf_is_bitwise_numpy_cumsum(obe, hip):
    """tuning_parameters['strict_cdf']: the device CDF equals np.cumsum(w)/cumsum[-1] bit
    for bit; the parallel scan agrees to ~1e-13 and gives the same indices."""
    import torch
    from optbayesexpt_amd import _lib
    from optbayesexpt_amd.particlepdf import _ptr
    g = np.random.default_rng(11)
    for n in (1, 5, 64, 1000, 2048, 2049, 300001):
        w = g.exponential(1.0, n) ** 2
        w /= w.sum()
        ref = oracle.weight_cdf(w)
        wd = torch.from_numpy(w).cuda()
        ws = torch.zeros(hip.workspace_bytes(n, 1, 1, 1) // 8 + 1, dtype=torch.float64, device="cuda")
        out = {}
        for strict in (1, 0):
            cdf = torch.empty(n, dtype=torch.float64, device="cuda")
            tot = np.zeros(1)
            hip.call("obe_weight_cdf", _ptr(wd), n, strict, _ptr(cdf), _lib.host_ptr(tot), _ptr(ws),
                     ws.numel() * 8, None)
            out[strict] = cdf.cpu().numpy()
        assert_array_equal(out[1], ref)
        assert_allclose(out[0], ref, rtol=1e-12)
        assert out[0][-1] == 1.0
        u = g.random(4096)
        ud = torch.from_numpy(u).cuda()
        for strict in (1, 0):
            idx = torch.empty(u.size, dtype=torch.int64, device="cuda")
            cdf = torch.from_numpy(out[strict]).cuda()
            hip.call("obe_cdf_search", _ptr(cdf), n, _ptr(ud), u.size, _ptr(idx), None, 0, None)
            assert_array_equal(idx.cpu().numpy(), ref.searchsorted(u, side="right"))


@pytest.mark.parametrize("kind", ["uniform", "one_heavy", "many_zeros", "steps", "tiny_tail"])
def test_guided_cdf_search_gives_searchsorted_indices(hip, kind):
    """The guide-table search of a resample-sized draw (N uniforms into an N-entry CDF): exactly
    numpy's searchsorted(side='right') whatever the CDF looks like — one particle carrying almost all
    the weight (one bucket chain spans most of the table), long runs of equal entries (zero weights),
    coarse steps, uniforms on bucket edges and on CDF entries themselves."""
    import torch
    from optbayesexpt_amd.particlepdf import _ptr
    g = np.random.default_rng(hash(kind) % 1000)
    n = 200003
    w = {"uniform": lambda: np.ones(n),
         "one_heavy": lambda: np.where(np.arange(n) == n // 3, 1e7, g.exponential(1.0, n)),
         "many_zeros": lambda: np.where(g.random(n) < 0.9, 0.0, g.exponential(1.0, n)),
         "steps": lambda: np.where(np.arange(n) % 1000 == 0, 1.0, 0.0),
         "tiny_tail": lambda: np.concatenate([g.exponential(1.0, n // 2), np.full(n - n // 2, 1e-300)])}[kind]()
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    u = g.random(n)
    u[:2000] = (np.arange(2000) * 97 % n) / n                      # bucket edges b / n
    u[2000:4000] = np.minimum(cdf[g.integers(0, n, 2000)], np.nextafter(1.0, 0.0))      # CDF entries themselves
    u[4000:4004] = [0.0, np.nextafter(1.0, 0.0), 5e-324, 0.5]
    cd, ud = torch.from_numpy(cdf).cuda(), torch.from_numpy(u).cuda()
    ws = torch.empty(n + 16, dtype=torch.float64, device="cuda")
    out = {}
    for guided in (False, True):
        idx = torch.full((n,), -1, dtype=torch.int64, device="cuda")
        hip.call("obe_cdf_search", _ptr(cd), n, _ptr(ud), n, _ptr(idx), _ptr(ws) if guided else None,
                 ws.numel() * 8 if guided else 0, None)
        out[guided] = idx.cpu().numpy()
    ref = cdf.searchsorted(u, side="right")
    assert_array_equal(out[False], ref)
    assert_array_equal(out[True], ref)


# ------------------------------------------------------------- K2 Bayes update
def test_bayes_update_nan_inf_and_zero_likelihood(obe, unit):
    n = unit["bu_w"].size
    pdf = obe.ParticlePDF(np.zeros((1, n)), auto_resample=False)
    pdf.particle_weights = unit["bu_w"].copy()
    pdf.bayesian_update(unit["bu_lik"])
    assert_allclose(pdf.particle_weights, unit["bu_out"], rtol=1e-12)
    assert_array_equal(pdf.particle_weights == 0, unit["bu_out"] == 0)           # NaN -> 0
    pdf.particle_weights = unit["bu_w"].copy()
    pdf.bayesian_update(np.zeros(n))                                               # 0/0 -> zeros
    assert not np.any(pdf.particle_weights)
    pdf.resample_test()
    assert np.isinf(pdf.last_n_eff) and pdf.just_resampled is False
    lik = np.ones(n)
    lik[7] = np.inf                                                                # inf -> DBL_MAX
    pdf.particle_weights = unit["bu_w"].copy()
    pdf.bayesian_update(lik)
    with np.errstate(all="ignore"):
        assert_allclose(pdf.particle_weights, oracle.normalized_product(unit["bu_w"], lik), rtol=1e-12)


def test_infer_matches_reference_and_analytic(obe, unit):
    """reference tests/test_zinference.py:89-108."""
    xs = unit["infer_x"]
    n = xs.size

    class MyObe(obe.OptBayesExpt):
        def enforce_parameter_constraints(self):
            bad = np.argwhere(self.parameters[1] < 0)
            for index in bad:
                self.particle_weights[index] = 0
            self.particle_weights = self.particle_weights / np.sum(self.particle_weights)

    o = MyObe(obe.models.first_parameter(), (0,), (xs, np.ones(n)), (0,))
    o.tuning_parameters["resample_threshold"] = 0
    o.pdf_update(((), 1.0, 1.0))
    known = np.exp(-(1.0 - xs) ** 2 / 2) / np.sqrt(2 * np.pi)
    known /= np.sum(known)
    assert_allclose(o.particle_weights, known, atol=1e-15, rtol=1e-15)    # the reference's own bar
    assert_allclose(o.particle_weights, unit["infer_w"], rtol=1e-12, atol=1e-300)
    # sigma passed as an (N_p,) array: only element 0 is consumed (zip truncation)
    o2 = MyObe(obe.models.first_parameter(), (0,), (xs, np.ones(n)), (0,))
    o2.tuning_parameters["resample_threshold"] = 0
    o2.pdf_update(((), 1.0, o2.parameters[1]))
    assert_allclose(o2.particle_weights, unit["infer_w"], rtol=1e-12, atol=1e-300)


def test_likelihood_two_channel_choke(obe, unit):
    n = unit["lk_ym"].shape[1]
    o = obe.OptBayesExpt(obe.models.coil(), (np.logspace(4, 6, 5),), np.ones((4, n)), (), choke=0.6)
    got = o.likelihood(unit["lk_ym"], ((1.0,), (0.3, -0.2), (1.5, 0.7)))
    assert_allclose(got, unit["lk_out"], rtol=1e-12)


# ------------------------------------- the reference's own unit tests, restated
def _toy_pdf(obe):
    return obe.ParticlePDF((np.array([0, 1, 2, 3]), np.array([1, 3, 2, 4])))


def test_reference_particlepdf_literals(obe):
    """reference tests/test_particlepdf.py:17-152, same inputs and expected values."""
    pdf = _toy_pdf(obe)
    assert (pdf.n_dims, pdf.n_particles) == (2, 4)
    assert_array_equal(np.asarray([[0, 1, 2, 3], [1, 3, 2, 4]]), pdf.particles)
    assert_array_equal([.25, .25, .25, .25], pdf.particle_weights)
    assert pdf.just_resampled is False
    assert pdf.mean().shape == (2,)
    assert_allclose(pdf.mean(), (1.5, 2.5), rtol=1e-15)
    assert_allclose(pdf.covariance(), [[5 / 3, 4 / 3], [4 / 3, 5 / 3]])
    assert_allclose(pdf.std(), np.sqrt(np.array([1, 1]) * 5.0 / 4.0), rtol=1e-15)
    samples = np.arange(15).reshape((3, 5))
    pdf.set_pdf(samples)
    assert (pdf.n_dims, pdf.n_particles) == (3, 5)
    assert_array_equal(samples, pdf.particles)
    assert_array_equal(np.ones(5) / 5.0, pdf.particle_weights)
    pdf.set_pdf(samples, weights=np.array([1, 2, 3, 4, 5]))
    assert_allclose(pdf.particle_weights, np.array([1, 2, 3, 4, 5]) / 15, rtol=1e-15)
    with pytest.raises(ValueError):
        pdf.set_pdf(samples, weights=np.ones(4))
    pdf = _toy_pdf(obe)
    pdf.tuning_parameters["auto_resample"] = False
    lik = np.array([.5, 1.5, 1.5, .5])
    pdf.bayesian_update(lik)
    assert_allclose(pdf.particle_weights, lik / np.sum(lik), rtol=1e-15)
    pdf = _toy_pdf(obe)
    pdf.particle_weights = np.array([0, .5, .5, 0])
    pdf.resample()
    assert pdf.particles.shape == (2, 4)
    assert_array_equal([.25, .25, .25, .25], pdf.particle_weights)
    pdf = _toy_pdf(obe)
    pdf.particle_weights = np.array([.1, .4, .4, .1])      # N_eff 2.94
    pdf.resample_test()
    assert pdf.just_resampled is False
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        pdf.particle_weights = np.array([0, .75, .25, 0])  # N_eff 1.6
        pdf.resample_test()
    assert pdf.just_resampled is True


def test_reference_optbayesexpt_literals(obe):
    """reference tests/test_optbayesexpt.py:21-69."""
    pars = (np.array([0, 1, 2, 3]), np.array([1, 3, 2, 4]))
    o = obe.OptBayesExpt(obe.models.line_ab(), (np.array([0, 1, 2]),), pars, ())
    assert_array_equal((np.array([0, 1, 2]),), o.allsettings)
    assert_array_equal(pars, o.parameters)
    assert_array_equal([[1, 4, 4, 7]], o.eval_over_all_parameters((1,)))
    assert_array_equal([[1, 4, 7]], o.eval_over_all_settings([1, 3]))
    ymodel = np.array(((1, 4, 4, 7),))
    assert_allclose(o.likelihood(ymodel, ((1,), (5.0,), 1.0)), np.exp(-(ymodel - 5.0) ** 2 / 2)[0], rtol=1e-15)
    lkl = np.exp(-(np.array((1, 4, 4, 7)) - 5.0) ** 2 / 2)
    parts, wts = o.pdf_update(((1,), 5.0, 1.0))
    # the reference's own assertion, verbatim (tests/test_optbayesexpt.py:66-69): assert_array_equal — the update's
    # sum is formed in np.sum's order on a cloud this small (tuning_parameters['strict_sums'])
    assert_array_equal(lkl / np.sum(lkl), o.particle_weights, err_msg="incorrect updated weights")
    assert_array_equal(parts, o.particles)
    with pytest.raises(SyntaxError):
        obe.OptBayesExpt(obe.models.line_ab(), (np.array([0, 1, 2]),), pars, (), utility_method="nope")
    with pytest.raises(SyntaxError):
        obe.OptBayesExpt(obe.models.line_ab(), (np.array([0, 1, 2]),), pars, (), selection_method="nope")
    with pytest.raises(RuntimeError):
        obe.OptBayesExptNoiseParameter(obe.models.line_ab(), (np.array([0, 1, 2]),), pars, (),
                                       noise_parameter_index=(0, 1))


def test_strict_sums_are_np_sum_bit_for_bit(obe, hip):
    """tuning_parameters['strict_sums'] / obe_strict_sums: the update's sum t and the resample test's sum
    nan_to_num(w'^2) in the ORDER np.sum adds (oracle.numpy_pairwise_sum, pinned against np.sum on the CPU) — through
    the C ABI for lengths on every branch of that order (< 8, <= 128, the 8-aligned splits, the 8192-element pieces)
    and through the classes: weights == nan_to_num(t / np.sum(t)) and N_eff == 1 / np.sum(w^2) as BITS
    (particlepdf.py:136-139, 243-244)."""
    import torch
    from optbayesexpt_amd import _lib
    from optbayesexpt_amd.particlepdf import _ptr
    g = np.random.default_rng(77)
    out = _lib.pinned_array(4)
    lengths = [1, 2, 7, 8, 9, 15, 16, 17, 127, 128, 129, 136, 143, 255, 256, 257, 1000, 2047, 4095, 4096, 4097, 4103,
               4104, 4105, 5000, 8191, 8192, 8193, 16391, 70001] + [int(v) for v in g.integers(1, 9000, 40)]
    assert hip.cdll.obe_strict_sums(1) == 0
    try:
        for n in lengths:
            w = g.exponential(1.0, n) * 10.0 ** g.integers(-6, 6, n)
            w /= w.sum()
            lik = np.exp(-g.exponential(3.0, n) ** 2)
            if n > 20:
                lik[g.integers(0, n, 3)] = [np.nan, 0.0, np.inf][:3]          # nan_to_num on the way
            wd, ld = torch.from_numpy(w.copy()).cuda(), torch.from_numpy(lik).cuda()
            ws = torch.empty(hip.workspace_bytes(n, 1, 1, 1) // 8 + 1, dtype=torch.float64, device="cuda")
            hip.call("obe_bayes_update_lik", _ptr(ld), n, _ptr(wd), _ptr(ws), ws.numel() * 8, _lib.host_ptr(out), None)
            with np.errstate(all="ignore"):
                t = np.nan_to_num(w * lik)
                want = np.nan_to_num(t / np.sum(t))
                want_w2 = np.sum(np.nan_to_num(want * want))
            assert np.float64(out[0]).tobytes() == np.float64(np.sum(t)).tobytes(), (n, out[0], np.sum(t))
            assert_array_equal(wd.cpu().numpy(), want, err_msg=f"n = {n}")
            assert np.float64(out[1]).tobytes() == np.float64(want_w2).tobytes(), (n, out[1], want_w2)
            assert np.float64(np.sum(t)).tobytes() == oracle.numpy_pairwise_sum(t).tobytes()
    finally:
        assert hip.cdll.obe_strict_sums(0) == 1
    # through the classes: 'auto' is on up to 4096 particles, off above; forced either way
    for n, mode, strict in ((3000, "auto", True), (5000, "auto", False), (5000, True, True), (3000, False, False)):
        x = g.normal(size=(2, n))
        pdf = obe.ParticlePDF(x, auto_resample=False)
        pdf.tuning_parameters["strict_sums"] = mode
        assert pdf._strict_sums() is strict
        lik = np.exp(-0.5 * (x[0] - 0.3) ** 2 / 0.01)
        pdf.bayesian_update(lik)
        t = np.nan_to_num(np.ones(n) / n * lik)
        same = np.array_equal(pdf.particle_weights, np.nan_to_num(t / np.sum(t)))
        assert same or not strict, (n, mode)
        assert_allclose(pdf.particle_weights, t / np.sum(t), rtol=1e-14)
        if strict:                         # what resample_test() (particlepdf.py:243-244) decides from
            wv = np.array(pdf.particle_weights)
            assert pdf._sum_w2() == np.sum(np.nan_to_num(wv * wv))
    assert hip.cdll.obe_strict_sums(-1) == 0               # (every class call leaves the thread's switch off)


# --------------------------------------------------- K1 full sweep (D1-ii) parity
def _full_cases(obe, f):
    m = obe.models
    return [("lor", m.lorentzian(), omodels.lorentzian, (f["fs_lor_x"],), (0.1,), None, 500.0),
            ("ml7", m.lorentzian(7), omodels.multi_lorentzian(7), (f["fs_lor_x"],), (0.1,), 9, None),
            ("coil", m.coil(), omodels.coil, (f["fs_coil_w"],), (), (3, 3), None),
            ("rabi", m.rabi(), omodels.rabi, (f["fs_rabi_s0"], f["fs_rabi_s1"]),
             (100000.0, 0.01, 2.0), None, 300.0)]


def _make(obe, dm, sv, prior, cons, noise_idx, noise_std, **kw):
    if noise_idx is None:
        return obe.OptBayesExpt(dm, sv, prior.copy(), cons, utility_method="variance_full",
                                default_noise_std=noise_std, **kw)
    return obe.OptBayesExptNoiseParameter(dm, sv, prior.copy(), cons, utility_method="variance_full",
                                          noise_parameter_index=noise_idx, **kw)


def test_full_sweep_uniform_matches_reference(obe):
    """Reference driven with every particle as a draw (golden) vs the HIP full sweep."""
    f = _replay.load("full_sweep_uniform.npz")
    for tag, dm, _, sv, cons, nidx, nstd in _full_cases(obe, f):
        o = _make(obe, dm, sv, f[f"fs_{tag}_prior"], cons, nidx, nstd)
        yvar = o.yvar_from_parameter_draws()
        ref = f[f"fs_{tag}_yvar"]
        assert yvar.shape == ref.shape
        assert_rel(yvar, ref, RTOL, f"{tag} variance")
        util = o.utility()
        assert_rel(util, f[f"fs_{tag}_utility"], RTOL, f"{tag} utility")
        o.opt_setting()
        assert o.last_setting_index == int(np.argmax(f[f"fs_{tag}_utility"]))


def test_full_sweep_nonuniform_weights_matches_oracle(obe):
    """After real updates the weights are non-uniform: HIP weighted variance vs the
    oracle's two-pass weighted variance (not expressible with the reference itself)."""
    f = _replay.load("full_sweep_uniform.npz")
    g = np.random.default_rng(99)
    for tag, dm, fn, sv, cons, nidx, nstd in _full_cases(obe, f):
        prior = f[f"fs_{tag}_prior"]
        o = _make(obe, dm, sv, prior, cons, nidx, nstd, auto_resample=False)
        w = g.exponential(1.0, prior.shape[1]) ** 2
        w /= w.sum()
        o.particle_weights = w
        yvar = o.yvar_from_parameter_draws()
        ref = oracle.yvar_full_sweep(fn, oracle.flatten_settings(sv), prior, w, cons, chunk=512)
        assert_rel(yvar, ref, RTOL, f"{tag} variance, non-uniform weights")


@pytest.mark.parametrize("shift", ["always", "never", "auto"])
def test_weighted_full_sweep_matches_the_reference_with_integer_multiplicities(obe, shift):
    """the weighted full sweep — the mode every BASELINE config runs in — against the REAL
    reference.  The fixture drove the reference with randdraw() returning particle i duplicated k_i times
    (k in 0..7, zeros included): its np.var over the sum(k) draws (obe_base.py:463-489) is the weighted
    variance with w = k / sum(k) exactly.  HIP `variance_full` with those weights, both variance-shift
    variants (and what the adaptive policy picks), 1e-10; the chosen setting exact."""
    f = _replay.load("full_sweep_integer_weights.npz")
    m = obe.models
    cases = [("lor", m.lorentzian(), (f["iw_x48"],), (0.1,), None, 500.0),
             ("lornarrow", m.lorentzian(), (f["iw_x48"],), (0.1,), None, 500.0),
             ("ml7", m.lorentzian(7), (f["iw_x48"],), (0.1,), 9, None),
             ("coil", m.coil(), (f["iw_coil_w"],), (), (3, 3), None),
             ("rabi", m.rabi(), (f["iw_rabi_s0"], f["iw_rabi_s1"]), (100000.0, 0.01, 2.0), None, 300.0)]
    for tag, dm, sv, cons, nidx, nstd in cases:
        if shift == "never" and tag == "lornarrow":
            continue          # kappa ~ 1e5: the unshifted one-pass variance is not meant for it (the policy never picks it)
        k = f[f"iw_{tag}_k"]
        o = _make(obe, dm, sv, f[f"iw_{tag}_prior"], cons, nidx, nstd, auto_resample=False)
        o.tuning_parameters["sweep_shift"] = shift
        o.particle_weights = k / k.sum()
        yvar = o.yvar_from_parameter_draws()
        ref = f[f"iw_{tag}_yvar"]
        assert yvar.shape == ref.shape
        assert_rel(yvar, ref, RTOL, f"{tag} {shift} variance")
        want = f[f"iw_{tag}_utility"]
        assert_rel(o.utility(), want, RTOL, f"{tag} {shift} utility")
        o.opt_setting()
        assert o.last_setting_index == int(np.argmax(want))
        if shift == "auto" and tag == "lornarrow":
            assert o.last_sweep["shifted"] and o.last_sweep["kappa"] > obe.OptBayesExpt.KAPPA_ENTER


def test_sweep_shapes_ragged(obe):
    """Setting counts around every tile boundary, particle counts around chunk/tile
    boundaries, draws mode and full mode; against the oracle."""
    g = np.random.default_rng(3)
    # (the last five: particle counts around the per-wave quarter of a chunk and the 4-particle
    # prefetch group of the scalar-path sweep kernel, at 8 / 2 / 1 settings per lane)
    for ns, n in [(1, 1), (1, 300), (63, 64), (257, 1000), (1025, 2049), (4099, 513),
                  (4099, 2), (4099, 7), (4100, 21), (520, 4099), (130, 16389)]:
        prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
        sv = (np.linspace(1.5, 4.5, ns),)
        w = g.exponential(1.0, n)
        w /= w.sum()
        o = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior.copy(), (0.1,),
                             utility_method="variance_full", auto_resample=False, default_noise_std=7.0)
        o.particle_weights = w
        ref = oracle.yvar_full_sweep(omodels.lorentzian, oracle.flatten_settings(sv), prior, w, (0.1,))
        # (floor: n = 1 — one particle is every draw; the reference variance is rounding debris, see assert_rel)
        assert_rel(o.yvar_from_parameter_draws(), ref, RTOL, f"full sweep {ns} x {n}", garbage_floor=1e-18 if n == 1 else 0.0)
        util = oracle.utility_from_yvar(ref, 49.0, 1.0)
        o.opt_setting()
        assert o.last_setting_index == int(np.argmax(util))
        # reference-semantics mode on the same cloud
        o2 = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior.copy(), (0.1,), n_draws=30,
                              auto_resample=False)
        o2.particle_weights = w
        o2.rng = np.random.default_rng(8)
        got = o2.yvar_from_parameter_draws()
        idx = o2.last_draw_indices
        ref2 = oracle.yvar_from_draws(omodels.lorentzian, oracle.flatten_settings(sv), prior[:, idx], (0.1,))
        assert_array_equal(idx, oracle.choice_indices(w, np.random.default_rng(8).random(30)))
        # (floor: 30 draws from 1, 2 or 7 particles are often all the same particle: (eps*y)^2 debris in the reference)
        assert_rel(got, ref2, RTOL, f"30 draws, {ns} x {n}", garbage_floor=1e-18 if n <= 7 else 0.0)


@pytest.mark.parametrize("ns,n,nd", [(7, 50, 1), (40, 5, 256), (500, 1000, 257), (4500, 2049, 30), (1, 3, 2),
                                      (513, 70000, 255)])
def test_draws_mode_on_both_sides_of_the_one_workgroup_limit(obe, ns, n, nd):
    """Reference-semantics sweeps: the one-workgroup kernel serves N_s * N_d <= 131072 with
    N_d <= 256, the tiled kernels everything else; draw indices exact, variances and the chosen
    setting against the oracle."""
    g = np.random.default_rng(ns * 1000 + nd)
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    sv = (np.linspace(1.5, 4.5, ns),)
    w = g.exponential(1.0, n)
    w /= w.sum()
    o = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior.copy(), (0.1,), n_draws=nd, auto_resample=False,
                         default_noise_std=3.0)
    o.particle_weights = w
    o.rng = np.random.default_rng(8)
    x = o.opt_setting()
    idx = o.last_draw_indices
    assert_array_equal(idx, oracle.choice_indices(w, np.random.default_rng(8).random(nd)))
    ref = oracle.yvar_from_draws(omodels.lorentzian, oracle.flatten_settings(sv), prior[:, idx], (0.1,))
    # (floor: one draw, or two draws of three particles that coincide — identical draws, see assert_rel)
    floor = 1e-18 if nd <= 2 else 0.0
    assert_rel(o._yvar_dev.cpu().numpy(), ref, RTOL, f"{ns} x {nd} draws of {n}", garbage_floor=floor)
    util = oracle.utility_from_yvar(ref, 9.0, 1.0)
    got_u = o._utility_dev.cpu().numpy()
    assert_rel(got_u, util, RTOL, f"{ns} x {nd} draws of {n}: utility", garbage_floor=floor / 9.0)
    assert got_u[o.last_setting_index] == got_u.max() and x == (sv[0][o.last_setting_index],)


def test_argmax_semantics(obe, hip):
    """np.argmax: first maximum wins, NaN beats everything."""
    import torch
    from optbayesexpt_amd import _lib
    from optbayesexpt_amd.particlepdf import _ptr
    g = np.random.default_rng(1)
    ws = torch.zeros(hip.workspace_bytes(1, 1 << 20, 1, 1) // 8 + 1, dtype=torch.float64, device="cuda")
    cases = []
    for n in (1, 2, 255, 256, 257, 70001):
        v = g.normal(size=n)
        cases.append(v)
        t = v.copy()
        t[[0, n // 2, n - 1]] = v.max() + 1          # ties
        cases.append(t)
        q = v.copy()
        q[n // 3] = np.nan
        q[n - 1] = np.nan
        cases.append(q)
        cases.append(np.full(n, -np.inf))
    for v in cases:
        best, idx = np.zeros(1), np.zeros(1, dtype=np.int64)
        hip.call("obe_argmax", _ptr(torch.from_numpy(v).cuda()), v.size, _lib.host_ptr(best),
                 _lib.host_ptr(idx), _ptr(ws), ws.numel() * 8, None)
        assert int(idx[0]) == int(np.argmax(v))


# ----------------------------------------------------------- hooks and mirrors
def test_hooks_cost_array_and_inplace_weight_writes(obe):
    """demos/lockin/lockin_of_coil.py:115-152 pattern: a subclass that overrides
    cost_estimate() with a per-setting array and zeroes weights in place."""
    g = np.random.default_rng(21)
    n, ns = 3000, 97
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    sv = (np.linspace(1.5, 4.5, ns),)

    class Sub(obe.OptBayesExpt):
        def cost_estimate(self):
            cost = np.ones_like(self.allsettings[0]) * 3.0
            cost[self.last_setting_index] = 1.0
            return cost

        def enforce_parameter_constraints(self):
            bad = np.argwhere(self.parameters[0] < 2.5).flatten()
            for i in bad:
                self.particle_weights[i] = 0
            self.particle_weights = self.particle_weights / np.sum(self.particle_weights)

    class OSub(oracle.OracleOptBayesExpt):
        cost_estimate = Sub.cost_estimate
        enforce_parameter_constraints = Sub.enforce_parameter_constraints

    a = Sub(obe.models.lorentzian(), sv, prior.copy(), (0.1,), scale=False, default_noise_std=100.0)
    b = OSub(omodels.lorentzian, sv, prior.copy(), (0.1,), scale=False, default_noise_std=100.0)
    a.rng, b.rng = np.random.default_rng(5), np.random.default_rng(5)
    sim = np.random.default_rng(6)
    resamples = 0
    for cyc in range(25):
        xa, xb = a.opt_setting(), b.opt_setting()
        assert a.last_setting_index == b.last_setting_index
        y = float(omodels.lorentzian(xb, (3.0, -1000.0, 50000.0), (0.1,)) + 100 * sim.standard_normal())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            a.pdf_update((xa, y, 100.0))
            b.pdf_update((xb, y, 100.0))
        assert a.just_resampled == b.just_resampled
        resamples += a.just_resampled
        assert_allclose(a.particle_weights, b.particle_weights, rtol=RTOL, atol=1e-13 * b.particle_weights.max())
    assert resamples >= 2
    assert np.sum(a.particle_weights == 0) == np.sum(b.particle_weights == 0)


def test_set_pdf_keeps_stale_parameters_alias(obe):
    """set_pdf() re-initialises particles but not ``parameters`` (obe_base.py:185,395):
    the next pdf_update evaluates the model on the old samples, as in the reference."""
    g = np.random.default_rng(2)
    n = 500
    p1 = np.array([g.uniform(-1, 1, n), g.uniform(-1, 1, n)])
    p2 = np.array([g.uniform(-1, 1, n), g.uniform(-1, 1, n)])
    sv = (np.linspace(0, 1, 11),)
    a = obe.OptBayesExpt(obe.models.line_ab(), sv, p1.copy(), (), auto_resample=False)
    b = oracle.OracleOptBayesExpt(omodels.line_ab, sv, p1.copy(), (), auto_resample=False)
    for o in (a, b):
        o.set_pdf(p2.copy())
        o.pdf_update(((0.5,), 0.2, 0.3))
    assert_array_equal(a.particles, p2)
    assert_allclose(a.particle_weights, b.particle_weights, rtol=RTOL)
    for o in (a, b):
        o.pdf_update(((0.25,), 0.1, 0.3))
    assert_allclose(a.particle_weights, b.particle_weights, rtol=RTOL)


# ------------------------------------------- device continuation of the numpy stream
@pytest.mark.parametrize("seed,n_uniform,n_normal", [(1, 5000, 15000), (2, 1, 70000), (3, 1048576, 3145728),
                                                     (99, 4096, 4096 * 10), (12345, 333, 100001)])
def test_device_rng_is_bitwise_numpy(obe, hip, seed, n_uniform, n_normal):
    """PCG64 uniforms and ziggurat normals generated on the device are the numbers numpy
    would have produced, and the host generator ends in numpy's state."""
    import torch
    from optbayesexpt_amd import _devrng
    rng = np.random.default_rng(seed)
    rng.random(7)                                  # start somewhere inside the stream
    ref = np.random.default_rng(seed)
    ref.random(7)
    ds = _devrng.DeviceStream(hip, torch.device("cuda", 0), None, rng, n_uniform, n_normal)
    u = ds.uniforms().cpu().numpy()
    z = ds.normals().cpu().numpy()
    assert_array_equal(u, ref.random(n_uniform))
    zr = ref.standard_normal(n_normal)
    body = np.abs(zr) <= 3.6541528853610088
    assert_array_equal(z[body], zr[body])                       # 99.97 % of draws: bit-identical
    # ziggurat tail draws go through libm's log1p on the host (device: ocml log1p):
    # same draw, at most the last bit differs
    assert_allclose(z[~body], zr[~body], rtol=2.3e-16, atol=0)
    assert np.sum(z != zr) <= max(8, n_normal // 100000)
    assert rng.bit_generator.state == ref.bit_generator.state
    assert_array_equal(rng.random(5), ref.random(5))           # and the streams stay in step


@pytest.mark.parametrize("seed,n_uniform,n_normal", [(3, 5000, 15000), (11, 262144, 786432), (12, 70001, 700010)])
def test_generator_state_to_uniforms_and_normals_without_a_raw_buffer(obe, hip, seed, n_uniform, n_normal):
    """Round 4: obe_pcg64_uniforms_classify + obe_ziggurat_finish produce the uniforms and normals of a
    resample straight from the PCG64 state (every thread carries the state of its position; nothing raw is
    stored) — the same numbers as numpy, the same count of raw values consumed."""
    import torch
    from optbayesexpt_amd import _devrng, _lib
    rng = np.random.default_rng(seed)
    rng.random(11)
    ref = np.random.default_rng(seed)
    ref.random(11)
    st, h_state = _devrng.pcg64_state(rng)
    n_rel = n_normal + n_normal // 24 + 4096
    dev = torch.device("cuda", 0)
    u = torch.empty(n_uniform, dtype=torch.float64, device=dev)
    z = torch.empty(n_normal, dtype=torch.float64, device=dev)
    ws = torch.empty(int(hip.cdll.obe_ziggurat_workspace_bytes(n_rel)) // 8 + 1, dtype=torch.float64, device=dev)
    tables = _devrng._tables(dev)
    consumed = np.zeros(2, dtype=np.int64)
    P = _lib.c_void_p
    hip.call("obe_pcg64_uniforms_classify", _lib.host_ptr(h_state), n_uniform, n_rel, P(u.data_ptr()), P(tables.data_ptr()),
             P(ws.data_ptr()), ws.numel() * 8, None)
    hip.call("obe_ziggurat_finish", n_rel, n_normal, P(z.data_ptr()), _lib.host_ptr(consumed), P(ws.data_ptr()),
             ws.numel() * 8, None)
    assert_array_equal(u.cpu().numpy(), ref.random(n_uniform))
    zr = ref.standard_normal(n_normal)
    zz = z.cpu().numpy()
    body = np.abs(zr) <= 3.6541528853610088
    assert_array_equal(zz[body], zr[body])
    assert_allclose(zz[~body], zr[~body], rtol=2.3e-16, atol=0)
    _devrng.advance(rng, st, n_uniform + int(consumed[0]))
    assert rng.bit_generator.state == ref.bit_generator.state


@pytest.mark.parametrize("d,scale", [(3, False), (10, True)])
def test_pipelined_resample_is_the_step_by_step_resample(obe, d, scale):
    """resample() enqueued without host waits (asynchronous host results, one wait for the
    covariance, SVD under the ziggurat kernels) against the same resample issued call by call:
    same indices, same particles, same generator state; and numpy's validation of p still raises
    before the generator has moved."""
    g = np.random.default_rng(77 + d)
    n = 70001
    prior = g.normal(0.0, 1.0, (d, n)) * np.arange(1, d + 1)[:, None]
    w = g.exponential(1.0, n) ** 3
    w /= w.sum()
    out = {}
    for piped in (True, False):
        pdf = obe.ParticlePDF(prior.copy(), scale=scale)
        pdf.tuning_parameters["pipelined_resample"] = piped
        pdf.particle_weights = w.copy()
        pdf.rng = np.random.default_rng(99)
        pdf.resample()
        out[piped] = (pdf.last_resample_indices_device.cpu().numpy(), np.array(pdf.particles),
                      np.array(pdf.particle_weights), pdf.rng.bit_generator.state, pdf.mean(), pdf.covariance())
    for a, b in zip(out[True], out[False]):
        if isinstance(a, dict):
            assert a == b
        else:
            assert_array_equal(a, b)
    ref = np.random.default_rng(99)
    assert_array_equal(out[True][0], oracle.choice_indices(w, ref.random(n)))
    # invalid probabilities: ValueError, generator untouched (numpy validates before drawing)
    pdf = obe.ParticlePDF(prior.copy(), scale=scale)
    pdf.particle_weights = w * 1.01
    pdf.rng = np.random.default_rng(5)
    before = pdf.rng.bit_generator.state
    with pytest.raises(ValueError):
        pdf.resample()
    assert pdf.rng.bit_generator.state == before


@pytest.mark.parametrize("d", [1, 3, 10])
def test_resample_gather_layouts_agree(hip, d):
    """obe_resample_particles gathers from an (N, D) copy of the old cloud when it is given scratch
    for it, and from the (D, N) rows directly otherwise: the same numbers either way, and equal to
    old[:, idx] + z @ F.T computed on the host."""
    import torch
    from optbayesexpt_amd import _lib
    from optbayesexpt_amd.particlepdf import _ptr
    g = np.random.default_rng(60 + d)
    n = 70001
    old = g.normal(0.0, 1.0, (d, n)) * np.arange(1, d + 1)[:, None]
    idx = g.integers(0, n, n)
    z = g.standard_normal((n, d))
    f = g.normal(0.0, 0.1, (d, d))
    mean = old.mean(axis=1)
    od, zd, ix = torch.from_numpy(old).cuda(), torch.from_numpy(z).cuda(), torch.from_numpy(idx).cuda()
    ws = torch.empty(d * n + 64, dtype=torch.float64, device="cuda")
    out = {}
    for with_ws in (False, True):
        new = torch.empty((d, n), dtype=torch.float64, device="cuda")
        w = torch.empty(n, dtype=torch.float64, device="cuda")
        hip.call("obe_resample_particles", _ptr(od), n, d, n, _ptr(ix), _ptr(zd), _lib.host_ptr(np.ascontiguousarray(f)),
                 _lib.host_ptr(mean), 0.98, 1, _ptr(new), n, _ptr(w), _ptr(ws) if with_ws else None,
                 ws.numel() * 8 if with_ws else 0, None)
        out[with_ws] = (new.cpu().numpy(), w.cpu().numpy())
    assert_array_equal(out[True][0], out[False][0])
    assert_array_equal(out[True][1], np.full(n, 1.0 / n))
    ref = (old[:, idx] + (z @ f.T).T) * 0.98 + (mean * (1 - 0.98))[:, None]
    assert_allclose(out[True][0], ref, rtol=1e-13, atol=1e-13)


def test_sweep_timing_counts_the_launches_of_real_cycles(obe):
    """obe_sweep_timing (bench.py's roofline leg): events around every sweep-kernel launch that
    returns its result to the host."""
    import ctypes
    g = np.random.default_rng(3)
    n, ns = 40000, 4200
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    o = obe.OptBayesExpt(obe.models.lorentzian(), (np.linspace(1.5, 4.5, ns),), prior, (0.1,),
                         utility_method="variance_full", default_noise_std=500.0, auto_resample=False)
    o.tuning_parameters["sweep_shift"] = "always"
    o.tuning_parameters["speculative_sweep"] = False
    tot, cnt = ctypes.c_double(-1.0), ctypes.c_int64(-1)
    o._mlib.call("obe_sweep_timing", 1, None, None)
    for _ in range(3):
        o.opt_setting()
        o.pdf_update(((3.0,), 49000.0, 500.0))
    o._mlib.call("obe_sweep_timing", 0, ctypes.byref(tot), ctypes.byref(cnt))
    assert cnt.value == 3 and 0.0 < tot.value < 100.0
    # sweeps that pdf_update() enqueues behind its update are counted when their events are read: the
    # first opt_setting() launches its own, every update the next cycle's (the last one is never asked for)
    o.tuning_parameters["speculative_sweep"] = True
    o._mlib.call("obe_sweep_timing", 1, None, None)
    for _ in range(3):
        o.opt_setting()
        o.pdf_update(((3.0,), 49000.0, 500.0))
    o._mlib.call("obe_sweep_timing", 0, ctypes.byref(tot), ctypes.byref(cnt))
    assert cnt.value == 4 and 0.0 < tot.value < 100.0
    o.tuning_parameters["speculative_sweep"] = False
    o.opt_setting()
    o._mlib.call("obe_sweep_timing", -1, ctypes.byref(tot), ctypes.byref(cnt))
    assert cnt.value == 0 and tot.value == 0.0          # stopped: nothing accumulates


def test_device_rng_is_used_and_can_be_disabled(obe, unit):
    """Same resample through the device stream and through host calls on self.rng."""
    out = {}
    for dev in (True, False):
        pdf = obe.ParticlePDF(unit["rs_s0_x"].copy(), scale=False)
        pdf.tuning_parameters["device_rng"] = dev
        pdf.particle_weights = unit["rs_s0_w"].copy()
        pdf.rng = np.random.default_rng(4242)
        pdf.randdraw(30)
        pdf.resample()
        out[dev] = (pdf.last_draw_indices, np.array(pdf.particles), pdf.rng.bit_generator.state)
    assert_array_equal(out[True][0], out[False][0])
    assert_allclose(out[True][1], out[False][1], rtol=1e-15)     # tail normals: last bit (libm log1p)
    assert out[True][2] == out[False][2]
    assert_array_equal(out[True][0], unit["rs_s0_resample_idx"])


def test_adaptive_variance_shift(obe):
    """The unshifted sweep (one instruction fewer per evaluation) is used only while the
    reported cancellation factor kappa is small; both variants agree with the oracle and
    a badly centred cloud forces the shifted one."""
    g = np.random.default_rng(31)
    n, ns = 6000, 300
    sv = (np.linspace(1.5, 4.5, ns),)
    w = g.exponential(1.0, n)
    w /= w.sum()
    # broad prior: the spread dominates the mean (kappa small)
    broad = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    # tight posterior: amplitude and centre almost known, tiny background spread (kappa huge)
    tight = np.array([g.normal(3.0, 1e-4, n), g.normal(-1000, 0.01, n), g.normal(50000, 0.001, n)])
    for cloud, expect_unshifted in ((broad, True), (tight, False)):
        ref, ybar = oracle.yvar_full_sweep(omodels.lorentzian, oracle.flatten_settings(sv), cloud, w, (0.1,),
                                           return_mean=True)
        # (the tight cloud: a spread of 1e-3 on a background of 5e4 — the reference's own variance is good to
        # ~1e-10 there; pure 1e-10 wherever the conditioning allows it, which is everywhere on the broad cloud)
        tol = conditioning_rtol(ref, ybar, 1.0 / np.sum(w * w), floor=RTOL)
        assert expect_unshifted == bool(np.all(tol == RTOL))
        res = {}
        for mode in ("always", "never", "auto"):
            o = obe.OptBayesExpt(obe.models.lorentzian(), sv, cloud.copy(), (0.1,),
                                 utility_method="variance_full", auto_resample=False)
            o.tuning_parameters["sweep_shift"] = mode
            o.particle_weights = w
            first = o.yvar_from_parameter_draws()
            assert o.last_sweep["shifted"] == (mode != "never")     # auto starts shifted
            second = o.yvar_from_parameter_draws()
            res[mode] = (first, second, dict(o.last_sweep))
        kappa = res["always"][2]["kappa"]
        assert (kappa < obe.OptBayesExpt.KAPPA_ENTER) == expect_unshifted
        assert res["auto"][2]["shifted"] == (not expect_unshifted)
        for mode in ("always", "auto"):
            for got in res[mode][:2]:
                assert_rel(got, ref, tol, f"{mode}, kappa {kappa:.3g}")
        if expect_unshifted:
            assert_rel(res["never"][1], ref, RTOL, f"never, kappa {kappa:.3g}")
        else:   # what the guard protects against: unshifted accumulation on this cloud is off
            assert np.abs(res["never"][1] / ref - 1).max() > 1e-9
    # the guard also catches a cloud that changes under an unshifted object
    o = obe.OptBayesExpt(obe.models.lorentzian(), sv, broad.copy(), (0.1,), utility_method="variance_full",
                         auto_resample=False)
    o.particle_weights = w
    o.yvar_from_parameter_draws()
    assert o._sweep_unshifted
    o.set_pdf(tight.copy(), weights=w)
    got = o.yvar_from_parameter_draws()                  # starts unshifted, sees kappa, redoes shifted
    assert o.last_sweep["shifted"] and not o._sweep_unshifted
    ref, ybar = oracle.yvar_full_sweep(omodels.lorentzian, oracle.flatten_settings(sv), tight, w, (0.1,), return_mean=True)
    assert_rel(got, ref, conditioning_rtol(ref, ybar, 1.0 / np.sum(w * w), floor=RTOL), "tight cloud after set_pdf")


def test_unshifted_sweep_accuracy_below_the_kappa_threshold(obe):
    """Clouds whose cancellation factor sits just under KAPPA_LEAVE (the largest kappa at which an
    unshifted result is accepted): the unshifted variance must still agree with the shifted one —
    and with the oracle — far inside the 1e-10 parity tolerance."""
    g = np.random.default_rng(123)
    n, ns = 200000, 600
    sv = (np.linspace(1.5, 4.5, ns),)
    w = g.exponential(1.0, n)
    w /= w.sum()
    z = g.normal(size=(3, n))
    seen = []
    for scale in (0.2, 0.1, 0.06, 0.04):
        cloud = np.array([3.0 + 0.02 * scale * z[0], -1000.0 + 300.0 * scale * z[1], 50000.0 + 200.0 * scale * z[2]])
        res = {}
        for mode in ("always", "never"):
            o = obe.OptBayesExpt(obe.models.lorentzian(), sv, cloud.copy(), (0.1,), utility_method="variance_full",
                                 auto_resample=False)
            o.tuning_parameters["sweep_shift"] = mode
            o.particle_weights = w
            res[mode] = o.yvar_from_parameter_draws()[0]
            kappa = o.last_sweep["kappa"]
        seen.append(kappa)
        if kappa < obe.OptBayesExpt.KAPPA_LEAVE:
            assert_rel(res["never"], res["always"], 2e-11, f"unshifted vs shifted at kappa {kappa:.3g}")
    assert any(0.3 * obe.OptBayesExpt.KAPPA_LEAVE < k < obe.OptBayesExpt.KAPPA_LEAVE for k in seen), seen
    ref = oracle.yvar_full_sweep(omodels.lorentzian, oracle.flatten_settings(sv), cloud, w, (0.1,))
    assert_rel(res["always"], ref[0], RTOL, "shifted vs oracle")


def test_update_back_to_back_stress(obe, hip):
    """2000 back-to-back Bayes updates whose totals differ by orders of magnitude: the
    partial sums of one launch must never leak into the next (workspace reuse, kernel-boundary
    visibility across XCDs).  Checked through an independent reduction."""
    import torch
    from optbayesexpt_amd import _lib
    from optbayesexpt_amd.particlepdf import _ptr
    g = np.random.default_rng(17)
    n = 524288 + 77
    ws = torch.zeros(hip.workspace_bytes(n, 1, 1, 1) // 8 + 1, dtype=torch.float64, device="cuda")
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    liks = [torch.from_numpy(g.exponential(1.0, n) * 10.0 ** g.integers(-30, 30)).cuda() for _ in range(16)]
    w0 = torch.from_numpy(g.exponential(1.0, n)).cuda()
    out, sums = np.zeros(4), np.zeros(4)
    bad = 0
    for it in range(2000):
        w.copy_(w0)
        lik = liks[it % 16]
        hip.call("obe_bayes_update_lik", _ptr(lik), n, _ptr(w), _ptr(ws), ws.numel() * 8,
                 _lib.host_ptr(out), None)
        if it % 10 == 0:                      # independent check through separate launches
            hip.call("obe_weight_sums", _ptr(w), n, _ptr(ws), ws.numel() * 8, _lib.host_ptr(sums), None)
            expect_total = float(torch.sum(w0 * lik))
            bad += not (abs(sums[1] - 1.0) < 1e-12 and abs(out[0] / expect_total - 1.0) < 1e-12
                        and abs(out[1] / sums[0] - 1.0) < 1e-12)
    assert bad == 0


def test_expression_model_full_sweep_matches_oracle(obe):
    """variance_full through a plugin-compiled expression model, non-uniform weights, both
    shift variants, against the oracle."""
    import _expr_models
    f = _replay.load("full_sweep_uniform.npz")
    g = np.random.default_rng(5)
    prior, sv = f["fs_lor_prior"], (f["fs_lor_x"],)
    w = g.exponential(1.0, prior.shape[1]) ** 2
    w /= w.sum()
    ref = oracle.yvar_full_sweep(omodels.lorentzian, oracle.flatten_settings(sv), prior, w, (0.1,))
    # what an UNSHIFTED one-pass variance can be held to, from the oracle's own numbers: the cancellation factor
    # (mean y)^2 / var of every setting (tests/_replay.py: the measured ~1e-15 * kappa, x4)
    y = omodels.lorentzian(oracle.flatten_settings(sv)[:, :, None], prior[:, None, :], (0.1,))
    mean_y = np.sum(w * y, axis=-1).reshape(ref.shape)
    kappa_ref = mean_y ** 2 / ref
    report = {}
    for mode in ("always", "auto", "never"):
        o = obe.OptBayesExpt(_expr_models.expression_models()["lorentzian"], sv, prior.copy(), (0.1,),
                             utility_method="variance_full", auto_resample=False, default_noise_std=500.0)
        o.tuning_parameters["sweep_shift"] = mode
        o.particle_weights = w
        if mode == "auto":
            o._sweep_unshifted = True          # as if an earlier, well-conditioned cloud had switched the shift off
        got = o.yvar_from_parameter_draws()
        kappa = o.last_sweep["kappa"]
        if mode == "never":
            # forced unshifted: held to 1e-10 where the policy itself would keep this form (kappa <= KAPPA_LEAVE),
            # to its conditioning (4e-15 * kappa of the setting, from the ORACLE's mean and variance) where not
            assert not o.last_sweep["shifted"], o.last_sweep
            report[mode] = assert_rel(got, ref, np.maximum(RTOL, 4e-15 * kappa_ref), f"expression model, {mode}")
        else:
            # 'auto', started unshifted: the result is kept only if kappa <= KAPPA_LEAVE — otherwise the policy
            # REFUSES it and the sweep is repeated with the shift (SweepState.sweep_reported_kappa)
            if mode == "always" or np.max(kappa_ref) > 1.05 * o.KAPPA_LEAVE:
                assert o.last_sweep["shifted"], (mode, o.last_sweep)
            elif np.max(kappa_ref) < 0.95 * o.KAPPA_LEAVE:
                assert not o.last_sweep["shifted"], (mode, o.last_sweep)
            assert_rel(kappa, np.max(kappa_ref), 1e-6, "kappa reported by the sweep vs the oracle's")
            report[mode] = assert_rel(got, ref, RTOL, f"expression model, {mode}")
        o.opt_setting()
        assert o.last_setting_index == int(np.argmax(ref[0]))
    print(f"expression-model sweep, worst relative error: {report}; largest kappa {np.max(kappa_ref):.3g} "
          f"(KAPPA_LEAVE = {o.KAPPA_LEAVE})")
    assert_array_equal(o.eval_over_all_settings([3.0, -1000.0, 50000.0]),
                       np.atleast_2d(omodels.lorentzian(sv, (3.0, -1000.0, 50000.0), (0.1,))))


def test_expression_model_out_of_range_batch_is_repeated_safely(obe):
    """The fast sweep form of an expression model batches its divisions without a branch and
    poisons a batch whose denominators leave the exactly-invertible range; the sweep then comes
    back with kappa = NaN and is repeated with one IEEE reciprocal per element.  Denominators
    of ~1e50 (pair products 1e100) must give the oracle's numbers; a true pole gives NaN at that
    setting exactly as NumPy does."""
    import _expr_models
    g = np.random.default_rng(15)
    n = 1536
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    w = g.exponential(1.0, n)
    w /= w.sum()
    model = _expr_models.expression_models()["lorentzian"]
    # the batch size follows the settings count (8 per lane from 4096 settings, 4 from 1024, 2 from 512)
    worst = {}
    for ns in (4100, 1030, 520):
        sv = (np.linspace(1.5, 4.5, ns),)
        for d, expect_safe in ((0.1, False), (1e-25, True)):
            o = obe.OptBayesExpt(model, sv, prior.copy(), (d,), utility_method="variance_full", auto_resample=False,
                                 default_noise_std=500.0)
            o.particle_weights = w
            ref = oracle.yvar_full_sweep(omodels.lorentzian, oracle.flatten_settings(sv), prior, w, (d,))
            got = o.yvar_from_parameter_draws()
            assert o.last_sweep["safe"] is expect_safe, (ns, d, o.last_sweep)
            worst[(ns, d)] = assert_rel(got, ref, RTOL, f"expression model ns={ns} d={d}")
    # reference semantics (30 draws) through the same path (8200 settings: beyond the one-workgroup
    # sweep, which evaluates element by element and never needs the repeat)
    sv = (np.linspace(1.5, 4.5, 8200),)
    o = obe.OptBayesExpt(model, sv, prior.copy(), (1e-25,), default_noise_std=500.0)
    b = oracle.OracleOptBayesExpt(omodels.lorentzian, sv, prior.copy(), (1e-25,), default_noise_std=500.0)
    o.rng, b.rng = np.random.default_rng(2), np.random.default_rng(2)
    o.opt_setting(), b.opt_setting()
    assert o.last_sweep["safe"] and o.last_setting_index == b.last_setting_index
    # (30 draws of a 1536-particle cloud at a delta-narrow peak: where all draws give the same y the reference's
    # two-pass variance is rounding debris ~(eps y)^2 / sigma^2 — nothing to compare relatively below that)
    worst["draws"] = assert_rel(o._utility_dev.cpu().numpy(), b.last_utility, RTOL, "30 draws, safe form",
                                garbage_floor=(64 * 2.3e-16 * 6e4 / 500.0) ** 2)
    # a pole: y = a / (x - x0) with one particle sitting exactly on a setting
    pole = _expr_models.expression_models()["pole"]
    x = np.linspace(1.0, 2.0, 4100)
    pr = np.array([g.uniform(3.0, 4.0, 500), g.uniform(1.0, 2.0, 500)])
    pr[0, 17] = x[5]
    o = obe.OptBayesExpt(pole, (x,), pr, (), utility_method="variance_full", auto_resample=False)
    with np.errstate(all="ignore"):
        ref = oracle.yvar_full_sweep(lambda s, p, c: p[1] / (s[0] - p[0]), oracle.flatten_settings((x,)), pr,
                                     np.full(500, 1 / 500), ())
    got = o.yvar_from_parameter_draws()
    assert o.last_sweep["safe"] and np.isnan(got[0, 5]) and np.isnan(ref[0, 5])
    keep = np.arange(len(x)) != 5
    worst["pole"] = assert_rel(got[0, keep], ref[0, keep], RTOL, "pole model, settings off the pole")
    print(f"out-of-range repeats, worst relative error per case: {worst}")


def test_expression_model_fast_elementary_functions(obe):
    """The inner level of a generated sweep form evaluates sin / cos / sqrt / hypot with the
    range-checked fast versions (Cody-Waite + polynomial, rsq + Newton): arguments up to 1e6
    radians must give the oracle's variances to 1e-10; beyond the checked range (1e9) the NaN
    they return triggers the repeat with the safe twin, which calls ocml."""
    import _expr_models
    model = _expr_models.expression_models()["trig"]
    g = np.random.default_rng(77)
    n = 1024

    def numpy_model(sets, pars, cons):
        t, = sets
        w, p, a, b = pars
        c, = cons
        return a * np.sin(w * t + p) + b * np.cos(w * t) * np.sqrt(t + c) + np.hypot(a * t, b)

    t = np.linspace(0.0, 10.0, 4100)
    wts = g.exponential(1.0, n)
    wts /= wts.sum()
    for w_scale, expect_safe in ((3.0, False), (1e5, False), (3e9, True)):
        prior = np.array([g.uniform(0.5, 1.0, n) * w_scale, g.uniform(0, 6.0, n), g.normal(2.0, 0.5, n),
                          g.normal(-1.0, 0.7, n)])
        o = obe.OptBayesExpt(model, (t,), prior.copy(), (0.5,), utility_method="variance_full", auto_resample=False)
        o.particle_weights = wts
        got = o.yvar_from_parameter_draws()
        assert o.last_sweep["safe"] is expect_safe, (w_scale, o.last_sweep)
        if w_scale < 1e9:       # (beyond that the argument reduction of NumPy and of ocml differ themselves)
            ref = oracle.yvar_full_sweep(numpy_model, oracle.flatten_settings((t,)), prior, wts, (0.5,))
            assert_rel(got, ref, RTOL, f"fast elementary functions, w_scale {w_scale}")
        else:
            assert np.all(np.isfinite(got))


def test_device_limits_4_settings_16_parameters_4_channels(obe):
    """OBE_MAX_SETDIMS / OBE_MAX_DIMS / OBE_MAX_CHANNELS exercised together through an
    expression model with a noise parameter per channel: cycles against the oracle."""
    import _expr_models
    model = _expr_models.expression_models()["limits"]
    g = np.random.default_rng(404)
    n = 3000
    prior = np.vstack([g.normal(1.0, 0.3, (12, n)), g.uniform(0.5, 2.0, (4, n))])
    # s0 > 0 and an asymmetric s1 grid: with s0 = 0 the formulas are even in s1, and settings
    # +-1/3 would tie to the last bit (the winner then depends on rounding, not on the method)
    sv = (np.linspace(0.2, 1, 3), np.linspace(-1, 1.4, 4), np.linspace(0, 3, 5), np.linspace(0.1, 2, 2))
    kw = dict(scale=False, noise_parameter_index=(12, 13, 14, 15))
    true = np.r_[np.ones(12), np.ones(4)]
    for method in ("variance_approx", "variance_full"):
        a = obe.OptBayesExptNoiseParameter(model, sv, prior.copy(), (), utility_method=method, **kw)
        b = oracle.OracleOptBayesExptNoiseParameter(model, sv, prior.copy(), (), utility_method=method,
                                                    n_channels=4, **kw)
        assert a.allsettings.shape == (4, 120) and a.n_channels == 4 and a.n_dims == 16
        a.rng, b.rng = np.random.default_rng(9), np.random.default_rng(9)
        sim = np.random.default_rng(10)
        for cyc in range(12):
            xa, xb = a.opt_setting(), b.opt_setting()
            assert a.last_setting_index == b.last_setting_index, (method, cyc)
            assert_allclose(a._utility_dev.cpu().numpy(), b.last_utility, rtol=RTOL)
            y = tuple(np.asarray(model(xb, true, ())) + sim.standard_normal(4))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                a.pdf_update((xa, y))
                b.pdf_update((xb, y))
            assert a.just_resampled == b.just_resampled
            assert_allclose(a.particle_weights, b.particle_weights, rtol=RTOL,
                            atol=1e-13 * b.particle_weights.max())
        assert_allclose(a.covariance(), b.covariance(), rtol=1e-9, atol=1e-12)


# ------------------------------------------------- host mirrors, device binding, sharded draws
def test_host_views_are_tracked_or_refused(obe):
    """The reference's in-place idioms reach the device; any other in-place write is refused
    loudly (never a silently stale device copy); resizing the cloud re-plans the workspace."""
    g = np.random.default_rng(11)
    x = g.normal(0, 1, (2, 1000))
    pdf = obe.ParticlePDF(x)
    pdf.mean()                                        # device copy in use
    w = pdf.particle_weights
    w[x[0] < 0] = 0                                   # obe_noiseparam.py:71 idiom
    pdf.particle_weights = pdf.particle_weights / np.sum(pdf.particle_weights)
    ref_w = np.where(x[0] < 0, 0.0, 1e-3)
    ref_w /= ref_w.sum()
    assert_allclose(pdf.mean(), oracle.weighted_mean(x, ref_w), rtol=1e-12)
    pdf.particle_weights *= 1.0                       # ufunc with out=
    with pytest.raises(ValueError, match="read-only"):
        np.copyto(pdf.particle_weights, 1e-3)
    with pytest.raises(ValueError, match="read-only"):
        pdf.particles.sort()
    assert_allclose(pdf.mean(), oracle.weighted_mean(x, ref_w), rtol=1e-12)
    # a bigger cloud through the setter: scratch is re-planned, stale weights are reported
    big = g.normal(0, 1, (2, 300000))
    pdf.particles = big
    assert pdf.n_particles == 300000
    with pytest.raises(ValueError, match="different lengths"):
        pdf.mean()
    pdf.particle_weights = np.full(300000, 1 / 300000)
    assert_allclose(pdf.mean(), big.mean(axis=1), rtol=1e-11, atol=1e-14)
    assert_allclose(pdf.covariance(), oracle.weighted_covariance(big, np.full(300000, 1 / 300000)), rtol=1e-10,
                    atol=1e-13)


def test_object_on_a_device_that_is_not_current(obe):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    g = np.random.default_rng(12)
    x = g.normal(0, 1, (3, 5000))
    torch.cuda.set_device(0)
    pdf = obe.ParticlePDF(x, device="cuda:1")
    assert torch.cuda.current_device() == 0
    assert_allclose(pdf.mean(), x.mean(axis=1), rtol=1e-11, atol=1e-14)
    pdf.bayesian_update(np.exp(-x[0] ** 2))
    assert torch.cuda.current_device() == 0 and pdf._weights.tensor().device.index == 1


@pytest.mark.parametrize("k", [2, 3, 4, 5, 7, 8])
def test_multi_peak_lorentzian_sweep_forms(obe, k):
    """Lorentz<K>: up to 2 peaks are inverted peak by peak (two particles per reciprocal), from 3
    peaks on the K peaks of an evaluation are combined into one fraction whose denominators the
    settings of a lane invert together (range-checked; the peak-by-peak form is its safe twin).
    Both against the oracle's two-pass weighted variance, for every settings-per-lane variant,
    odd particle counts, and a grid so wide against d that the combined form leaves its range."""
    g = np.random.default_rng(100 + k)
    n = 1537
    prior = np.vstack([g.uniform(2, 4, (k, n)), g.uniform(400, 2000, (1, n)), g.normal(500, 1000, (1, n))])
    w = g.exponential(1.0, n)
    w /= w.sum()
    fn = omodels.multi_lorentzian(k)
    worst = {}
    for ns in (4100, 1030, 520, 300):                     # 8 / 4 / 2 / 1 settings per lane
        sv = (np.linspace(1.5, 4.5, ns),)
        for d in (0.1, 1e-3, 1e-7):
            o = obe.OptBayesExpt(obe.models.lorentzian(k), sv, prior.copy(), (d,), utility_method="variance_full",
                                 auto_resample=False, default_noise_std=500.0)
            o.particle_weights = w
            ref = oracle.yvar_full_sweep(fn, oracle.flatten_settings(sv), prior, w, (d,))
            got = o.yvar_from_parameter_draws()
            # q <= 1 + (2.5/d)^2; the combined tree of SPT settings spans q^(K*SPT) < 1e250
            spt = 8 if ns >= 4096 else 4 if ns >= 1024 else 2 if ns >= 512 else 1
            out_of_range = k >= 3 and spt > 1 and (k * spt) * np.log10(1 + (2.5 / d) ** 2) > 250
            in_range = k < 3 or (k * spt) * np.log10(1 + (3.0 / d) ** 2) < 240
            if out_of_range:
                assert o.last_sweep["safe"], (k, ns, d, o.last_sweep)
            if in_range:
                assert not o.last_sweep["safe"], (k, ns, d, o.last_sweep)
            worst[(ns, d)] = assert_rel(got, ref, RTOL, f"K={k} ns={ns} d={d} {o.last_sweep}")
    if k >= 3:
        # a grid that always leaves the range: the model's range hint (grid and cloud extremes, on
        # the host) starts the very first sweep with the safe form ...
        sv = (np.linspace(1.5, 4.5, 4100),)
        o = obe.OptBayesExpt(obe.models.lorentzian(k), sv, prior.copy(), (1e-7,), utility_method="variance_full",
                             auto_resample=False, default_noise_std=500.0)
        ref = oracle.yvar_full_sweep(fn, oracle.flatten_settings(sv), prior, np.full(n, 1.0 / n), (1e-7,))
        got = o.yvar_from_parameter_draws()
        assert o.last_sweep["safe"] and o._sweep_safe_streak == o.SAFE_STREAK and o._sweep_safe_run == 1
        worst["hinted"] = assert_rel(got, ref, RTOL, f"K={k}, safe form from the range hint")
        # ... and without the hint the sweep finds out by itself: after SAFE_STREAK repeats the fast
        # attempt is skipped, and tried again once every SAFE_RETRY sweeps
        o.particle_weights = w
        ref = oracle.yvar_full_sweep(fn, oracle.flatten_settings(sv), prior, w, (1e-7,))
        o._sweep_safe_streak = o._sweep_safe_run = 0
        o.SAFE_RETRY = 4
        for rep in range(o.SAFE_STREAK + 2):
            got = o.yvar_from_parameter_draws()
            assert o.last_sweep["safe"] and o._sweep_safe_streak == min(rep + 1, o.SAFE_STREAK)
            worst["repeats"] = max(worst.get("repeats", 0.0), assert_rel(got, ref, RTOL, f"K={k}, repeat {rep}"))
        assert o._sweep_safe_run == 2
        o.yvar_from_parameter_draws()
        assert o._sweep_safe_run == 3 and o.last_sweep["safe"]
        o.yvar_from_parameter_draws()              # the retry: one fast attempt, poisoned again, pinned again
        assert o._sweep_safe_run == 0 and o._sweep_safe_streak == o.SAFE_STREAK and o.last_sweep["safe"]
        # a cloud back inside the range (set by the host): the hint re-arms the fast form at once
        o2 = obe.OptBayesExpt(obe.models.lorentzian(k), sv, prior.copy(), (0.1,), utility_method="variance_full",
                              auto_resample=False, default_noise_std=500.0)
        o2._sweep_safe_streak = o2.SAFE_STREAK       # as if pinned by an earlier, wider cloud
        o2.particles = prior.copy()
        o2.yvar_from_parameter_draws()
        assert not o2.last_sweep["safe"] and o2._sweep_safe_streak == 0
    print(f"Lorentz<{k}> sweep forms, worst relative error per case (pure relative, bound {RTOL:g}): {worst}")


@pytest.mark.parametrize("k", [1, 2, 7])
@pytest.mark.parametrize("ratio", [1e10, 1e15])
def test_lorentzian_peaks_far_narrower_than_the_grid(obe, k, ratio):
    """|x - x0| / d of 1e10 and 1e15 — peaks ten to fifteen orders of magnitude
    narrower than the settings span.  The reference's arithmetic (obe_base.py:483-488 over
    demos/find_peak/sequentialLorentzian.py:53-75) stays finite there: a / (t^2 + 1) is ~1e-17 and the variance is
    that of the background.  The fast sweep forms leave their range (K < 3: the 16 denominators of two particles
    x 8 settings overflow their product, which poisons the batch; K >= 3: the range check of the combined
    fraction), the sweep is repeated with the SAFE form, and that form — element-by-element reciprocals where a
    product does not fit — gives the oracle's numbers at 1e-10 for every settings-per-lane variant, in full and
    in draws mode; the model's range hint sends the first sweep to the SAFE form without a poisoned attempt."""
    g = np.random.default_rng(31 + k)
    n = 3001
    d = 2.5 / ratio
    prior = np.vstack([g.uniform(2, 4, (k, n)), g.uniform(400, 2000, (1, n)), g.normal(500, 1000, (1, n))])
    w = g.exponential(1.0, n)
    w /= w.sum()
    fn = omodels.multi_lorentzian(k) if k > 1 else omodels.lorentzian
    for ns in (4100, 1030, 520, 300):                     # 8 / 4 / 2 / 1 settings per lane
        sv = (np.linspace(1.5, 4.5, ns),)
        # a few settings ON a particle's peak, so that the peaks are not invisible to every setting
        sv[0][::97] = prior[0, :sv[0][::97].size]
        ref = oracle.yvar_full_sweep(fn, oracle.flatten_settings(sv), prior, w, (d,))
        assert np.all(np.isfinite(ref))
        spt = 8 if ns >= 4096 else 4 if ns >= 1024 else 2 if ns >= 512 else 1
        for hinted in (True, False):
            o = obe.OptBayesExpt(obe.models.lorentzian(k), sv, prior.copy(), (d,), utility_method="variance_full",
                                 auto_resample=False, default_noise_std=500.0)
            o.particle_weights = w
            if not hinted:
                o._sweeps.range_hint_key = o._particles.version          # (as if the cloud lived on the device only)
            got = o.yvar_from_parameter_draws()
            # K >= 3: the combined fraction of a lane's settings is range-checked at 1e250; K < 3: the product of
            # two particles' denominators overflows at ~1.8e308 (the hint pins the SAFE form from 1e245 on)
            # (the hint pins the SAFE form from 1e245 on)
            decades = (k if k >= 3 else 2) * spt * np.log10(1 + ratio ** 2)
            batched = k >= 3 or spt >= 2
            limit = 250 if k >= 3 else 308.5
            leaves = batched and decades > (245 if hinted else limit)
            assert not (batched and 245 <= decades <= limit and not hinted), "choose a ratio outside the grey zone"
            assert o.last_sweep["safe"] == bool(leaves), (k, ratio, ns, hinted, o.last_sweep, o.sweep_state())
            if leaves:      # found out by a poisoned attempt (streak 1), or predicted (pinned at once)
                assert o._sweep_safe_streak == (o.SAFE_STREAK if hinted else 1)
            assert_rel(got, ref, RTOL, f"K={k} ratio={ratio:g} ns={ns} hinted={hinted}")
            o.opt_setting()
            assert o.last_setting_index == int(np.argmax(ref[0]))
    # reference semantics (30 weighted draws), tiled kernels (8 settings per lane) and the one-workgroup kernel
    for ns in (4500, 200):
        sv = (np.linspace(1.5, 4.5, ns),)
        o = obe.OptBayesExpt(obe.models.lorentzian(k), sv, prior.copy(), (d,), n_draws=30, auto_resample=False,
                             default_noise_std=500.0)
        o.particle_weights = w
        o.rng = np.random.default_rng(8)
        o.opt_setting()
        idx = o.last_draw_indices
        ref = oracle.yvar_from_draws(fn, oracle.flatten_settings(sv), prior[:, idx], (d,))
        assert_rel(o._yvar_dev.cpu().numpy(), ref, RTOL, f"K={k} ratio={ratio:g} draws mode ns={ns}")
        assert o.last_setting_index == int(np.argmax(ref[0]))


@pytest.mark.parametrize("k,noise,n", [(1, False, 70001), (7, True, 30011), (1, False, 300)])
def test_fused_update_moments_is_update_then_moments(obe, k, noise, n):
    """pdf_update() normalises the weights and accumulates the first moments of the posterior in one
    pass (obe_bayes_update_model_moments): same weights, and mean / std / the K3 block bit for bit
    what obe_bayes_update_model followed by obe_moments gives; N_eff to rounding (its partial sums are
    grouped differently).  mean() and std() afterwards launch nothing."""
    g = np.random.default_rng(500 + k)
    rows = [g.uniform(2, 4, (k, n)), g.uniform(400, 2000, (1, n)), g.normal(500, 1000, (1, n))]
    if noise:
        rows.append(g.exponential(500, (1, n)) + 1.0)
    prior = np.vstack(rows)
    w = g.exponential(1.0, n)
    w /= w.sum()
    sv = (np.linspace(1.5, 4.5, 300),)
    out = {}
    for fused in (True, False):
        if noise:
            o = obe.OptBayesExptNoiseParameter(obe.models.lorentzian(k), sv, prior.copy(), (0.1,), scale=False,
                                               noise_parameter_index=k + 2, auto_resample=False)
            rec = ((2.9,), 1400.0)
        else:
            o = obe.OptBayesExpt(obe.models.lorentzian(k), sv, prior.copy(), (0.1,), scale=False, auto_resample=False)
            rec = ((2.9,), 1400.0, 300.0)
        o.tuning_parameters["fused_moments"] = fused
        o.tuning_parameters["strict_sums"] = False        # (the 300-particle case would take the np.sum-ordered form)
        o.particle_weights = w
        o.pdf_update(rec)
        key = (o._particles.version, o._weights.version)
        assert (o._mom_host_key is not None and o._mom_host_key[:2] == key) is fused
        block = np.array(o._moments(False)[:2 + 4 * o.n_dims])
        out[fused] = (np.array(o.particle_weights), o.mean(), o.std(), block, 1.0 / o._sum_w2(), o.covariance())
    for a, b in zip(out[True][:4], out[False][:4]):
        assert_array_equal(a, b)
    assert_allclose(out[True][4], out[False][4], rtol=1e-14)
    assert_array_equal(out[True][5], out[False][5])            # the covariance pass starts from the same mean
    fn = omodels.multi_lorentzian(k) if k > 1 else omodels.lorentzian
    y = fn((2.9,), prior, (0.1,))
    sigma = prior[k + 2] if noise else 300.0
    w1 = oracle.normalized_product(w, oracle.gauss_likelihood(y, 1400.0, sigma))
    assert_allclose(out[True][0], w1, rtol=1e-10, atol=1e-13 * w1.max())
    assert_allclose(out[True][1], oracle.weighted_mean(prior, w1), rtol=1e-10)
    assert_allclose(out[True][4], oracle.effective_particles(w1), rtol=1e-10)


def test_good_setting_validates_its_probabilities_like_numpy(obe):
    """good_setting() draws with rng.choice(p = utility**pickiness / sum) (obe_base.py:781-785): when every
    utility is zero p is 0/0 and numpy raises ValueError before it consumes a uniform — so does the
    device path, and the generator is where it was."""
    g = np.random.default_rng(4)
    n = 300
    prior = np.array([g.uniform(2, 4, n), g.uniform(400, 2000, n), g.normal(500, 300, n)])
    sv = (np.linspace(1.5, 4.5, 40),)
    for method in ("variance_approx", "variance_full"):
        o = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior.copy(), (0.1,), scale=False, n_draws=5,
                             utility_method=method, auto_resample=False)
        o.rng = np.random.default_rng(9)
        x = o.good_setting(pickiness=7)                       # a regular draw first
        assert len(x) == 1
        o.default_noise_std = np.full((1, 1), np.inf)         # utility = variance / inf = 0 for every setting
        ref = np.random.default_rng(9)
        ref.random(2 * 5 + 1 if method == "variance_approx" else 1)     # (the failing call's own draws succeeded)
        with pytest.raises(ValueError):
            o.good_setting(pickiness=7)
        assert o.rng.bit_generator.state == ref.bit_generator.state
        assert o.opt_setting() == (sv[0][0],)                 # np.argmax of zeros: the first setting


@pytest.mark.parametrize("n", [300, 70001])
def test_negative_probabilities_are_refused_like_numpy(obe, n):
    """Generator.choice(p=w) raises "Probabilities are not non-negative" for a negative weight (what a
    noise parameter sigma < 0 in the prior produces through the likelihood 1/sigma): every draw on the
    device does the same — small draws, resample-sized draws, the strict CDF — in numpy's order (NaN
    first), and leaves the generator where it was."""
    g = np.random.default_rng(n)
    x = g.normal(0.0, 1.0, (2, n))
    w = g.exponential(1.0, n)
    w[5] = -w[5]
    w /= w.sum()
    with pytest.raises(ValueError, match="not non-negative"):
        np.random.default_rng(0).choice(n, p=w)
    for strict in (False, True):
        pdf = obe.ParticlePDF(x.copy())
        pdf.tuning_parameters["strict_cdf"] = strict
        pdf.particle_weights = w
        pdf.rng = np.random.default_rng(3)
        before = pdf.rng.bit_generator.state
        for call in (lambda: pdf.randdraw(7), lambda: pdf.randdraw(n), pdf.resample):
            with pytest.raises(ValueError, match="not non-negative"):
                call()
            assert pdf.rng.bit_generator.state == before
        w2 = w.copy()
        w2[9] = np.nan
        pdf.particle_weights = w2
        with pytest.raises(ValueError, match="contain NaN"):
            pdf.randdraw(7)
        assert pdf.rng.bit_generator.state == before


def test_expression_model_is_compiled_on_this_box_or_refused_clearly(obe, tmp_path, monkeypatch):
    """every plugin library the suite loads was prebuilt in the build container, so
    `build.build_plugin` (hipcc for a generated model) had never run where it will be used.  A formula with
    a constant nobody has compiled before, into a fresh plugin directory: where hipcc is present the kernels
    are compiled here and now and must agree with NumPy (evaluation bit for bit, the full sweep and a Bayes
    update at 1e-10); where it is not, the error must name the missing compiler, and a plain Python model
    function under AUTO_TRANSLATE must fall back — with a warning — to host-callable mode and still work."""
    import os
    import time
    import warnings as _w
    from optbayesexpt_amd import build, models
    monkeypatch.setattr(build, "PLUGIN_DIR", str(tmp_path))
    salt = 1.0 + (time.time_ns() % 1000003) * 1e-9                # a literal no cached library was built for
    formula = f"b + a / (((x - x0) / d)**2 + {salt!r})"
    g = np.random.default_rng(31)
    n = 3000
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    sv = (np.linspace(1.5, 4.5, 300),)

    def fn(sets, pars, cons):
        x, = sets
        x0, a, b = pars
        d, = cons
        return b + a / (((x - x0) / d) ** 2 + salt)

    if os.path.exists(build.HIPCC):
        t0 = time.time()
        model = models.from_expression(formula, settings=("x",), parameters=("x0", "a", "b"), constants=("d",))
        assert model.plugin_path.startswith(str(tmp_path)) and os.path.exists(model.plugin_path)
        print(f"plugin compiled on this box in {time.time() - t0:.1f} s: {os.path.basename(model.plugin_path)}")
        o = obe.OptBayesExpt(model, sv, prior.copy(), (0.1,), utility_method="variance_full", auto_resample=False,
                             default_noise_std=500.0)
        assert o._mlib is not o._lib
        assert_array_equal(o.eval_over_all_parameters((3.1,))[0], fn((3.1,), prior, (0.1,)))
        w = g.exponential(1.0, n)
        w /= w.sum()
        o.particle_weights = w
        ref = oracle.yvar_full_sweep(fn, oracle.flatten_settings(sv), prior, w, (0.1,))
        assert_rel(o.yvar_from_parameter_draws(), ref, RTOL, "formula compiled on this box")
        o.pdf_update(((3.1,), 49500.0, 500.0))
        want = oracle.normalized_product(w, oracle.gauss_likelihood(fn((3.1,), prior, (0.1,)), 49500.0, 500.0))
        assert_allclose(o.particle_weights, want, rtol=RTOL, atol=1e-13 * want.max())
    else:
        with pytest.raises(RuntimeError, match="hipcc"):
            models.from_expression(formula, settings=("x",), parameters=("x0", "a", "b"), constants=("d",))
        monkeypatch.setattr(models, "AUTO_TRANSLATE", True)
        with _w.catch_warnings(record=True) as caught:
            _w.simplefilter("always")
            o = obe.OptBayesExpt(fn, sv, prior.copy(), (0.1,), default_noise_std=500.0)
        assert o._device_model is None                       # host-callable mode
        assert any("kept on the host" in str(c.message) for c in caught)
        o.rng = np.random.default_rng(3)
        x = o.opt_setting()
        o.pdf_update((x, 49500.0, 500.0))
        assert np.isfinite(o.mean()).all()


def test_instance_level_hooks_are_honoured(obe):
    """a hook replaced on the INSTANCE (obe.cost_estimate = f), or patched onto the class after the
    first cycle, must be called like one overridden in a subclass (the per-class cache of hook look-ups and
    the cost / noise shortcuts used to miss both)."""
    g = np.random.default_rng(8)
    n = 2000
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    sv = (np.linspace(1.5, 4.5, 90),)

    class Mine(obe.OptBayesExpt):
        pass

    o = Mine(obe.models.lorentzian(), sv, prior.copy(), (0.1,), utility_method="variance_full", auto_resample=False,
             default_noise_std=500.0)
    base = np.array(o.utility())
    o.cost_estimate = lambda: 2.0                                   # on the instance
    assert_allclose(o.utility(), base / 2.0, rtol=1e-15)
    del o.cost_estimate
    assert_allclose(o.utility(), base, rtol=1e-15)
    o.yvar_noise_model = lambda: np.full((1, 1), 4.0 * 500.0 ** 2)   # on the instance
    assert_allclose(o.utility(), base / 4.0, rtol=1e-15)
    del o.yvar_noise_model
    Mine.cost_estimate = lambda self: np.linspace(1.0, 3.0, 90)      # on the class, after it has been looked up
    try:
        assert_allclose(o.utility(), base / np.linspace(1.0, 3.0, 90), rtol=1e-15)
    finally:
        del Mine.cost_estimate
    assert_allclose(o.utility(), base, rtol=1e-15)


def test_fused_and_unfused_update_take_the_same_resample_decisions(obe):
    """pdf_update()'s fused route (update + first moments, sum w'^2 folded from 256 workgroup
    partials) and the plain route (768 partials) deliver sum w'^2 with different summation orders — equal to
    a few ulp, which include/obe_hip.h states.  Over a whole experiment that must never flip a resample
    decision: two objects, one on each route, through 150 cycles of the find-peak loop from the same seeds."""
    g = np.random.default_rng(77)
    n = 20000
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    sv = (np.linspace(1.5, 4.5, 201),)
    objs = []
    for fused in (True, False):
        o = obe.OptBayesExpt(obe.models.lorentzian(), sv, prior.copy(), (0.1,), scale=False, default_noise_std=500.0)
        o.tuning_parameters["fused_moments"] = fused
        o.rng = np.random.default_rng(5)
        objs.append(o)
    sim = np.random.default_rng(6)
    resamples, worst = 0, 0.0
    for cyc in range(150):
        xs = [o.opt_setting() for o in objs]
        assert xs[0] == xs[1], cyc
        y = float(omodels.lorentzian(xs[0], (3.0, -1000.0, 50000.0), (0.1,))) + 500.0 * sim.standard_normal()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            for o in objs:
                o.pdf_update((xs[0], y, 500.0))
        assert objs[0].just_resampled == objs[1].just_resampled, cyc
        resamples += objs[0].just_resampled
        worst = max(worst, abs(objs[0].last_n_eff / objs[1].last_n_eff - 1.0))
        assert_array_equal(objs[0].particle_weights, objs[1].particle_weights)      # the weights are the same bits
    assert resamples >= 5 and worst < 1e-14


def test_shifted_sweep_accuracy_on_a_converged_posterior(obe):
    """the batched reciprocal of the sweep carries one Newton step (relative error <= 2e-15 on
    every inverse).  Late in an experiment the posterior is narrow and (mean y)^2 / var reaches 1e5-1e7: the
    shifted one-pass variance must still agree with the oracle's two-pass weighted variance at 1e-10 — single
    Lorentzian and the 7-peak model (combined-fraction form)."""
    g = np.random.default_rng(2)
    n = 40000
    sv = (np.linspace(1.5, 4.5, 4100),)
    w = g.exponential(1.0, n)
    w /= w.sum()
    for scale in (1e-2, 1e-3, 1e-4):
        narrow = np.array([g.normal(3.0, 0.02 * scale * 10, n), g.normal(-1000.0, 30.0 * scale * 10, n),
                           g.normal(50000.0, 50.0 * scale * 10, n)])
        o = obe.OptBayesExpt(obe.models.lorentzian(), sv, narrow.copy(), (0.1,), utility_method="variance_full",
                             auto_resample=False, default_noise_std=500.0)
        o.particle_weights = w
        got = o.yvar_from_parameter_draws()
        ref = oracle.yvar_full_sweep(omodels.lorentzian, oracle.flatten_settings(sv), narrow, w, (0.1,))
        assert o.last_sweep["shifted"] and o.last_sweep["kappa"] > 1e4, o.last_sweep
        assert_rel(got, ref, RTOL, f"scale {scale}, kappa {o.last_sweep['kappa']:.3g}")
    centres = np.array([2.2, 2.5, 2.8, 3.1, 3.4, 3.7, 3.9])
    n7 = 12000
    w7 = g.exponential(1.0, n7)
    w7 /= w7.sum()
    narrow7 = np.vstack([centres[:, None] + g.normal(0, 2e-4, (7, n7)), g.normal(1000.0, 0.5, (1, n7)),
                         g.normal(500.0, 0.5, (1, n7))])
    o = obe.OptBayesExpt(obe.models.lorentzian(7), sv, narrow7.copy(), (0.1,), utility_method="variance_full",
                         auto_resample=False, default_noise_std=500.0)
    o.particle_weights = w7
    got = o.yvar_from_parameter_draws()
    ref = oracle.yvar_full_sweep(omodels.multi_lorentzian(7), oracle.flatten_settings(sv), narrow7, w7, (0.1,))
    assert o.last_sweep["kappa"] > 1e4
    assert_rel(got, ref, RTOL, f"7 peaks, kappa {o.last_sweep['kappa']:.3g}")


# ------------------------------------------------ beyond the fused kernels' widths
def _cubic5(sets, pars, cons):
    x, = sets
    return np.array([pars[4 * c] + pars[4 * c + 1] * x + pars[4 * c + 2] * x * x + pars[4 * c + 3] * x * x * x
                     for c in range(5)])


def test_wide_model_20_parameters_5_channels_matches_the_oracle(obe):
    """The reference takes any number of parameters and channels (obe_base.py:174-176, 807-824; particlepdf.py:105).
    A 20-parameter, 5-channel model (five cubics) as a compiled device model: sweeps (draws and full), the 5-channel
    likelihood and update, and a resample of the 20-row cloud — whose moments and gather run TILED over 8 rows at a
    time, the widths above OBE_FAST_DIMS = 16 having no kernels of their own — against the oracle classes from the
    same seeds: draws, chosen settings, resample decisions and resample indices exact, the rest 1e-10."""
    import _expr_models
    from optbayesexpt_amd import _lib
    model = _expr_models.expression_models()["wide"]
    assert model.n_read == 20 and model.n_channels == 5 and 20 > _lib.OBE_FAST_DIMS
    g = np.random.default_rng(2005)
    n = 6000
    prior = g.normal(0.0, 1.0, (20, n)) * (1.0 + np.arange(20)[:, None] / 7.0) + np.arange(20)[:, None] * 0.1
    sv = (np.linspace(-1.0, 1.5, 301),)
    true = g.normal(0.0, 1.0, 20)
    for method in ("variance_approx", "variance_full"):
        kw = dict(scale=False, utility_method=method, default_noise_std=2.0)
        a = obe.OptBayesExpt(model, sv, prior.copy(), (), **kw)
        b = oracle.OracleOptBayesExpt(_cubic5, sv, prior.copy(), (), n_channels=5, **kw)
        assert a._device_model is model and a.n_channels == 5
        a.rng, b.rng = np.random.default_rng(31), np.random.default_rng(31)
        sim = np.random.default_rng(32)
        resamples = 0
        for cyc in range(6):
            xa, xb = a.opt_setting(), b.opt_setting()
            if method == "variance_approx":
                assert_array_equal(a.last_draw_indices, b.last_draw_indices)
            assert a.last_setting_index == b.last_setting_index and xa == xb, (method, cyc)
            assert_rel(a._utility_dev.cpu().numpy(), b.last_utility, RTOL, f"{method} utility, cycle {cyc}")
            y = _cubic5(xb, true, ()) + 2.0 * sim.standard_normal(5)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                a.pdf_update((xa, tuple(y), (2.0,) * 5))
                b.pdf_update((xb, tuple(y), (2.0,) * 5))
            assert a.just_resampled == b.just_resampled, (method, cyc)
            resamples += a.just_resampled
            if a.just_resampled:
                assert_array_equal(a.last_resample_indices_device.cpu().numpy(), b.last_draw_indices)
                pa, pb = np.array(a.particles), np.array(b.particles)
                floor = 256 * 2.3e-16 * np.sqrt(np.max(np.linalg.eigvalsh(np.cov(prior))))
                for i in range(20):
                    assert_allclose(pa[i], pb[i], rtol=RTOL, atol=floor, err_msg=f"row {i}, cycle {cyc}")
            else:
                _replay.close_weights(a.particle_weights, b.particle_weights, RTOL, f"weights, cycle {cyc}")
            assert_allclose(a.mean(), b.mean(), rtol=RTOL, atol=1e-12)
            _replay.close_cov(a.covariance(), b.covariance(), b.mean(), b.std(), RTOL, f"covariance, cycle {cyc}")
        assert resamples >= 1, method
        assert a.rng.bit_generator.state == b.rng.bit_generator.state


def test_device_limits_8_settings_32_parameters_8_channels(obe):
    """OBE_MAX_SETDIMS = 8, OBE_MAX_DIMS = 32, OBE_MAX_CHANNELS = 8 together (round 6): a compiled model of 8 setting
    dimensions (3^8 = 6561 settings) and 8 channels over a cloud of 32 rows — 24 model parameters + one noise
    parameter per channel (the noise rows 24..31 travel to the sweep in the 40 bits of OBE_NOISE_FROM_MOMENTS) —
    through reference-semantics and full sweeps, 8-channel updates with per-particle sigmas, a resample of the
    32-row cloud (tiled moments, run-time-loop gather) and the constraint mask, against the oracle class."""
    import _expr_models
    from optbayesexpt_amd import _lib
    model = _expr_models.expression_models()["limits8"]
    assert (model.n_setdims, model.n_channels, model.n_read) == (_lib.OBE_MAX_SETDIMS, _lib.OBE_MAX_CHANNELS, 24)

    def numpy_model(sets, pars, cons):
        return np.array([pars[3 * c] + pars[3 * c + 1] * sets[c] + pars[3 * c + 2] * sets[(c + 1) % 8] * sets[(c + 3) % 8]
                         for c in range(8)])

    g = np.random.default_rng(808)
    n = 5000
    prior = np.vstack([g.normal(0.5, 1.0, (24, n)), g.exponential(0.8, (8, n)) + 0.02])
    sv = tuple(np.linspace(-1.0 + 0.1 * k, 1.0, 3) for k in range(8))
    true = g.normal(0.5, 1.0, 24)
    for method in ("variance_approx", "variance_full"):
        kw = dict(scale=False, utility_method=method, noise_parameter_index=tuple(range(24, 32)))
        a = obe.OptBayesExptNoiseParameter(model, sv, prior.copy(), (), **kw)
        b = oracle.OracleOptBayesExptNoiseParameter(numpy_model, sv, prior.copy(), (), n_channels=8, **kw)
        assert a._device_model is model and a.n_dims == 32 == _lib.OBE_MAX_DIMS and a.allsettings.shape == (8, 6561)
        a.rng, b.rng = np.random.default_rng(61), np.random.default_rng(61)
        sim = np.random.default_rng(62)
        resamples = 0
        for cyc in range(5):
            xa, xb = a.opt_setting(), b.opt_setting()
            if method == "variance_approx":
                assert_array_equal(a.last_draw_indices, b.last_draw_indices)
            assert a.last_setting_index == b.last_setting_index and tuple(xa) == tuple(xb), (method, cyc)
            assert_rel(a._utility_dev.cpu().numpy(), b.last_utility, RTOL, f"{method} utility, cycle {cyc}")
            y = numpy_model(xb, true, ()) + 0.8 * sim.standard_normal(8)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                a.pdf_update((xa, tuple(y)))
                b.pdf_update((xb, tuple(y)))
            assert a.just_resampled == b.just_resampled, (method, cyc)
            resamples += a.just_resampled
            wa, wb = np.array(a.particle_weights), np.array(b.particle_weights)
            assert_array_equal(wa == 0.0, wb == 0.0)
            if a.just_resampled:
                assert_array_equal(a.last_resample_indices_device.cpu().numpy(), b.last_draw_indices)
                # (the nudge of a 32-row cloud is a 32-term product with the factors of a 32 x 32 LAPACK SVD: its
                # absolute round-off floor is taken at 2048 units, as for the 10-parameter cloud of c5; measured 720)
                floor = 2048 * 2.3e-16 * np.sqrt(np.max(np.linalg.eigvalsh(np.cov(prior))))
                assert_allclose(np.array(a.particles), np.array(b.particles), rtol=RTOL, atol=floor)
                assert_allclose(wa, wb, rtol=RTOL)
            else:
                _replay.close_weights(wa, wb, RTOL, f"weights, cycle {cyc}")
            assert_allclose(a.mean(), b.mean(), rtol=RTOL, atol=1e-12)
            assert_allclose(a.yvar_noise_model(), b.yvar_noise_model(), rtol=1e-11)
        assert resamples >= 1, method
        assert a.rng.bit_generator.state == b.rng.bit_generator.state


def test_cloud_of_40_parameters_every_particlepdf_method(obe):
    """ParticlePDF alone has no model and no width limit (particlepdf.py:105): 40 rows — mean, std, covariance
    (55 tile pairs), update, randdraw and resample against NumPy / the oracle class."""
    g = np.random.default_rng(40)
    d, n = 40, 5000
    mix = g.normal(size=(d, d)) / np.sqrt(d) + np.eye(d)
    x = mix @ g.normal(size=(d, n)) + np.arange(d)[:, None]
    a = obe.ParticlePDF(x.copy(), scale=False, auto_resample=False)
    b = oracle.OracleParticlePDF(x.copy(), scale=False, auto_resample=False)
    lik = np.exp(-0.5 * ((x[3] - 3.2) / 0.8) ** 2)
    a.bayesian_update(lik)
    b.bayesian_update(lik)
    assert_allclose(a.particle_weights, b.particle_weights, rtol=1e-13)
    assert_allclose(a.mean(), b.mean(), rtol=1e-12)
    assert_allclose(a.std(), b.std(), rtol=1e-9)
    _replay.close_cov(a.covariance(), b.covariance(), b.mean(), b.std(), 1e-11, "40 x 40 covariance")
    assert_array_equal(a.covariance(), a.covariance().T)
    a.rng, b.rng = np.random.default_rng(41), np.random.default_rng(41)
    assert_array_equal(a.randdraw(7), b.randdraw(7))
    floor = 256 * 2.3e-16 * np.sqrt(np.max(np.linalg.eigvalsh(np.cov(x))))
    for scale in (False, True):          # particlepdf.py:303-306: the contraction towards the mean
        for o in (a, b):
            o.tuning_parameters["scale"] = scale
            o.bayesian_update(lik if not scale else np.exp(-0.5 * ((np.asarray(o.particles)[7] - 7.1) / 0.9) ** 2))
        assert_allclose(a.particle_weights, b.particle_weights, rtol=1e-9)
        a.resample()
        b.resample()
        assert_array_equal(a.last_draw_indices, b.last_draw_indices)
        assert_allclose(a.particles, b.particles, rtol=RTOL, atol=floor, err_msg=f"scale={scale}")
        assert_array_equal(a.particle_weights, b.particle_weights)
        floor *= 40.0                    # (the second resample starts from particles that agree to the first one's floor)
    assert a.rng.bit_generator.state == b.rng.bit_generator.state


def test_models_beyond_the_device_limits_run_as_host_callable_models(obe):
    """Nothing about the SHAPE of a model makes this package refuse it (VERDICT r5 missing #3): nine output
    channels, a cloud of 33 rows under a built-in device model, a formula of nine channels — each runs with the
    model function on the host (a RuntimeWarning says so where a device model was asked for) and reproduces the
    oracle: draws and chosen settings exact, utility / weights 1e-10."""
    from optbayesexpt_amd import _lib, models

    def nine(sets, pars, cons):
        x, = sets
        return np.array([pars[0] * np.cos(k * x) + pars[1] * k + pars[2] * x for k in range(9)])

    g = np.random.default_rng(99)
    n = 3000
    prior = g.normal(1.0, 0.5, (3, n))
    sv = (np.linspace(0.0, 3.0, 97),)
    a = obe.OptBayesExpt(nine, sv, prior.copy(), (), scale=False, default_noise_std=0.5)
    b = oracle.OracleOptBayesExpt(nine, sv, prior.copy(), (), scale=False, default_noise_std=0.5, n_channels=9)
    assert a.n_channels == 9 > _lib.OBE_MAX_CHANNELS and a._device_model is None
    a.rng, b.rng = np.random.default_rng(5), np.random.default_rng(5)
    sim = np.random.default_rng(6)
    for cyc in range(5):
        xa, xb = a.opt_setting(), b.opt_setting()
        assert_array_equal(a.last_draw_indices, b.last_draw_indices)
        assert a.last_setting_index == b.last_setting_index, cyc
        assert_rel(np.asarray(a.last_utility).reshape(-1), b.last_utility, RTOL, f"nine channels, utility, cycle {cyc}")
        y = nine(xb, (1.2, 0.8, 1.1), ()) + 0.5 * sim.standard_normal(9)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            a.pdf_update((xa, tuple(y), (0.5,) * 9))
            b.pdf_update((xb, tuple(y), (0.5,) * 9))
        assert a.just_resampled == b.just_resampled
        if not a.just_resampled:
            _replay.close_weights(a.particle_weights, b.particle_weights, RTOL, f"nine channels, weights, cycle {cyc}")
        assert_allclose(a.mean(), b.mean(), rtol=RTOL)
    # nine channels whose noise levels are nine parameter rows (obe_noiseparam.py:81-136): the likelihood takes its
    # sigma from the particles, group by group; the constraint mask over nine rows
    prior12 = np.vstack([g.normal(1.0, 0.5, (3, n)), g.exponential(0.5, (9, n)) + 0.01])
    kw = dict(scale=False, noise_parameter_index=tuple(range(3, 12)))
    a = obe.OptBayesExptNoiseParameter(nine, sv, prior12.copy(), (), **kw)
    b = oracle.OracleOptBayesExptNoiseParameter(nine, sv, prior12.copy(), (), n_channels=9, **kw)
    a.rng, b.rng = np.random.default_rng(15), np.random.default_rng(15)
    for cyc in range(4):
        xa, xb = a.opt_setting(), b.opt_setting()
        assert a.last_setting_index == b.last_setting_index, cyc
        assert_rel(np.asarray(a.last_utility).reshape(-1), b.last_utility, RTOL, f"nine noise rows, utility, cycle {cyc}")
        y = nine(xb, (1.2, 0.8, 1.1), ()) + 0.5 * sim.standard_normal(9)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            a.pdf_update((xa, tuple(y)))
            b.pdf_update((xb, tuple(y)))
        assert a.just_resampled == b.just_resampled, cyc
        if a.just_resampled:
            assert_array_equal(np.array(a.particle_weights) == 0.0, np.array(b.particle_weights) == 0.0)
        else:
            _replay.close_weights(a.particle_weights, b.particle_weights, RTOL, f"nine noise rows, weights, cycle {cyc}")
    # a built-in device model over a cloud of 33 rows (the model reads 3 of them): OBE_MAX_DIMS = 32 rows is what a
    # device model may be given — the 34th makes it a host-callable model, with a warning
    prior33 = np.vstack([g.uniform(2, 4, (1, n)), g.uniform(-2000, -400, (1, n)), g.normal(50000, 1000, (1, n)),
                         g.normal(0, 1, (30, n))])
    xs = (np.linspace(1.5, 4.5, 101),)
    with pytest.warns(RuntimeWarning, match="evaluated on the host"):
        a = obe.OptBayesExpt(obe.models.lorentzian(), xs, prior33.copy(), (0.1,), scale=False, default_noise_std=500.0)
    b = oracle.OracleOptBayesExpt(omodels.lorentzian, xs, prior33.copy(), (0.1,), scale=False, default_noise_std=500.0,
                                  n_channels=1)
    assert a._device_model is None and a.n_dims == 33 > _lib.OBE_MAX_DIMS
    a.rng, b.rng = np.random.default_rng(7), np.random.default_rng(7)
    xa, xb = a.opt_setting(), b.opt_setting()
    assert a.last_setting_index == b.last_setting_index
    a.pdf_update((xa, 49500.0, 500.0))
    b.pdf_update((xb, 49500.0, 500.0))
    _replay.close_weights(a.particle_weights, b.particle_weights, RTOL, "33 rows, weights")
    # a formula of nine channels: from_expression hands back the NumPy form, with a warning, and nothing is compiled
    with pytest.warns(RuntimeWarning, match="kept on the host"):
        f9 = models.from_expression(tuple(f"a*cos({k}*x) + b*{k} + c*x" for k in range(9)), settings=("x",),
                                    parameters=("a", "b", "c"))
    assert not isinstance(f9, models.DeviceModel)
    assert_allclose(f9(sv, (1.2, 0.8, 1.1), ()), nine(sv, (1.2, 0.8, 1.1), ()), rtol=1e-15)
