"""Sweeper composition on the device and the command surface the
reference's TCP server drives (GPU).

The golden sweeps come from the reference's own demos/sweeper/obe_sweeper.py class
(tests/golden/make_golden.py::sweeper_cases).
"""
import json
import warnings

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

import _replay
import oracle
from oracle import models as omodels

pytestmark = pytest.mark.gpu


def make(obe, fx, cls=None, **extra):
    ctor = dict(fx["meta"]["ctor"])
    ctor.update(extra)
    cls = cls or obe.OptBayesExptSweeper
    return cls(obe.models.lorentzian(), (fx["setval_0"],), fx["prior"].copy(), tuple(fx["cons"]), **ctor)


@pytest.mark.parametrize("name", _replay.SWEEPER_TRAJECTORIES)
def test_sweeper_trajectory_matches_reference(hip, name):
    """Sweep utility over all (start, stop) pairs to 1e-10, the chosen pair and the draw
    indices exactly, then one update per point of the sweep (270 / 165 points with
    23 / 18 resamples) and the posterior moments to 1e-10."""
    import optbayesexpt_amd as obe
    from optbayesexpt_amd import sweeper
    fx = _replay.load_traj(name)
    o = make(obe, fx)
    assert o.start_stop_values.shape == (len(fx["pairs"]), 2) and o.cost_of_new_sweep == 5.0
    stats = _replay.replay_sweeper(fx, o, 1e-10, lambda g: setattr(sweeper, "rng", g),
                                   get_draw_idx=lambda x: x.last_draw_indices,
                                   get_utility=lambda x: x._sweep_utility_dev.cpu().numpy())
    assert stats["points"] == len(fx["y_concat"])


def test_sweeper_full_sweep_and_strict_cumsum_against_oracle(hip):
    """Weighted all-particle point utility ('variance_full') under the sweep composition,
    with np.cumsum's serial rounding reproduced (strict_cdf): against the oracle."""
    import optbayesexpt_amd as obe
    fx = _replay.load_traj("sweeper_opt")
    a = make(obe, fx, utility_method="variance_full")
    a.tuning_parameters["strict_cdf"] = True
    ctor = dict(fx["meta"]["ctor"], utility_method="variance_full")
    b = oracle.OracleOptBayesExptSweeper(omodels.lorentzian, (fx["setval_0"],), fx["prior"].copy(),
                                         tuple(fx["cons"]), **ctor)
    a.rng, b.rng = np.random.default_rng(5), np.random.default_rng(5)
    sim = np.random.default_rng(6)
    x = fx["setval_0"]
    for cyc in range(4):
        pa, pb = a.opt_setting(), b.opt_setting()
        assert_allclose(a._sweep_utility_dev.cpu().numpy(), b.last_utility, rtol=1e-10,
                        atol=1e-13 * np.max(np.abs(b.last_utility)))
        assert a.last_setting_index == b.last_setting_index and tuple(pa) == tuple(pb)
        # the running sum itself: bitwise np.cumsum of the device's own point utility
        assert_array_equal(a._cum_dev.cpu().numpy(), np.cumsum(a._utility_dev.cpu().numpy()))
        xs = x[pa[0]:pa[1]]
        ys = omodels.lorentzian((xs,), (3.1, 1200.0, 300.0), (0.1,)) + 800.0 * sim.standard_normal(len(xs))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            a.pdf_update(((xs,), ys))
            b.pdf_update(((xs,), ys))
        assert_allclose(a.mean(), b.mean(), rtol=1e-9)
        assert_allclose(a.std(), b.std(), rtol=1e-8)


def test_reference_style_host_subclass_runs_unchanged(hip):
    """The reference's sweeper is *user code*: a subclass composing ``self.utility()`` with
    NumPy on the host and calling ``super().pdf_update`` per point.  The same composition
    written against this package's OptBayesExptNoiseParameter gives the device sweeper's
    numbers — i.e. such subclasses keep working on the host mirrors."""
    import optbayesexpt_amd as obe

    class HostSweeper(obe.OptBayesExptNoiseParameter):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            grid = list(range(0, len(self.setting_values[0]), 3))
            if grid[-1] != len(self.setting_values[0]) - 1:
                grid.append(len(self.setting_values[0]) - 1)
            self.pairs = np.array([[s, e] for i, s in enumerate(grid) for e in grid[i + 1:]])

        def pdf_update(self, record):
            (xs,), ys = record
            for x, y in zip(xs, ys):
                super().pdf_update(((x,), y))

        def opt_setting(self):
            run = np.cumsum(self.utility())
            u = (run[self.pairs[:, 1]] - run[self.pairs[:, 0]]) / (self.pairs[:, 1] - self.pairs[:, 0] + 5.0)
            self.host_sweep_utility = u
            self.last_setting_index = int(np.argmax(u))
            return self.pairs[self.last_setting_index]

    fx = _replay.load_traj("sweeper_opt")
    h = make(obe, fx, cls=HostSweeper)
    d = make(obe, fx)
    assert_array_equal(h.pairs, d.start_stop_indices)
    h.rng, d.rng = np.random.default_rng(77), np.random.default_rng(77)
    x, pos = fx["setval_0"], 0
    for cyc in range(3):
        ph, pd = h.get_setting(), d.get_setting()
        assert_allclose(d._sweep_utility_dev.cpu().numpy(), h.host_sweep_utility, rtol=1e-12)
        assert tuple(ph) == tuple(pd)
        xs = x[pd[0]:pd[1]]
        ys = fx["y_concat"][pos:pos + len(xs)]
        pos += len(xs)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            h.pdf_update(((xs,), ys))
            d.pdf_update(((xs,), ys))
        # h updated point by point (K2 + host resample test each), d in device batches with the
        # test on the device: same bits, through the resamples inside the sweeps
        assert_array_equal(h.particle_weights, d.particle_weights)
        assert_array_equal(h.mean(), d.mean())
        assert_array_equal(h.particles, d.particles)
        assert sum(a for _, a in d.last_sweep_batches) == len(xs)
        assert len(d.last_sweep_batches) < len(xs) or len(xs) < 3
    # a sweep of ONE point goes through the per-point path: no device batches, and none left over from the sweep before
    h.pdf_update(((xs[:1],), ys[:1]))
    d.pdf_update(((xs[:1],), ys[:1]))
    assert d.last_sweep_batches == []
    assert_array_equal(h.particle_weights, d.particle_weights)


def test_sweeper_overridden_cost_and_other_selection_methods(hip):
    import optbayesexpt_amd as obe
    from optbayesexpt_amd import sweeper
    fx = _replay.load_traj("sweeper_good")

    class Costly(obe.OptBayesExptSweeper):
        def sweep_cost_estimate(self):
            return (self.start_stop_indices[:, 1] - self.start_stop_indices[:, 0]) ** 1.5 + 2.0

    a, b = make(obe, fx, cls=Costly), make(obe, fx)
    a.rng, b.rng = np.random.default_rng(1), np.random.default_rng(1)
    ua, ub = a.sweep_utility(), b.sweep_utility()
    assert_allclose(ua * a.sweep_cost_estimate(), ub * b.sweep_cost_estimate(), rtol=1e-13)
    # good_setting consumes exactly one uniform of the module generator, random_setting one integer
    sweeper.rng = np.random.default_rng(3)
    ref = np.random.default_rng(3)
    pair = b.good_setting()
    ref.random()
    assert sweeper.rng.bit_generator.state == ref.bit_generator.state
    assert tuple(pair) == tuple(b.start_stop_indices[b.last_setting_index]) and pair[1] > pair[0]
    pair = b.random_setting()
    assert tuple(pair) == tuple(b.start_stop_indices[b.last_setting_index])
    c = make(obe, fx, selection_method="random")
    assert c.get_setting.__func__ is obe.OptBayesExptSweeper.random_setting
    # changing the sub-sampling the way the reference's attributes allow
    b.start_stop_subsample = 7
    b.start_stop_indices = b._generate_start_stop_indices()
    u7 = b.sweep_utility()
    assert len(u7) == len(b.start_stop_indices) == 15 * 16 // 2 and np.all(np.isfinite(u7))


def test_sweeper_at_16384_settings(hip):
    """Config-5 sized setting axis: 5462 grid points -> 14.9 M (start, stop) pairs; the pair
    table is generated, uploaded once, and differenced/arg-maxed on the device."""
    import optbayesexpt_amd as obe
    g = np.random.default_rng(12)
    n, ns = 20000, 16384
    prior = np.array([g.uniform(2, 4, n), g.uniform(400, 2000, n), g.normal(500, 1000, n), g.exponential(500, n)])
    x = np.linspace(1.5, 4.5, ns)
    o = obe.OptBayesExptSweeper(obe.models.lorentzian(), (x,), prior, (0.1,), 3, scale=False,
                                utility_method="variance_full")
    n_grid = len(range(0, ns, 3))          # 16383 = 3 * 5461 is on the grid: no extra end point
    assert o.start_stop_indices.shape == (n_grid * (n_grid - 1) // 2, 2)
    assert_array_equal(o.start_stop_indices[:3], [[0, 3], [0, 6], [0, 9]])
    assert_array_equal(o.start_stop_indices[-1], [ns - 4, ns - 1])
    pair = o.opt_setting()
    u = o._utility_dev.cpu().numpy()
    run = np.cumsum(u)
    p = o.start_stop_indices
    want = (run[p[:, 1]] - run[p[:, 0]]) / (p[:, 1] - p[:, 0] + 5.0)
    got = o._sweep_utility_dev.cpu().numpy()
    assert_allclose(got, want, rtol=1e-9, atol=1e-12 * want.max())
    assert got[o.last_setting_index] == got.max() and int(np.argmax(got)) == o.last_setting_index
    assert want[o.last_setting_index] >= want.max() * (1 - 1e-9)
    assert tuple(pair) == tuple(p[o.last_setting_index])
    key = o._pairs_key
    o.opt_setting()
    assert o._pairs_key is key            # no second upload of the 240 MB table


def test_server_command_surface_is_json_serialisable(hip):
    """What the reference's OBE_Server.run() does with its ``obe_engine`` (obe_server.py:245-313),
    restated without the socket: every reply it sends must survive json.dumps, and the
    records it builds from JSON lists must be accepted by pdf_update."""
    import optbayesexpt_amd as obe
    g = np.random.default_rng(8)
    n = 3000
    prior = (g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n))
    sets = (np.linspace(1.5, 4.5, 61),)
    engine = obe.OptBayesExpt(obe.models.lorentzian(), sets, prior, (0.1,), scale=False)

    def send(obj):
        return json.loads(json.dumps(obj))

    assert np.shape(send(np.array(engine.allsettings).tolist())) == (1, 61)        # 'getset'
    assert np.shape(send(engine.parameters.tolist())) == (3, n)                       # 'getpar'
    assert send(engine.cons) == [0.1]                                                 # 'getcon'
    assert len(send(engine.particle_weights.tolist())) == n                           # 'getwgt'
    opt = send(engine.opt_setting())                                                  # 'optset'
    assert len(opt) == 1 and 1.5 <= opt[0] <= 4.5
    good = send(engine.good_setting(pickiness=json.loads("7")))                       # 'goodset' + pickiness
    good2 = send(engine.good_setting())
    assert len(good) == len(good2) == 1
    message = json.loads(json.dumps({"command": "newdat", "x": opt, "y": 49500.0, "s": 500.0}))
    engine.pdf_update((message["x"], message["y"], message["s"]))                     # 'newdat'
    w = send(engine.particle_weights.tolist())
    assert abs(sum(w) - 1.0) < 1e-12 and max(w) > 1.0 / n
    mean, std, cov = send(engine.mean().tolist()), send(engine.std().tolist()), send(engine.covariance().tolist())
    assert len(mean) == len(std) == 3 and np.shape(cov) == (3, 3)                     # 'getmean' 'getstd' 'getcov'
    # 'newrun' builds a fresh engine from the stored constructor arguments (obe_server.py:72-94)
    engine2 = obe.OptBayesExpt(obe.models.lorentzian(), sets, prior, (0.1,), scale=False)
    assert_allclose(engine2.particle_weights, 1.0 / n)
    # the reference's own server test, its literals and its assert_array_equal (tests/test_server.py:111-131;
    # the engine of tests/server_script_61983.py): 'getwgt' before and after a 'newdat'
    eng = obe.OptBayesExpt(obe.models.line_ab(), (np.array([0, 1, 2]),), (np.array([0, 1, 2, 3]), np.array([1, 3, 2, 4])), ())
    assert_array_equal(np.ones(4) / 4.0, send(eng.particle_weights.tolist()), err_msg="weights not echoed correctly")
    msg = send({"command": "newdat", "x": (1,), "y": 5.0, "s": 1.0})
    eng.pdf_update((msg["x"], msg["y"], msg["s"]))
    lkl = np.exp(-(np.array((1, 4, 4, 7)) - 5.0) ** 2 / 2)
    assert_array_equal(lkl / np.sum(lkl), send(eng.particle_weights.tolist()), err_msg="incorrect updated weights")


@pytest.mark.parametrize("call", ["opt_setting", "good_setting", "sweep_utility"])
def test_sweeper_validates_the_weights_of_its_draws(hip, call):
    """A draws-mode sweep nobody waits for defers numpy's validation of p (Generator.choice inside
    randdraw, particlepdf.py:330) to the caller's own synchronisation; the sweeper's selection
    methods are such callers: un-normalised weights raise ValueError from them and the generator
    is back where it was before the draw (numpy validates before it consumes uniforms)."""
    import optbayesexpt_amd as obe
    fx = _replay.load_traj("sweeper_opt")
    o = make(obe, fx)
    o.rng = np.random.default_rng(11)
    getattr(o, call)()                                   # a valid cycle first
    o.particle_weights = np.array(o.particle_weights) * 1.01
    before = o.rng.bit_generator.state
    with pytest.raises(ValueError):
        getattr(o, call)()
    assert o.rng.bit_generator.state == before
    w = np.array(o.particle_weights) / 1.01
    w[3] = np.nan
    o.particle_weights = w
    with pytest.raises(ValueError):
        getattr(o, call)()
    assert o.rng.bit_generator.state == before
