#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/make_golden.py

It imports optbayesexpt 1.2.0 read-only from /root/reference, drives its
ParticlePDF / OptBayesExpt / OptBayesExptNoiseParameter classes through seeded
measurement cycles (seeding recipe: construct, then ``obe.rng = default_rng(seed)``,
SURVEY.md §8c) and stores inputs and outputs as small ``.npz`` files.  The model
functions below are plain restatements of the demo formulas (the same ones as
oracle/models.py); nothing else of the reference is stored.

Fixture layout (trajectory cases): ``meta`` (JSON), ``prior`` (D,Np), ``setval_k``,
``cons``, and per cycle: ``chosen_index``, ``y_meas``, ``utility``, ``draw_idx``,
``resampled``, ``mean``, ``std``, ``cov`` , ``sum_w2``; weight snapshots at the
cycles listed in ``w_cycles``; particle snapshots after the first resamples.
"""
import json
import os
import sys
import warnings

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))

import optbayesexpt as ref                      # noqa: E402  (the real reference)
from oracle import models                       # noqa: E402  (plain model formulas)

HERE = os.path.dirname(os.path.abspath(__file__))
assert ref.__version__ == "1.2.0", ref.__version__


class RecordingRNG:
    """Delegates to a seeded Generator and records what ``choice`` returned, so
    the draw indices (never exposed by the reference) become part of the fixture."""

    def __init__(self, seed):
        self._g = np.random.default_rng(seed)
        self.choices = []

    def choice(self, a, size=None, p=None, **kw):
        out = self._g.choice(a, size=size, p=p, **kw)
        self.choices.append(np.array(out, dtype=np.int64).reshape(-1))
        return out

    def __getattr__(self, name):
        return getattr(self._g, name)


MODELS = {
    "lorentzian": models.lorentzian,
    "multi_lorentzian_7": models.multi_lorentzian(7),
    "line_ab": models.line_ab,
    "line_mb": models.line_mb,
    "first_parameter": models.first_parameter,
    "rabi": models.rabi,
    "coil": models.coil,
}


def run_trajectory(name, model, setting_values, prior, cons, true_pars, sigma_meas,
                   n_cycles, seed, cls="base", ctor=None, selection="opt",
                   pickiness=None, max_particle_snaps=3, max_weight_snaps=None):
    ctor = dict(ctor or {})
    fn = MODELS[model]
    prior = np.asarray(prior, dtype=np.float64)
    if cls == "base":
        obe = ref.OptBayesExpt(fn, setting_values, prior.copy(), cons, **ctor)
    else:
        obe = ref.OptBayesExptNoiseParameter(fn, setting_values, prior.copy(), cons, **ctor)
    rng = RecordingRNG(seed)
    obe.rng = rng
    import optbayesexpt.obe_base as ref_base
    ref_base.rng = np.random.default_rng(seed + 2)        # module-level generator (full_kld noise)
    sim = np.random.default_rng(seed + 1)
    n_s = obe.allsettings.shape[1]
    C, D, Np = obe.n_channels, obe.n_dims, obe.n_particles

    out = dict(chosen_index=[], y_meas=[], utility=[], draw_idx=[], resampled=[],
               mean=[], std=[], cov=[], sum_w2=[], noise_var=[])
    w_cycles, w_snaps, p_cycles, p_snaps, resample_idx = [], [], [], [], []
    for cyc in range(n_cycles):
        rng.choices.clear()
        if selection == "opt":
            util = obe.utility()             # consumes the draws
            draw_idx = rng.choices[0].copy()
            best = int(np.argmax(util))
            obe.last_setting_index = best
            x = tuple(obe.allsettings[:, best])
            # cross-check: this is exactly what opt_setting() does (obe_base.py:745-756)
        else:
            # good_setting(): utility, then one more choice over the settings
            x = obe.good_setting(pickiness)
            draw_idx = rng.choices[0].copy()
            best = int(obe.last_setting_index)
            util = None
        out["noise_var"].append(np.asarray(obe.yvar_noise_model(), dtype=np.float64).reshape(C))
        y_true = np.atleast_1d(np.asarray(fn(x, true_pars, cons), dtype=np.float64))
        y = y_true + sigma_meas * sim.standard_normal(C)
        rng.choices.clear()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            if cls == "base":
                rec = (x, tuple(y) if C > 1 else float(y[0]),
                       tuple([sigma_meas] * C) if C > 1 else sigma_meas)
            else:
                rec = (x, tuple(y) if C > 1 else float(y[0]))
            obe.pdf_update(rec)
        out["chosen_index"].append(best)
        out["y_meas"].append(y)
        out["utility"].append(np.asarray(util).reshape(-1)[:n_s] if util is not None else np.full(n_s, np.nan))
        out["draw_idx"].append(draw_idx)
        out["resampled"].append(bool(obe.just_resampled))
        out["mean"].append(obe.mean())
        out["std"].append(obe.std())
        out["cov"].append(obe.covariance())
        out["sum_w2"].append(np.sum(obe.particle_weights ** 2))
        if obe.just_resampled:
            if len(p_snaps) < max_particle_snaps:
                p_cycles.append(cyc)
                p_snaps.append(np.array(obe.particles, dtype=np.float64))
                resample_idx.append(rng.choices[0].copy())
        if max_weight_snaps is not None:
            keep = len(w_snaps) < max_weight_snaps - 1 and (cyc == 0 or obe.just_resampled) or cyc == n_cycles - 1
        else:
            keep = cyc < 3 or obe.just_resampled and len(w_snaps) < 8 or cyc == n_cycles - 1
        if keep:
            w_cycles.append(cyc)
            w_snaps.append(np.array(obe.particle_weights, dtype=np.float64))

    meta = dict(name=name, model=model, cls=cls, ctor=ctor, selection=selection,
                pickiness=pickiness, seed=seed, n_cycles=n_cycles,
                sigma_meas=sigma_meas, true_pars=[float(t) for t in true_pars],
                n_setdims=len(setting_values), n_channels=C,
                numpy=np.__version__, reference=ref.__version__)
    arrays = {k: np.array(v) for k, v in out.items()}
    arrays.update(prior=prior, cons=np.array(cons, dtype=np.float64),
                  w_cycles=np.array(w_cycles), w_snaps=np.array(w_snaps),
                  p_cycles=np.array(p_cycles),
                  p_snaps=np.array(p_snaps) if p_snaps else np.zeros((0, D, Np)),
                  resample_idx=np.array(resample_idx) if resample_idx else np.zeros((0, Np), dtype=np.int64),
                  meta=np.array(json.dumps(meta)))
    for k, sv in enumerate(setting_values):
        arrays[f"setval_{k}"] = np.asarray(sv, dtype=np.float64)
    path = os.path.join(HERE, f"traj_{name}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {n_cycles} cycles, {int(np.sum(out['resampled']))} resamples, "
          f"{os.path.getsize(path) / 1024:.0f} KiB")


def demo_size():
    """The reference's find_peak demo at its own size (demos/find_peak/sequentialLorentzian.py:
    200 settings, 50 000 particles, good_setting(pickiness=19), scale=False): large enough for the
    multi-chunk sweep kernel, the guided CDF search and the pipelined device-generator resample to be
    pinned to the real reference (indices exact)."""
    g = np.random.default_rng(20240425)
    run_trajectory("lorentz3_demo", "lorentzian", (np.linspace(1.5, 4.5, 200),), lorentz_prior(g, 50000), (0.1,),
                   (3.0, -1000.0, 50000.0), 500.0, 36, 808, ctor=dict(scale=False),
                   selection="good", pickiness=19, max_particle_snaps=1, max_weight_snaps=3)


def lorentz_prior(g, n):
    return np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])


def trajectories():
    g = np.random.default_rng(20240424)
    x64 = np.linspace(1.5, 4.5, 64)

    # A. find_peak-style Lorentzian, demos' scale=False
    run_trajectory("lorentz3_opt", "lorentzian", (x64,), lorentz_prior(g, 4096), (0.1,),
                   (3.0, -1000.0, 50000.0), 100.0, 40, 101, ctor=dict(scale=False))
    # B. class defaults (scale=True) + choke + non-default a_param and n_draws
    run_trajectory("lorentz3_scale_choke", "lorentzian", (x64,), lorentz_prior(g, 3000), (0.1,),
                   (3.1, -1200.0, 50200.0), 100.0, 30, 202,
                   ctor=dict(choke=0.7, a_param=0.9, n_draws=17, default_noise_std=100.0))
    # C. good_setting selection
    run_trajectory("lorentz3_good", "lorentzian", (x64,), lorentz_prior(g, 2048), (0.1,),
                   (2.7, -800.0, 49500.0), 100.0, 25, 303, ctor=dict(scale=False),
                   selection="good", pickiness=19)
    # D. NoiseParameter, line model, sigma prior reaching down to ~0 so that a
    #    resample nudges some sigma <= 0 (obe_noiseparam.py:65-79)
    n = 2500
    prior = np.array([g.uniform(-1, 1, n), g.uniform(-1, 1, n), g.exponential(0.1, n)])
    run_trajectory("line_noiseparam", "line_mb", (np.linspace(0, 1, 101),), prior, (),
                   (0.4, -0.3, 0.05), 0.05, 40, 404, cls="noise",
                   ctor=dict(scale=False, noise_parameter_index=2))
    # E. two channels (coil), one noise parameter shared by both channels.  Normalised
    #    units (L, C ~ 1): with SI values (C ~ 1e-9 next to sigma ~ 1e2) the covariance is
    #    so badly scaled that the reference's SVD-based nudge amplifies last-bit
    #    differences of the covariance to ~1e-7 of the small parameters.
    n = 2048
    prior = np.array([g.uniform(0.9, 1.1, n), g.uniform(0.08, 0.12, n),
                      g.uniform(0.9, 1.1, n), g.exponential(0.3, n)])
    run_trajectory("coil_2ch_noise", "coil", (np.logspace(-1, 1, 60),), prior, (),
                   (1.0, 0.1, 1.0, 0.3), 0.3, 30, 505, cls="noise",
                   ctor=dict(scale=False, noise_parameter_index=(3, 3)))
    # F. two setting dimensions (pipulse), 21 x 17 grid, known sigma
    n = 2048
    prior = np.array([g.uniform(1.0, 6.0, n), g.uniform(-4, 4, n)])
    run_trajectory("rabi_2set", "rabi", (np.linspace(0.02, 1, 21), np.linspace(-10, 10, 17)),
                   prior, (100000.0, 0.01, 2.0), (3.0, 1.5), 300.0, 30, 606,
                   ctor=dict(scale=False, default_noise_std=300.0))
    # G. 7-Lorentzian sum, 10 parameters, NoiseParameter (config 5 in miniature).  The
    #    noise level keeps N_eff healthy: a filter that collapses every cycle has a
    #    rank-deficient covariance and the SVD nudge becomes LAPACK noise.
    n = 3072
    prior = np.vstack([g.uniform(2, 4, (7, n)), g.uniform(400, 2000, (1, n)),
                       g.normal(500, 1000, (1, n)), g.exponential(1000, (1, n))])
    run_trajectory("multilorentz7_noise", "multi_lorentzian_7", (np.linspace(1.5, 4.5, 48),),
                   prior, (0.1,), (2.2, 2.5, 2.8, 3.1, 3.4, 3.7, 3.9, 1000.0, 500.0, 1000.0),
                   1000.0, 24, 707, cls="noise", ctor=dict(scale=False, noise_parameter_index=9),
                   max_particle_snaps=2)


def yspace_utilities():
    """max_min (n_draws = 2 and 7), pseudo_utility (Ebrahimi window at 30 draws,
    van Es at 9), full_kld_utility."""
    g = np.random.default_rng(4711)
    x64 = np.linspace(1.5, 4.5, 64)
    for name, method, nd, seed in (("maxmin2", "max_min", 2, 811), ("maxmin7", "max_min", 7, 812),
                                   ("pseudo30", "pseudo_utility", 30, 813), ("pseudo9", "pseudo_utility", 9, 814),
                                   ("fullkld", "full_kld_utility", 30, 815)):
        run_trajectory(f"util_{name}", "lorentzian", (x64,), lorentz_prior(g, 2048), (0.1,),
                       (3.0, -1000.0, 50000.0), 100.0, 15, seed,
                       ctor=dict(scale=False, utility_method=method, n_draws=nd, default_noise_std=100.0),
                       max_particle_snaps=1)


def unit_cases():
    g = np.random.default_rng(777)
    arrays = {}

    # moments on random weighted clouds (a3-a5), D in {1, 3, 10}, N not a power of 2
    for d in (1, 3, 10):
        n = 3001
        x = g.normal(0, 1, (d, n)) * g.uniform(0.5, 50, (d, 1)) + g.uniform(-100, 100, (d, 1))
        w = g.exponential(1.0, n)
        w /= w.sum()
        pdf = ref.ParticlePDF(x)
        pdf.particle_weights = w
        arrays[f"mom{d}_x"], arrays[f"mom{d}_w"] = x, w
        arrays[f"mom{d}_mean"], arrays[f"mom{d}_cov"], arrays[f"mom{d}_std"] = \
            pdf.mean(), pdf.covariance(), pdf.std()

    # randdraw indices (a8) and one full resample (a9), both scale settings
    for tag, scale in (("s0", False), ("s1", True)):
        n, d = 5000, 3
        x = lorentz_prior(g, n)
        w = g.exponential(1.0, n) ** 3
        w /= w.sum()
        pdf = ref.ParticlePDF(x.copy(), scale=scale)
        pdf.particle_weights = w.copy()
        rng = RecordingRNG(4242)
        pdf.rng = rng
        draws = pdf.randdraw(30)
        arrays[f"rs_{tag}_x"], arrays[f"rs_{tag}_w"] = x, w
        arrays[f"rs_{tag}_draw_idx"], arrays[f"rs_{tag}_draws"] = rng.choices[0].copy(), draws
        rng.choices.clear()
        pdf.resample()
        arrays[f"rs_{tag}_resample_idx"] = rng.choices[0].copy()
        arrays[f"rs_{tag}_particles"] = np.array(pdf.particles)
        arrays[f"rs_{tag}_weights"] = np.array(pdf.particle_weights)

    # Bayes update incl. nan_to_num paths: inf and nan in the likelihood (a6, a7)
    n = 1000
    w = g.exponential(1.0, n)
    w /= w.sum()
    lik = g.exponential(1.0, n)
    lik[[3, 500]] = np.nan
    pdf = ref.ParticlePDF(np.zeros((1, n)), auto_resample=False)
    pdf.particle_weights = w.copy()
    with np.errstate(all="ignore"):
        pdf.bayesian_update(lik)
    arrays["bu_w"], arrays["bu_lik"], arrays["bu_out"] = w, lik, np.array(pdf.particle_weights)
    lik0 = np.zeros(n)
    pdf.particle_weights = w.copy()
    with np.errstate(all="ignore"):
        pdf.bayesian_update(lik0)          # 0/0 -> all-zero weights
    arrays["bu_zero_out"] = np.array(pdf.particle_weights)

    # test_zinference.py::test_infer scenario (5000 particles on a linspace)
    n = 5000
    xs = np.linspace(-5, 5, n)
    obe = ref.OptBayesExpt(models.first_parameter, (0,), (xs, np.ones(n)), (0,))
    obe.tuning_parameters["resample_threshold"] = 0
    obe.pdf_update(((), 1.0, 1.0))
    arrays["infer_x"], arrays["infer_w"] = xs, np.array(obe.particle_weights)

    # likelihood + choke on a (2, N) model output
    n = 777
    ym = g.normal(0, 2, (2, n))
    obe2 = ref.OptBayesExpt(models.coil, (np.logspace(4, 6, 5),),
                            np.abs(g.normal(1, 0.1, (4, n))), (), choke=0.6)
    arrays["lk_ym"] = ym
    arrays["lk_out"] = obe2.likelihood(ym, ((1.0,), (0.3, -0.2), (1.5, 0.7)))

    np.savez_compressed(os.path.join(HERE, "unit_cases.npz"), **arrays)
    print("unit_cases: written")


class FullSweepRef(ref.OptBayesExpt):
    """Reference driven in the N_DRAWS -> all-particles limit: every
    particle is returned as a draw; valid as a *weighted* variance only while the
    weights are uniform, which is the state these fixtures use."""

    def randdraw(self, n_draws=1):
        if n_draws == self.n_particles:
            return np.array(self.particles, dtype=np.float64)
        return super().randdraw(n_draws)


class FullSweepRefNoise(ref.OptBayesExptNoiseParameter):
    def randdraw(self, n_draws=1):
        if n_draws == self.n_particles:
            return np.array(self.particles, dtype=np.float64)
        return super().randdraw(n_draws)


def full_sweep_cases():
    g = np.random.default_rng(31337)
    arrays = {}
    n = 2048
    x48 = np.linspace(1.5, 4.5, 48)
    prior = lorentz_prior(g, n)
    obe = FullSweepRef(models.lorentzian, (x48,), prior.copy(), (0.1,), n_draws=n,
                       default_noise_std=500.0)
    arrays["fs_lor_prior"], arrays["fs_lor_x"] = prior, x48
    arrays["fs_lor_yvar"] = obe.yvar_from_parameter_draws()
    arrays["fs_lor_utility"] = obe.utility()

    prior = np.vstack([g.uniform(2, 4, (7, n)), g.uniform(400, 2000, (1, n)),
                       g.normal(500, 1000, (1, n)), g.exponential(500, (1, n))])
    obe = FullSweepRefNoise(models.multi_lorentzian(7), (x48,), prior.copy(), (0.1,),
                            n_draws=n, noise_parameter_index=9)
    arrays["fs_ml7_prior"] = prior
    arrays["fs_ml7_yvar"] = obe.yvar_from_parameter_draws()
    arrays["fs_ml7_utility"] = obe.utility()

    n = 1024
    prior = np.array([g.uniform(0.9, 1.1, n), g.uniform(0.08, 0.12, n),
                      g.uniform(0.9, 1.1, n), g.exponential(0.3, n)])
    wset = np.logspace(-1, 1, 40)
    obe = FullSweepRefNoise(models.coil, (wset,), prior.copy(), (), n_draws=n,
                            noise_parameter_index=(3, 3))
    arrays["fs_coil_prior"], arrays["fs_coil_w"] = prior, wset
    arrays["fs_coil_yvar"] = obe.yvar_from_parameter_draws()
    arrays["fs_coil_utility"] = obe.utility()

    prior = np.array([g.uniform(1.0, 6.0, n), g.uniform(-4, 4, n)])
    sv = (np.linspace(0.02, 1, 13), np.linspace(-10, 10, 9))
    obe = FullSweepRef(models.rabi, sv, prior.copy(), (100000.0, 0.01, 2.0), n_draws=n,
                       default_noise_std=300.0)
    arrays["fs_rabi_prior"], arrays["fs_rabi_s0"], arrays["fs_rabi_s1"] = prior, sv[0], sv[1]
    arrays["fs_rabi_yvar"] = obe.yvar_from_parameter_draws()
    arrays["fs_rabi_utility"] = obe.utility()

    np.savez_compressed(os.path.join(HERE, "full_sweep_uniform.npz"), **arrays)
    print("full_sweep_uniform: written")


class IntegerWeightsMixin:
    """The real reference as the oracle of the WEIGHTED full sweep: ``randdraw``
    returns particle i duplicated k_i times (integer multiplicities, zeros included), so the
    reference's own ``np.var`` over the N_DRAWS = sum(k) model outputs (obe_base.py:463-489) IS the
    weighted variance with w_i = k_i / sum(k) — exactly, not statistically."""
    multiplicity = None

    def randdraw(self, n_draws=1):
        if self.multiplicity is not None and n_draws == int(self.multiplicity.sum()):
            return np.repeat(np.array(self.particles, dtype=np.float64), self.multiplicity, axis=1)
        return super().randdraw(n_draws)


class IntegerWeightsRef(IntegerWeightsMixin, ref.OptBayesExpt):
    pass


class IntegerWeightsRefNoise(IntegerWeightsMixin, ref.OptBayesExptNoiseParameter):
    pass


def full_sweep_integer_weights():
    g = np.random.default_rng(27182)
    arrays = {}

    def case(tag, cls, model, sv, prior, cons, **ctor):
        n = prior.shape[1]
        k = g.integers(0, 8, n)                     # k_i in [0, 7], about one particle in eight unweighted
        k[g.integers(0, n, 5)] = 0
        obe = cls(model, sv, prior.copy(), cons, n_draws=int(k.sum()), **ctor)
        obe.multiplicity = k
        obe.particle_weights = k / k.sum()          # (the noise-parameter class weights its sigma^2 with these)
        arrays[f"iw_{tag}_prior"], arrays[f"iw_{tag}_k"] = prior, k
        arrays[f"iw_{tag}_yvar"] = obe.yvar_from_parameter_draws()
        arrays[f"iw_{tag}_utility"] = obe.utility()
        print(f"integer weights {tag}: {n} particles, {int(k.sum())} draws, {int((k == 0).sum())} with k = 0")

    n = 2048
    x48 = np.linspace(1.5, 4.5, 48)
    arrays["iw_x48"] = x48
    case("lor", IntegerWeightsRef, models.lorentzian, (x48,), lorentz_prior(g, n), (0.1,), default_noise_std=500.0)
    # a narrowed cloud far from zero mean (kappa ~ 1e4: the regime of the shifted variance)
    narrow = np.array([g.normal(3.0, 0.004, n), g.normal(-1000.0, 8.0, n), g.normal(50000.0, 15.0, n)])
    case("lornarrow", IntegerWeightsRef, models.lorentzian, (x48,), narrow, (0.1,), default_noise_std=500.0)
    prior = np.vstack([g.uniform(2, 4, (7, n)), g.uniform(400, 2000, (1, n)),
                       g.normal(500, 1000, (1, n)), g.exponential(500, (1, n))])
    case("ml7", IntegerWeightsRefNoise, models.multi_lorentzian(7), (x48,), prior, (0.1,), noise_parameter_index=9)
    n = 1024
    prior = np.array([g.uniform(0.9, 1.1, n), g.uniform(0.08, 0.12, n),
                      g.uniform(0.9, 1.1, n), g.exponential(0.3, n)])
    wset = np.logspace(-1, 1, 40)
    arrays["iw_coil_w"] = wset
    case("coil", IntegerWeightsRefNoise, models.coil, (wset,), prior, (), noise_parameter_index=(3, 3))
    prior = np.array([g.uniform(1.0, 6.0, n), g.uniform(-4, 4, n)])
    sv = (np.linspace(0.02, 1, 13), np.linspace(-10, 10, 9))
    arrays["iw_rabi_s0"], arrays["iw_rabi_s1"] = sv
    case("rabi", IntegerWeightsRef, models.rabi, sv, prior, (100000.0, 0.01, 2.0), default_noise_std=300.0)
    np.savez_compressed(os.path.join(HERE, "full_sweep_integer_weights.npz"), **arrays)
    print("full_sweep_integer_weights: written")


def state_reset_10_parameters():
    """The 10-parameter noise-parameter trajectory (config 5 in miniature) once more, shorter, with the
    COMPLETE state before every cycle (particles, weights, generator state) and everything the cycle
    produces: a replay can start every cycle from the reference's own state, so that one step is
    compared at 1e-10 although the free-running trajectory drifts (the SVD nudge of resample()
    amplifies last-bit differences of the covariance: DESIGN.md section 5)."""
    g = np.random.default_rng(20240426)
    n, n_cycles, seed = 2048, 12, 717
    fn = MODELS["multi_lorentzian_7"]
    prior = np.vstack([g.uniform(2, 4, (7, n)), g.uniform(400, 2000, (1, n)),
                       g.normal(500, 1000, (1, n)), g.exponential(1000, (1, n))])
    sv = (np.linspace(1.5, 4.5, 48),)
    cons = (0.1,)
    true_pars = (2.2, 2.5, 2.8, 3.1, 3.4, 3.7, 3.9, 1000.0, 500.0, 1000.0)
    ctor = dict(scale=False, noise_parameter_index=9)
    obe = ref.OptBayesExptNoiseParameter(fn, sv, prior.copy(), cons, **ctor)
    rng = RecordingRNG(seed)
    obe.rng = rng
    sim = np.random.default_rng(seed + 1)
    out = dict(particles_before=[], weights_before=[], rng_state_before=[], rng_inc=[], draw_idx=[], utility=[],
               noise_var=[], chosen_index=[], y_meas=[], resampled=[], resample_idx=[], particles_after=[],
               weights_after=[], mean=[], std=[], cov=[], n_constrained=[])
    for cyc in range(n_cycles):
        st = rng._g.bit_generator.state
        assert st["bit_generator"] == "PCG64" and st["has_uint32"] == 0
        out["particles_before"].append(np.array(obe.particles, dtype=np.float64))
        out["weights_before"].append(np.array(obe.particle_weights, dtype=np.float64))
        s128, i128 = int(st["state"]["state"]), int(st["state"]["inc"])
        out["rng_state_before"].append([s128 >> 64, s128 & (2 ** 64 - 1)])
        out["rng_inc"].append([i128 >> 64, i128 & (2 ** 64 - 1)])
        rng.choices.clear()
        util = obe.utility()
        out["draw_idx"].append(rng.choices[0].copy())
        best = int(np.argmax(util))
        obe.last_setting_index = best
        x = tuple(obe.allsettings[:, best])
        out["noise_var"].append(np.asarray(obe.yvar_noise_model(), dtype=np.float64).reshape(-1))
        y = float(fn(x, true_pars, cons)) + 1000.0 * sim.standard_normal()
        rng.choices.clear()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            obe.pdf_update((x, y))
        out["utility"].append(np.asarray(util).reshape(-1))
        out["chosen_index"].append(best)
        out["y_meas"].append(y)
        out["resampled"].append(bool(obe.just_resampled))
        out["resample_idx"].append(rng.choices[0].copy() if obe.just_resampled else np.zeros(n, dtype=np.int64))
        out["particles_after"].append(np.array(obe.particles, dtype=np.float64))
        out["weights_after"].append(np.array(obe.particle_weights, dtype=np.float64))
        out["n_constrained"].append(int(np.sum(np.asarray(obe.particle_weights) == 0.0)) if obe.just_resampled else 0)
        out["mean"].append(obe.mean())
        out["std"].append(obe.std())
        out["cov"].append(obe.covariance())
    meta = dict(name="state10", model="multi_lorentzian_7", cls="noise", ctor=ctor, seed=seed, n_cycles=n_cycles,
                true_pars=list(true_pars), numpy=np.__version__, reference=ref.__version__)
    arrays = {k: np.array(v, dtype=np.uint64 if k.startswith("rng_") else None) for k, v in out.items()}
    arrays.update(prior=prior, cons=np.array(cons), setval_0=sv[0], meta=np.array(json.dumps(meta)))
    path = os.path.join(HERE, "state_multilorentz7_noise.npz")
    np.savez_compressed(path, **arrays)
    print(f"state10: {n_cycles} cycles, {int(np.sum(out['resampled']))} resamples, {os.path.getsize(path) / 1024:.0f} KiB")


def sweeper_cases():
    """the sweeper composition.  The reference's OptBayesExptSweeper lives in
    demos/sweeper/obe_sweeper.py (a subclass of OptBayesExptNoiseParameter); it is loaded
    from there, driven through seeded sweep cycles in both selection modes, and per cycle
    the sweep utility, the chosen (start, stop) pair, the simulated sweep and the posterior
    statistics are stored."""
    import importlib.util
    os.environ.setdefault("MPLBACKEND", "Agg")
    spec = importlib.util.spec_from_file_location("ref_obe_sweeper", "/root/reference/demos/sweeper/obe_sweeper.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.random.default_rng(90210)
    xvals = np.linspace(1.5, 4.5, 100)                     # demos/sweeper/sweeper.py:69
    cons = (0.1,)
    true_pars = (3.1, 1200.0, 300.0, 800.0)
    for name, selection, n_cycles, seed in (("sweeper_opt", "optimal", 10, 911), ("sweeper_good", "good", 10, 912)):
        n = 4096
        prior = np.array([g.uniform(2, 4, n), g.uniform(400, 2000, n), g.normal(500, 1000, n),
                          g.exponential(500, n)])      # sweeper.py:74-83
        ctor = dict(scale=False, utility_method="variance_approx", selection_method=selection, pickiness=20,
                    noise_parameter_index=3)
        obe = mod.OptBayesExptSweeper(models.lorentzian, (xvals,), prior.copy(), cons, **ctor)
        rng = RecordingRNG(seed)
        obe.rng = rng
        mod.rng = RecordingRNG(seed + 3)                   # that module's own generator (good_setting)
        sim = np.random.default_rng(seed + 1)
        out = dict(chosen_index=[], pair=[], sweep_utility=[], draw_idx=[], n_resamples=[], mean=[], std=[],
                   cov=[], sum_w2=[])
        ys, w_snaps = [], []
        for cyc in range(n_cycles):
            rng.choices.clear()
            if selection == "optimal":
                util = obe.sweep_utility()                 # what opt_setting() maximises (obe_sweeper.py:163)
                index = int(np.argmax(util))
                obe.last_setting_index = index
                pair = obe.start_stop_indices[index]
            else:
                pair = obe.get_setting()
                index = int(obe.last_setting_index)
                util = np.full(len(obe.start_stop_indices), np.nan)
            draw_idx = rng.choices[0].copy()
            start, stop = int(pair[0]), int(pair[1])
            sweep_x = xvals[start:stop]                    # sweeper.py:129
            y = models.lorentzian((sweep_x,), true_pars, cons) + true_pars[3] * sim.standard_normal(len(sweep_x))
            n_res = 0
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                # count the resamples inside the sweep: one full-size choice() each
                rng.choices.clear()
                obe.pdf_update(((sweep_x,), y))
                n_res = sum(1 for c in rng.choices if len(c) == n)
            out["chosen_index"].append(index)
            out["pair"].append([start, stop])
            out["sweep_utility"].append(util)
            out["draw_idx"].append(draw_idx)
            out["n_resamples"].append(n_res)
            out["mean"].append(obe.mean())
            out["std"].append(obe.std())
            out["cov"].append(obe.covariance())
            out["sum_w2"].append(np.sum(obe.particle_weights ** 2))
            ys.append(y)
            if cyc in (0, n_cycles - 1):
                w_snaps.append(np.array(obe.particle_weights))
        meta = dict(name=name, model="lorentzian", cls="sweeper", ctor=ctor, seed=seed, n_cycles=n_cycles,
                    true_pars=list(true_pars), numpy=np.__version__, reference=ref.__version__)
        arrays = {k: np.array(v) for k, v in out.items()}
        arrays.update(prior=prior, cons=np.array(cons), setval_0=xvals, y_concat=np.concatenate(ys),
                      w_snaps=np.array(w_snaps), pairs=np.array(obe.start_stop_indices),
                      sweep_cost=np.asarray(obe.sweep_cost_estimate(), dtype=np.float64),
                      meta=np.array(json.dumps(meta)))
        path = os.path.join(HERE, f"traj_{name}.npz")
        np.savez_compressed(path, **arrays)
        print(f"{name}: {n_cycles} sweeps, {len(arrays['y_concat'])} points, "
              f"{int(np.sum(out['n_resamples']))} resamples, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    which = sys.argv[1:] or ["trajectories", "yspace", "units", "full_sweep", "sweeper"]
    steps = dict(trajectories=trajectories, yspace=yspace_utilities, units=unit_cases,
                 full_sweep=full_sweep_cases, sweeper=sweeper_cases, demo_size=demo_size,
                 integer_weights=full_sweep_integer_weights, state10=state_reset_10_parameters)
    for w in which:
        steps[w]()
