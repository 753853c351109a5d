"""Replay helpers shared by the oracle tests (CPU) and the HIP parity tests (GPU).

A golden trajectory (tests/golden/traj_*.npz, produced from the real reference by
tests/golden/make_golden.py) is re-run through an implementation with the same
method surface as the reference classes, feeding it the recorded measurement values,
and every recorded output is compared:

* draw indices, resample indices, chosen setting index, resample flags: exact;
* utility, weights, moments, particles after a resample: relative tolerance ``rtol``
  (1e-10 for the HIP path — BASELINE.json north_star — tighter for the oracle).
"""
import json
import os
import warnings

import numpy as np
from numpy.testing import assert_allclose, assert_array_equal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TRAJECTORIES = ["lorentz3_opt", "lorentz3_scale_choke", "lorentz3_good",
                "line_noiseparam", "coil_2ch_noise", "rabi_2set", "multilorentz7_noise",
                "lorentz3_demo"]      # the last one at the reference demo's own size: 200 settings x 50 000 particles

# Relative tolerance of the HIP path per trajectory.  1e-10 is the bar of BASELINE.json.
# The 10-parameter trajectory is the exception: there the *reference itself* is only
# reproducible to ~1e-7 over 24 cycles — summing its covariance in a different (equally
# valid) order on the CPU already moves the utilities by 2e-10 after the first resample,
# 3e-9 by cycle 10 and 1e-7 by cycle 20
# (tests/test_oracle_golden.py::test_reference_conditioning_10_parameters), because the
# SVD-based nudge amplifies last-bit differences of a 10x10 covariance whose eigenvalues
# span six decades, and every resample compounds it.  No implementation with a
# different summation order can do better; integer outputs are still compared exactly.
UTILITY_TRAJECTORIES = ["util_maxmin2", "util_maxmin7", "util_pseudo30", "util_pseudo9", "util_fullkld"]

HIP_RTOL = {name: 1e-10 for name in TRAJECTORIES + UTILITY_TRAJECTORIES}
HIP_RTOL["multilorentz7_noise"] = 1e-6

# Absolute floor on particles right after a resample, in units of eps*sqrt(largest
# covariance eigenvalue): the nudge z @ (u sqrt(s)).T comes from an SVD whose entries
# carry LAPACK round-off of that size *regardless of the parameter's own scale*.
# Measured on the CPU by re-summing the reference's covariance in reverse order
# (same experiment as above): 5-75 units on the 3/4-parameter trajectories, 530 units at
# the first resample of the 10-parameter one, growing as the two runs drift apart.
NUDGE_FLOOR_UNITS = {name: 256 for name in TRAJECTORIES + UTILITY_TRAJECTORIES}
NUDGE_FLOOR_UNITS["multilorentz7_noise"] = 20000


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    fx = {k: z[k] for k in z.files}
    if "meta" in fx:
        fx["meta"] = json.loads(str(fx["meta"]))
    return fx


def load_traj(name):
    return load(f"traj_{name}.npz")


def setting_values(fx):
    return tuple(fx[f"setval_{k}"] for k in range(fx["meta"]["n_setdims"]))


def construct(fx, base_cls, noise_cls, model, extra=None):
    meta = fx["meta"]
    ctor = dict(meta["ctor"])
    if "noise_parameter_index" in ctor and isinstance(ctor["noise_parameter_index"], list):
        ctor["noise_parameter_index"] = tuple(ctor["noise_parameter_index"])
    ctor.update(extra or {})
    cls = base_cls if meta["cls"] == "base" else noise_cls
    cons = tuple(float(c) for c in fx["cons"])
    obe = cls(model, setting_values(fx), fx["prior"].copy(), cons, **ctor)
    obe.rng = np.random.default_rng(meta["seed"])        # seeding recipe, SURVEY §8c
    return obe


def noise_rng(fx):
    """The generator the fixture's run used as the reference's module-level obe_base.rng."""
    return np.random.default_rng(fx["meta"]["seed"] + 2)


WORST = {}          # what -> worst relative error seen by close() (printed by the trajectory tests)


def close(actual, desired, rtol, what, scale=None):
    """PURE relative comparison, element by element (round 6: the floor tied to the array's largest value is gone —
    a weight of 1e-200 or the utility of a setting far from the peak is held to the same ``rtol`` as the largest).
    Only values below 1e-290 in the reference (weights that have underflowed into the subnormal range, where a
    double has fewer than 53 bits) are exempt.  Records the worst relative error per kind of quantity in WORST."""
    err = assert_rel(actual, desired, rtol, what, garbage_floor=1e-290)
    kind = what.split(",")[0]
    WORST[kind] = max(WORST.get(kind, 0.0), err)


def close_weights(actual, desired, rtol, what):
    """Particle weights: w_i is exp(-sum_k z_ik^2 / 2) / norm (obe_base.py:451-461, particlepdf.py:136-139), so what an
    implementation is good to is the EXPONENT: held to ``rtol`` relative it leaves the weight with a relative error of
    rtol * |ln(w_i / w_max)| — 1e-10 for the particles that carry the posterior, 1e-8 for a weight of 1e-48 whose
    exponent is ~100 (after a resample the model outputs behind z already differ by the ~1e-9 that the reference's
    own SVD nudge round-off, NUDGE_FLOOR_UNITS, moves them: measured 1.2e-10 on such a weight, util_maxmin2 cycle 2).
    Pure relative, per element, no floor tied to the largest weight; subnormal reference weights exempt."""
    desired = np.asarray(desired, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        depth = np.abs(np.log(desired / np.max(desired)))
    tol = rtol * np.maximum(1.0, np.where(np.isfinite(depth), depth, 1.0))
    err = assert_rel(actual, desired, tol, what, garbage_floor=1e-290)
    WORST["weights"] = max(WORST.get("weights", 0.0), err)


def close_cov(actual, desired, mean, std, rtol, what):
    """A covariance matrix: entry (i, j) is <x_i x_j> - <x_i><x_j> (particlepdf.py:185-198, np.cov with aweights
    subtracts the mean first; the device forms centred products as well), and an off-diagonal entry can be zero by
    cancellation — so the tolerance per entry is rtol * sd_i sd_j (the scale of the entry's own terms) plus the
    one-pass term 64 eps |m_i m_j| that std() carries, never a fraction of the LARGEST entry of the matrix."""
    actual, desired = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
    mean, std = np.asarray(mean, dtype=np.float64), np.asarray(std, dtype=np.float64)
    tol = rtol * np.maximum(np.abs(desired), np.outer(std, std)) + 64 * 2.3e-16 * np.abs(np.outer(mean, mean))
    err = np.abs(actual - desired)
    k = np.unravel_index(int(np.argmax(err / np.maximum(tol, 1e-300))), err.shape)
    assert np.all(err <= tol), f"{what}: entry {k}: error {err[k]:.3g} > tolerance {tol[k]:.3g} (value {desired[k]:.6g})"
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.max(err / np.maximum(np.abs(desired), np.outer(std, std)))
    WORST["covariance"] = max(WORST.get("covariance", 0.0), float(rel))


def utility_rtol(fx, want_u, rtol, spread_ulps=64):
    """The relative tolerance of a recorded utility vector, per element: ``rtol`` — except for utility_max_min
    (obe_base.py:520-535, 602-626), whose value is ((max - min of the model outputs over the draws) / sigma_n)^2: a
    difference of outputs y ~ 5e4.  Each output is good to about an ulp in any implementation, and after a resample
    the particles themselves carry the absolute round-off of the reference's SVD nudge (NUDGE_FLOOR_UNITS below),
    which moves an output by a few ulp more; where the draws nearly agree, that — not the device — limits the
    utility.  So the spread max - min = sigma_n sqrt(u) is held to ``spread_ulps`` ulp(y) in ABSOLUTE terms (measured
    worst: 13 ulp, util_maxmin2 cycle 2, identical with the model evaluated by NumPy on the host), i.e. the utility to
    2 x that / spread relative — a floor in the natural variable of the quantity, computed from the fixture alone
    (|y| <= largest |amplitude| + |background| of the prior), never from the largest utility of the vector.
    1e-10 for every other utility, and for every max_min element whose spread is larger than ~1e4 ulp(y)."""
    meta = fx["meta"]
    if meta["ctor"].get("utility_method") != "max_min":
        return rtol
    y_scale = float(np.sum(np.max(np.abs(fx["prior"]), axis=1)[-2:])) if meta["model"] == "lorentzian" else \
        float(np.max(np.abs(fx["prior"])))
    sigma_n = float(meta["ctor"].get("default_noise_std", 1.0))
    with np.errstate(divide="ignore", invalid="ignore"):
        spread = sigma_n * np.sqrt(np.abs(want_u))
        cond = np.where(spread > 0.0, 2.0 * spread_ulps * np.spacing(y_scale) / spread, np.inf)
    return np.maximum(rtol, cond)


def replay(fx, obe, rtol, get_draw_idx=None, get_utility=None, check_moments=True, floor_units=256):
    """Drive ``obe`` through the recorded cycles.  ``get_draw_idx(obe)`` returns the
    particle indices of the most recent utility draw (implementation-specific
    accessor); ``get_utility(obe)`` the most recent utility vector."""
    meta = fx["meta"]
    C = meta["n_channels"]
    n_cycles = meta["n_cycles"]
    w_at = {int(c): i for i, c in enumerate(fx["w_cycles"])}
    p_at = {int(c): i for i, c in enumerate(fx["p_cycles"])}
    stats = dict(cycles=0, resamples=0, utility_err=[])     # utility_err: worst relative error of the utility per cycle
    for cyc in range(n_cycles):
        if meta["selection"] == "opt":
            x = obe.opt_setting()
            if get_utility is not None:
                u, want_u = np.asarray(get_utility(obe), dtype=np.float64).reshape(-1), fx["utility"][cyc].reshape(-1)
                with np.errstate(divide="ignore", invalid="ignore"):
                    stats["utility_err"].append(float(np.nanmax(np.abs(u - want_u) / np.abs(want_u))))
                close(u, want_u, utility_rtol(fx, want_u, rtol), f"utility, cycle {cyc}")
        else:
            x = obe.good_setting(meta["pickiness"])
        if get_draw_idx is not None:
            assert_array_equal(np.asarray(get_draw_idx(obe)), fx["draw_idx"][cyc],
                               err_msg=f"draw indices, cycle {cyc}")
        assert int(obe.last_setting_index) == int(fx["chosen_index"][cyc]), \
            f"chosen setting index, cycle {cyc}"
        assert_array_equal(np.asarray(x, dtype=np.float64),
                           np.asarray(obe.allsettings)[:, int(fx["chosen_index"][cyc])])
        y = fx["y_meas"][cyc]
        if meta["cls"] == "base":
            s = meta["sigma_meas"]
            rec = (x, tuple(y) if C > 1 else float(y[0]), tuple([s] * C) if C > 1 else s)
        else:
            rec = (x, tuple(y) if C > 1 else float(y[0]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            obe.pdf_update(rec)
        assert bool(obe.just_resampled) == bool(fx["resampled"][cyc]), f"resample flag, cycle {cyc}"
        stats["resamples"] += int(obe.just_resampled)
        if cyc in w_at:
            close_weights(obe.particle_weights, fx["w_snaps"][w_at[cyc]], rtol, f"weights, cycle {cyc}")
        if cyc in p_at:
            snap = fx["p_snaps"][p_at[cyc]]
            got = np.asarray(obe.particles, dtype=np.float64)
            # The nudge is z @ (u sqrt(s)).T from an SVD of the covariance: its entries carry
            # an absolute LAPACK round-off of ~eps*sqrt(s_max) whatever the parameter's own
            # scale, so that is the floor below which the reference itself is noise.
            floor = floor_units * 2.3e-16 * np.sqrt(np.max(np.diag(fx["cov"][cyc])))
            for d in range(snap.shape[0]):
                assert_allclose(got[d], snap[d], rtol=rtol, atol=floor,
                                err_msg=f"particles[{d}] after resample, cycle {cyc}")
        if check_moments:
            # a mean is known to within a fraction of the spread: tolerance rtol*(|mean|+std)
            mtol = rtol * (np.abs(fx["mean"][cyc]) + fx["std"][cyc])
            merr = np.abs(np.asarray(obe.mean()) - fx["mean"][cyc])
            assert np.all(merr <= mtol), f"mean, cycle {cyc}: err {merr} tol {mtol}"
            sd = fx["std"][cyc]
            # std() is the one-pass <x^2> - <x>^2 (particlepdf.py:209-214): its
            # attainable accuracy is ~eps * <x^2> / var relative, whatever the
            # summation order, so that term is added to the tolerance.
            tol = rtol * sd + 64 * 2.3e-16 * fx["mean"][cyc] ** 2 / np.maximum(sd, 1e-300)
            err = np.abs(np.asarray(obe.std()) - sd)
            assert np.all(err <= tol), f"std, cycle {cyc}: err {err} tol {tol}"
            close_cov(obe.covariance(), fx["cov"][cyc], fx["mean"][cyc], sd, rtol, f"covariance, cycle {cyc}")
        close(np.sum(np.asarray(obe.particle_weights) ** 2), fx["sum_w2"][cyc], rtol,
              f"sum w^2, cycle {cyc}")
        stats["cycles"] += 1
    return stats


SWEEPER_TRAJECTORIES = ["sweeper_opt", "sweeper_good"]


def replay_sweeper(fx, obe, rtol, set_sweep_rng, get_draw_idx=None, get_utility=None):
    """Drive a sweeper object through the recorded sweeps: per cycle the sweep
    utility over all (start, stop) pairs, the chosen pair, one pdf_update per point of the
    simulated sweep, then the posterior statistics.  ``set_sweep_rng(generator)`` installs
    the generator that stands for the reference demo module's own ``rng``."""
    meta = fx["meta"]
    xvals = fx["setval_0"]
    obe.rng = np.random.default_rng(meta["seed"])
    set_sweep_rng(np.random.default_rng(meta["seed"] + 3))
    assert_array_equal(np.asarray(obe.start_stop_indices), fx["pairs"])
    assert_array_equal(np.asarray(obe.sweep_cost_estimate(), dtype=np.float64), fx["sweep_cost"])
    optimal = meta["ctor"]["selection_method"] == "optimal"
    pos, points = 0, 0
    for cyc in range(meta["n_cycles"]):
        pair = obe.get_setting()
        if optimal and get_utility is not None:
            close(get_utility(obe), fx["sweep_utility"][cyc], rtol, f"sweep utility, cycle {cyc}")
        if get_draw_idx is not None:
            assert_array_equal(np.asarray(get_draw_idx(obe)), fx["draw_idx"][cyc], err_msg=f"draw indices, cycle {cyc}")
        assert int(obe.last_setting_index) == int(fx["chosen_index"][cyc]), f"chosen pair index, cycle {cyc}"
        assert_array_equal(np.asarray(pair), fx["pair"][cyc])
        start, stop = int(pair[0]), int(pair[1])
        sweep_x = xvals[start:stop]
        y = fx["y_concat"][pos:pos + len(sweep_x)]
        pos += len(sweep_x)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            obe.pdf_update(((sweep_x,), y))
        points += len(sweep_x)
        mtol = rtol * (np.abs(fx["mean"][cyc]) + fx["std"][cyc])
        merr = np.abs(np.asarray(obe.mean()) - fx["mean"][cyc])
        assert np.all(merr <= mtol), f"mean, cycle {cyc}: err {merr} tol {mtol}"
        sd = fx["std"][cyc]
        tol = rtol * sd + 64 * 2.3e-16 * fx["mean"][cyc] ** 2 / np.maximum(sd, 1e-300)
        err = np.abs(np.asarray(obe.std()) - sd)
        assert np.all(err <= tol), f"std, cycle {cyc}: err {err} tol {tol}"
        close_cov(obe.covariance(), fx["cov"][cyc], fx["mean"][cyc], sd, rtol, f"covariance, cycle {cyc}")
        close(np.sum(np.asarray(obe.particle_weights) ** 2), fx["sum_w2"][cyc], rtol, f"sum w^2, cycle {cyc}")
        if cyc == 0:
            close_weights(obe.particle_weights, fx["w_snaps"][0], rtol, "weights, cycle 0")
    close_weights(obe.particle_weights, fx["w_snaps"][-1], rtol, "weights, last cycle")
    assert pos == len(fx["y_concat"])
    return dict(cycles=meta["n_cycles"], points=points)


def generator_from_words(state_words, inc_words):
    """A PCG64 Generator in the recorded state ([hi, lo] 64-bit halves of the 128-bit state and increment)."""
    g = np.random.Generator(np.random.PCG64())
    st = g.bit_generator.state
    st["state"]["state"] = (int(state_words[0]) << 64) | int(state_words[1])
    st["state"]["inc"] = (int(inc_words[0]) << 64) | int(inc_words[1])
    st["has_uint32"], st["uinteger"] = 0, 0
    g.bit_generator.state = st
    return g


def replay_state_reset(fx, obe, rtol, get_draw_idx, get_utility, get_resample_idx, particle_floor_units=256):
    """tests/golden/state_multilorentz7_noise.npz: EVERY cycle starts from the reference's own recorded
    state (particles, weights, generator), so one opt_setting + pdf_update step is compared with the
    reference's at ``rtol`` without the drift of a free-running 10-parameter trajectory.  Integers
    (draws, chosen setting, resample decision, resample indices, constrained particles) exact."""
    meta = fx["meta"]
    n_res = 0
    worst = dict(utility=0.0, weights=0.0, particles=0.0)
    for cyc in range(meta["n_cycles"]):
        obe.set_pdf(fx["particles_before"][cyc].copy())
        obe.particle_weights = fx["weights_before"][cyc].copy()
        obe.parameters = obe.particles                     # (set_pdf leaves the alias stale: obe_base.py:185,395)
        obe.rng = generator_from_words(fx["rng_state_before"][cyc], fx["rng_inc"][cyc])
        x = obe.opt_setting()
        assert_array_equal(np.asarray(get_draw_idx(obe)), fx["draw_idx"][cyc], err_msg=f"draw indices, cycle {cyc}")
        u = np.asarray(get_utility(obe), dtype=np.float64).reshape(-1)
        close(u, fx["utility"][cyc], rtol, f"utility, cycle {cyc}")
        worst["utility"] = max(worst["utility"], float(np.max(np.abs(u / fx["utility"][cyc] - 1))))
        assert int(obe.last_setting_index) == int(fx["chosen_index"][cyc]), f"chosen setting, cycle {cyc}"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            obe.pdf_update((x, float(fx["y_meas"][cyc])))
        assert bool(obe.just_resampled) == bool(fx["resampled"][cyc]), f"resample decision, cycle {cyc}"
        w = np.asarray(obe.particle_weights, dtype=np.float64)
        want_w = fx["weights_after"][cyc]
        close_weights(w, want_w, rtol, f"weights, cycle {cyc}")
        assert_array_equal(w == 0.0, want_w == 0.0, err_msg=f"zero weights, cycle {cyc}")
        worst["weights"] = max(worst["weights"], float(np.max(np.abs(w - want_w)) / np.max(want_w)))
        if fx["resampled"][cyc]:
            n_res += 1
            assert_array_equal(np.asarray(get_resample_idx(obe)), fx["resample_idx"][cyc],
                               err_msg=f"resample indices, cycle {cyc}")
            assert int(np.sum(w == 0.0)) == int(fx["n_constrained"][cyc])
            got, want = np.asarray(obe.particles, dtype=np.float64), fx["particles_after"][cyc]
            # absolute floor of the SVD nudge (see replay()): eps * sqrt(largest eigenvalue of the covariance)
            pre_cov = np.cov(fx["particles_before"][cyc], aweights=fx["weights_before"][cyc])
            floor = particle_floor_units * 2.3e-16 * np.sqrt(np.max(np.linalg.eigvalsh(pre_cov)))
            for d in range(want.shape[0]):
                assert_allclose(got[d], want[d], rtol=rtol, atol=floor, err_msg=f"particles[{d}] after resample, cycle {cyc}")
            worst["particles"] = max(worst["particles"], float(np.max(np.abs(got - want)) / floor * particle_floor_units))
        mtol = rtol * (np.abs(fx["mean"][cyc]) + fx["std"][cyc])
        assert np.all(np.abs(np.asarray(obe.mean()) - fx["mean"][cyc]) <= mtol), f"mean, cycle {cyc}"
        sd = fx["std"][cyc]
        tol = rtol * sd + 64 * 2.3e-16 * fx["mean"][cyc] ** 2 / np.maximum(sd, 1e-300)
        assert np.all(np.abs(np.asarray(obe.std()) - sd) <= tol), f"std, cycle {cyc}"
        close_cov(obe.covariance(), fx["cov"][cyc], fx["mean"][cyc], sd, rtol, f"covariance, cycle {cyc}")
    return dict(cycles=meta["n_cycles"], resamples=n_res, worst=worst)



def assert_rel(got, ref, rtol=1e-10, what="", garbage_floor=0.0):
    """Pure RELATIVE comparison of variances / utilities, element by element (no floor tied to
    the largest reference value, so a setting whose variance is 1e-6 of the peak's is held to the same 1e-10
    as the peak).  The failure message names the worst element.

    ``garbage_floor`` (default: none) is an ABSOLUTE allowance that applies ONLY to elements whose reference
    value is itself below it — the case of identical draws (one particle, or 30 draws that all hit the same
    particle), where the reference's two-pass np.var is not 0 but (eps * y)^2-sized rounding debris (~1e-22
    for y ~ 5e4) and the one-pass device variance is exactly 0 or debris of the same size: there is nothing
    to compare relatively.  Callers that pass it say why."""
    import numpy as np
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), f"{what}: NaN pattern differs"
    live = ~np.isnan(ref) & (np.abs(ref) > garbage_floor)
    err = np.zeros(ref.shape)
    with np.errstate(divide="ignore", invalid="ignore"):
        err[live] = np.abs(got[live] - ref[live]) / np.abs(ref[live])
    err[live & (got == ref)] = 0.0                     # (inf == inf, 0 == 0)
    if live.any():
        tol = np.broadcast_to(np.asarray(rtol, dtype=np.float64), err.shape)      # (a scalar, or one per element)
        k = np.unravel_index(int(np.argmax(np.where(live, err / tol, 0.0))), err.shape)
        assert err[k] <= tol[k], (f"{what}: worst relative error {err[k]:.3g} > {tol[k]:g} at element {k}: got "
                                  f"{got[k]!r}, reference {ref[k]!r} (largest reference value {np.nanmax(np.abs(ref)):.6g})")
    dead = ~np.isnan(ref) & ~live
    if dead.any():
        assert np.all(np.abs(got[dead]) <= garbage_floor), \
            f"{what}: {int(np.sum(np.abs(got[dead]) > garbage_floor))} value(s) above the debris floor {garbage_floor:g} where the reference is below it"
    return float(err.max()) if live.any() else 0.0


def conditioning_rtol(ref_var, ref_mean, n_eff, floor=1e-10, eps_y=2.3e-16):
    """Per-element relative tolerance for a variance whose samples y are each rounded to eps_y |y|: against a
    spread of sqrt(var) that is 2 eps_y |mean y| / sqrt(var) of the variance per sample, averaged down by the
    square root of the effective number of samples — what the REFERENCE's own two-pass np.var is good to when
    (mean y)^2 / var is large (a converged posterior on a 5e4 background).  Never below ``floor``; x4 margin.
    Computed from the oracle's mean and variance, not from anything the device reports."""
    import numpy as np
    ref_var, ref_mean = np.asarray(ref_var, dtype=np.float64), np.asarray(ref_mean, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        cond = np.where(ref_var > 0.0, np.abs(ref_mean) / np.sqrt(ref_var), 0.0)
    return np.maximum(floor, 4.0 * 2.0 * eps_y * cond / np.sqrt(max(float(n_eff), 1.0)))
