"""TEST INFRASTRUCTURE — CPU restatement of the optbayesexpt hot path.

Every function names the reference lines (relative to /root/reference) whose
arithmetic it restates.  The restatement is deliberately written from the
*primitive* operations (prefix sum + binary search instead of
``Generator.choice``; SVD factor x standard normals instead of
``multivariate_normal``; the explicit ``np.cov(aweights=)`` formula) because
those primitives are what the HIP kernels implement; SURVEY.md Appendix B
records that each primitive form is bit-identical to the NumPy call the
reference makes (numpy 2.2.6), and tests/test_oracle_golden.py re-checks it
against fixtures produced by the real reference.

Parity status: pinned (see oracle/__init__.py).
"""
import warnings

import numpy as np

__all__ = [
    "normalized_product", "numpy_pairwise_sum", "effective_particles", "weighted_mean",
    "weighted_covariance", "weighted_std", "weight_cdf", "choice_indices",
    "systematic_indices", "nudge_factor", "gauss_likelihood", "yvar_from_draws", "yvar_full_sweep",
    "utility_from_yvar", "mean_noise_variance", "flatten_settings",
    "OracleParticlePDF", "OracleOptBayesExpt", "OracleOptBayesExptNoiseParameter",
    "OracleOptBayesExptSweeper",
]

DEFAULT_N_DRAWS = 30          # obe_base.py:19


# --------------------------------------------------------------------------
# ParticlePDF arithmetic
# --------------------------------------------------------------------------

def normalized_product(weights, likelihood):
    """particlepdf.py:136-139 — w*l with NaN->0 / inf->DBL_MAX, then /sum,
    again passed through nan_to_num (0/0 -> all zeros)."""
    prod = np.nan_to_num(weights * likelihood)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.nan_to_num(prod / np.sum(prod))


NUMPY_REDUCE_PIECE = 8192     # elements per inner-loop call of a ufunc reduction (np.getbufsize() default)


def numpy_pairwise_sum(a):
    """np.sum / np.add.reduce of a contiguous float64 vector, operation by operation — the ORDER in which NumPy
    adds (numpy/_core/src/umath/loops_utils.h.src: pairwise_sum; reached from particlepdf.py:138 ``np.sum(tmp)`` and
    :243 ``np.sum(wsquared)``): fewer than 8 elements one after the other from 0.0; up to 128 in eight interleaved
    running sums r[j] += a[i + j], combined ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), the last n % 8 elements added one
    by one; longer vectors split at n/2 rounded down to a multiple of 8, the halves summed recursively and added.
    The reduction starts from the identity 0.0 and walks the vector in pieces of the ufunc buffer size (8192
    elements, np.getbufsize()'s default): res = 0.0; res += pairwise(piece) for every piece — found by pinning this
    restatement against np.sum, not from the source.  Pure-Python loops: what the device's
    ``strict_sums`` mode (csrc/obe_update.hip: numpy_order_sum) is held to BIT FOR BIT; pinned against np.sum itself
    for every length 0..5000 in tests/test_oracle_golden.py."""
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)

    def pw(lo, n):
        if n < 8:
            res = 0.0
            for i in range(lo, lo + n):
                res = res + a[i]
            return res
        if n <= 128:
            r = [a[lo + j] for j in range(8)]
            i = 8
            while i < n - (n % 8):
                for j in range(8):
                    r[j] = r[j] + a[lo + i + j]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < n:
                res = res + a[lo + i]
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return pw(lo, n2) + pw(lo + n2, n - n2)

    with np.errstate(all="ignore"):
        res = np.float64(0.0)
        for lo in range(0, a.size, NUMPY_REDUCE_PIECE):
            res = res + pw(lo, min(NUMPY_REDUCE_PIECE, a.size - lo))
        return res


def effective_particles(weights):
    """particlepdf.py:243-244 — N_eff = 1 / sum(nan_to_num(w^2))."""
    with np.errstate(divide="ignore"):
        return 1.0 / np.sum(np.nan_to_num(weights * weights))


def weighted_mean(particles, weights):
    """particlepdf.py:182-183 — np.average(axis=1, weights=w) = sum(x w)/sum(w)."""
    particles = np.asarray(particles, dtype=np.float64)
    return np.sum(particles * weights, axis=1) / np.sum(weights)


def weighted_covariance(particles, weights):
    """particlepdf.py:194-198 — np.cov(X, aweights=w): deviations from the
    weighted mean, normalised by  sum(w) - sum(w^2)/sum(w)  (ddof=1 with
    aweights, numpy lib/_function_base_impl.py:2870-2892).  Always (D, D)."""
    x = np.atleast_2d(np.asarray(particles, dtype=np.float64))
    w_sum = np.sum(weights)
    mu = np.sum(x * weights, axis=1) / w_sum
    dev = x - mu[:, None]
    fact = w_sum - np.sum(weights * weights) / w_sum
    cov = dev @ (dev * weights).T
    cov *= np.true_divide(1, fact)        # numpy scales by the reciprocal
    return cov


def weighted_std(particles, weights):
    """particlepdf.py:209-214 — per dimension sqrt(<x^2> - <x>^2) with the
    averages taken as plain dot products with the weights (no /sum(w))."""
    x = np.atleast_2d(np.asarray(particles, dtype=np.float64))
    out = np.empty(x.shape[0])
    for i, row in enumerate(x):
        m1 = np.dot(row, weights)
        m2 = np.dot(row * row, weights)
        out[i] = m2 - m1 ** 2
    return np.sqrt(out)


def weight_cdf(weights):
    """The CDF ``Generator.choice(p=w)`` searches (particlepdf.py:330-331;
    numpy/random/_generator.pyx choice(): cdf = p.cumsum(); cdf /= cdf[-1])."""
    cdf = np.cumsum(np.asarray(weights, dtype=np.float64))
    cdf /= cdf[-1]
    return cdf


def choice_indices(weights, uniforms):
    """Multinomial draw of particlepdf.py:330-331 from explicit uniforms:
    index = number of CDF entries <= u  (searchsorted side='right'), int64."""
    return weight_cdf(weights).searchsorted(uniforms, side="right").astype(np.int64)


def systematic_indices(weights, u0, n_draws):
    """Systematic resampling (extension; BASELINE.json north_star): the CDF searched at the
    stratified points (i + u0) / n_draws, one uniform u0 in [0, 1)."""
    points = (np.arange(n_draws, dtype=np.float64) + u0) / np.float64(n_draws)
    return weight_cdf(weights).searchsorted(points, side="right").astype(np.int64)


def nudge_factor(cov):
    """Factor F with  multivariate_normal(0, cov, n) == standard_normal((n,d)) @ F.T
    (particlepdf.py:300-301; numpy method='svd': F = u * sqrt(s))."""
    u, s, _ = np.linalg.svd(cov)
    return u * np.sqrt(s)


# --------------------------------------------------------------------------
# OptBayesExpt arithmetic
# --------------------------------------------------------------------------

def flatten_settings(setting_values):
    """obe_base.py:174-176 — all combinations, meshgrid indexing='ij', (S, N_s)."""
    grids = np.meshgrid(*setting_values, indexing="ij")
    return np.array([g.flatten() for g in grids])


def gauss_likelihood(y_model, y_meas, sigma):
    """obe_base.py:269-271 — exp(-((y_model - y)/sigma)^2 / 2) / sigma."""
    return np.exp(-((y_model - y_meas) / sigma) ** 2 / 2) / sigma


def yvar_from_draws(model, allsettings, draws, cons, return_mean=False):
    """obe_base.py:480-488 — evaluate the model over all settings for each drawn
    parameter set (columns of ``draws``), then the *unweighted, ddof=0* variance
    over the draw axis.  Returns (C, N_s)."""
    n_draws = draws.shape[1]
    ys = []
    for i in range(n_draws):
        y = np.atleast_2d(np.asarray(model(allsettings, draws[:, i], cons), dtype=np.float64))
        ys.append(np.broadcast_to(y, (y.shape[0], allsettings.shape[1])))
    ys = np.array(ys)
    if return_mean:
        return np.var(ys, axis=0), np.mean(ys, axis=0)
    return np.var(ys, axis=0)


def yvar_full_sweep(model, allsettings, particles, weights, cons, chunk=2048, return_mean=False):
    """Full-sweep limit of obe_base.py:480-488: every particle is
    a draw and the variance is *weighted*,  sum_p w_p (y_sp - ybar_s)^2 / sum_p w_p,
    two-pass like np.var.  The particle axis is chunked so the (N_p, C, N_s)
    temporary of the reference is never materialised.  Returns (C, N_s)."""
    particles = np.asarray(particles, dtype=np.float64)
    n_p = particles.shape[1]
    w_sum = np.sum(weights)

    def evaluate(lo, hi):
        pars = tuple(p[lo:hi, None] for p in particles)
        sets = tuple(s[None, :] for s in allsettings)
        y = np.asarray(model(sets, pars, cons), dtype=np.float64)
        if y.ndim == 2:
            y = y[None]
        return np.broadcast_to(y, (y.shape[0], hi - lo, allsettings.shape[1]))

    ybar = 0.0
    for lo in range(0, n_p, chunk):
        hi = min(n_p, lo + chunk)
        ybar = ybar + np.einsum("cps,p->cs", evaluate(lo, hi), weights[lo:hi])
    ybar = ybar / w_sum
    acc = 0.0
    for lo in range(0, n_p, chunk):
        hi = min(n_p, lo + chunk)
        dev = evaluate(lo, hi) - ybar[:, None, :]
        acc = acc + np.einsum("cps,p->cs", dev * dev, weights[lo:hi])
    if return_mean:
        return acc / w_sum, ybar
    return acc / w_sum


def utility_from_yvar(yvar, noise_var, cost):
    """obe_base.py:650-655 — sum over channels of var_p/var_n, divided by cost."""
    return np.sum(yvar / noise_var, axis=0) / cost


def mean_noise_variance(parameters, noise_index, weights):
    """obe_noiseparam.py:132-136 — weighted mean of sigma^2 per channel, (C, 1)."""
    sig2 = np.asarray(parameters)[np.atleast_1d(noise_index)] ** 2
    out = np.sum(sig2 * weights, axis=1) / np.sum(weights)
    return out.reshape((-1, 1))


# --------------------------------------------------------------------------
# Stateful mirrors (same method surface as the reference classes) so that whole
# seeded trajectories can be replayed against the golden fixtures.
# --------------------------------------------------------------------------

class OracleParticlePDF:
    """particlepdf.py:12-345."""

    def __init__(self, prior, a_param=0.98, resample_threshold=0.5,
                 auto_resample=True, scale=True, use_jit=True):
        self.tuning_parameters = {"a_param": a_param,
                                  "resample_threshold": resample_threshold,
                                  "auto_resample": auto_resample,
                                  "scale": scale}
        self.particles = np.asarray(prior)
        self.n_particles = self.particles.shape[-1]
        self.n_dims = self.particles.shape[0]
        self.particle_weights = np.ones(self.n_particles) / self.n_particles
        self.just_resampled = False
        self.rng = np.random.default_rng()
        # diagnostics for the parity tests
        self.last_draw_indices = None
        self.last_n_eff = None

    def set_pdf(self, samples, weights=None):
        """particlepdf.py:147-171."""
        self.particles = np.asarray(samples)
        self.n_particles = self.particles.shape[-1]
        self.n_dims = self.particles.shape[0]
        if weights is None:
            self.particle_weights = np.ones(self.n_particles) / self.n_particles
        elif len(weights) != self.n_particles:
            raise ValueError("Length of weights does not match the number of particles.")
        else:
            self.particle_weights = weights / np.sum(weights)

    def mean(self):
        return weighted_mean(self.particles, self.particle_weights)

    def covariance(self):
        return weighted_covariance(self.particles, self.particle_weights)

    def std(self):
        return weighted_std(self.particles, self.particle_weights)

    def bayesian_update(self, likelihood):
        """particlepdf.py:216-234."""
        self.particle_weights = normalized_product(self.particle_weights, likelihood)
        if self.tuning_parameters["auto_resample"]:
            self.resample_test()

    def resample_test(self):
        """particlepdf.py:236-258 — the <10 % branch warns *and* resamples."""
        n_eff = effective_particles(self.particle_weights)
        self.last_n_eff = n_eff
        if n_eff < 0.1 * self.n_particles:
            warnings.warn("\nParticle filter rejected > 90 % of particles. "
                          f"N_eff = {n_eff:.2f}. "
                          "Particle impoverishment may lead to errors.",
                          RuntimeWarning)
            self.resample()
            self.just_resampled = True
        elif n_eff / self.n_particles < self.tuning_parameters["resample_threshold"]:
            self.resample()
            self.just_resampled = True
        else:
            self.just_resampled = False

    def randdraw(self, n_draws=1):
        """particlepdf.py:312-345 — n uniforms from self.rng, CDF search, gather."""
        uniforms = self.rng.random(n_draws)
        idx = choice_indices(self.particle_weights, uniforms)
        self.last_draw_indices = idx
        # row-by-row into a C-ordered buffer, as the reference does (:327-343): the
        # memory order of the result decides NumPy's summation order in the next
        # mean()/covariance(), i.e. the last bit of every later resample
        draws = np.zeros((self.n_dims, n_draws))
        for i, row in enumerate(self.particles):
            draws[i] = row[idx]
        return draws

    def resample(self):
        """particlepdf.py:260-310.  RNG order: N uniforms, then N*D normals
        (row-major (N, D)).  Mean/covariance use the pre-resample weights."""
        n, d = self.n_particles, self.n_dims
        if self.tuning_parameters.get("resample_method", "multinomial") == "systematic":
            # extension (not in the reference): one uniform, draws at (i + u0) / N
            idx = systematic_indices(self.particle_weights, self.rng.random(), n)
            self.last_draw_indices = idx
            coords = np.empty((d, n))
            for i in range(d):
                coords[i] = np.asarray(self.particles[i], dtype=np.float64)[idx]
        else:
            coords = self.randdraw(n)
        cov = self.covariance()
        center = self.mean().reshape((d, 1))
        a = self.tuning_parameters["a_param"]
        z = self.rng.standard_normal((n, d))
        nudged = coords + (z @ nudge_factor((1 - a ** 2) * cov).T).T
        if self.tuning_parameters["scale"]:
            self.particles = nudged * a + center * (1 - a)
        else:
            self.particles = nudged
        self.particle_weights = np.full(n, 1.0 / n)


class OracleOptBayesExpt(OracleParticlePDF):
    """obe_base.py:21-824, variance utility + optimal/good selection only.

    ``utility_method='variance_full'`` is the full-sweep mode of SURVEY.md D1-ii
    (not a reference option)."""

    def __init__(self, measurement_model, setting_values, parameter_samples,
                 constants, n_draws=DEFAULT_N_DRAWS, choke=None, use_jit=True,
                 utility_method="variance_approx", selection_method="optimal",
                 pickiness=15, default_noise_std=1.0, n_channels=None, **kwargs):
        OracleParticlePDF.__init__(self, parameter_samples, use_jit=use_jit, **kwargs)
        self.model_function = measurement_model
        self.setting_values = setting_values
        self.allsettings = flatten_settings(setting_values)
        self.setting_indices = np.arange(self.allsettings.shape[1], dtype=int)
        self.parameters = self.particles
        self.cons = constants
        self.choke = choke
        self.N_DRAWS = n_draws
        self.pickiness = pickiness
        self.last_setting_index = 0
        if n_channels is None:
            # obe_base.py:807-824 (trial evaluation; here without consuming self.rng)
            probe = measurement_model(self.allsettings[:, 0],
                                      np.asarray(self.particles, dtype=np.float64)[:, :1],
                                      constants)
            n_channels = len(np.atleast_1d(probe))
        self.n_channels = n_channels
        self.default_noise_std = np.ones((self.n_channels, 1)) * default_noise_std
        if utility_method not in ("variance_approx", "variance_full", "max_min", "pseudo_utility",
                                  "full_kld_utility"):
            raise SyntaxError(f"Unknown utility method, {utility_method}.")
        self.utility_method = utility_method
        self.noise_rng = np.random.default_rng()     # the reference's module-level obe_base.rng
        if selection_method not in ("optimal", "good"):
            raise SyntaxError(f"Unknown selection_method, {selection_method}.")
        self.get_setting = self.opt_setting if selection_method == "optimal" else self.good_setting

    # model wrappers (obe_base.py:215-222, 298-338): always return (C, N)
    def _model(self, sets, pars):
        y = self.model_function(sets, pars, self.cons)
        if self.n_channels == 1:
            return (y,)
        return y

    def eval_over_all_parameters(self, onesettingset):
        return self._model(onesettingset, self.parameters)

    def eval_over_all_settings(self, oneparamset):
        return self._model(self.allsettings, oneparamset)

    def likelihood(self, y_model, measurement_record):
        """obe_base.py:451-461 — product over channels (zip truncates)."""
        _, y_meas, sigma = measurement_record
        lky = 1.0
        for y_m, y, s in zip(y_model, np.atleast_1d(y_meas), np.atleast_1d(sigma)):
            lky = lky * gauss_likelihood(y_m, y, s)
        if self.choke is not None:
            return np.power(lky, self.choke)
        return lky

    def enforce_parameter_constraints(self):
        pass

    def pdf_update(self, measurement_record, y_model_data=None):
        """obe_base.py:340-399."""
        if y_model_data is None:
            y_model_data = self.eval_over_all_parameters(measurement_record[0])
        self.bayesian_update(self.likelihood(y_model_data, measurement_record))
        self.parameters = self.particles
        if self.just_resampled:
            self.enforce_parameter_constraints()
        return self.particles, self.particle_weights

    def yvar_noise_model(self):
        return self.default_noise_std ** 2

    def cost_estimate(self):
        return 1.0

    def yvar_from_parameter_draws(self):
        """obe_base.py:463-489 (or the full-sweep limit)."""
        # (last_ymean: the predicted mean output per setting rides along, so that a checker can state the
        # conditioning (mean y)^2 / var of each variance from THIS side — tools/fuzz_parity.py)
        if self.utility_method == "variance_full":
            self.last_yvar, self.last_ymean = yvar_full_sweep(self.model_function, self.allsettings, self.particles,
                                                              self.particle_weights, self.cons, return_mean=True)
            return self.last_yvar
        draws = self.randdraw(self.N_DRAWS)
        self.last_yvar, self.last_ymean = yvar_from_draws(self.model_function, self.allsettings, draws, self.cons,
                                                          return_mean=True)
        return self.last_yvar

    def _y_space(self):
        """obe_base.py:480-484 — the model over all settings for N_DRAWS drawn parameter sets."""
        draws = self.randdraw(self.N_DRAWS)
        ys = []
        for i in range(self.N_DRAWS):
            y = np.atleast_2d(np.asarray(self.model_function(self.allsettings, draws[:, i], self.cons),
                                         dtype=np.float64))
            ys.append(np.broadcast_to(y, (y.shape[0], self.allsettings.shape[1])))
        return np.array(ys)

    def utility(self):
        from scipy.stats import differential_entropy as diffent        # as obe_base.py:7-10
        if self.utility_method == "max_min":                           # obe_base.py:520-535, 621-626
            ysp = self._y_space()
            var_p = (np.max(ysp, axis=0) - np.min(ysp, axis=0)) ** 2
        elif self.utility_method == "pseudo_utility":                  # obe_base.py:508-518, 681-686
            var_p = np.exp(2 * diffent(self._y_space(), axis=0)) / (2 * np.pi * np.e)
        elif self.utility_method == "full_kld_utility":                # obe_base.py:707-720
            draws = self.randdraw(self.N_DRAWS)
            nva = self.noise_rng.normal(0, 1.0, self.N_DRAWS * self.n_channels)
            noise = (nva.reshape((self.n_channels, self.N_DRAWS)) * np.sqrt(self.yvar_noise_model())).T
            ysp = np.array([np.atleast_2d(self.model_function(self.allsettings, draws[:, i], self.cons))
                            + noise[i] for i in range(self.N_DRAWS)])
            return np.exp(diffent(ysp, axis=0) - diffent(noise, axis=0)) - 1.0
        else:
            var_p = self.yvar_from_parameter_draws()
        return utility_from_yvar(var_p, self.yvar_noise_model(), self.cost_estimate())

    def opt_setting(self):
        """obe_base.py:733-756 — first maximum wins."""
        self.last_utility = self.utility()
        best = int(np.argmax(self.last_utility))
        self.last_setting_index = best
        return tuple(self.allsettings[:, best])

    def good_setting(self, pickiness=None):
        """obe_base.py:758-789 — utility**pickiness as selection probabilities."""
        if pickiness is None:
            pickiness = self.pickiness
        self.last_utility = self.utility()
        p = np.nan_to_num(self.last_utility ** pickiness)
        p = p / np.sum(p)
        u = self.rng.random()
        good = int(choice_indices(p, np.array([u]))[0])
        self.last_setting_index = good
        return tuple(self.allsettings[:, good])


class OracleOptBayesExptNoiseParameter(OracleOptBayesExpt):
    """obe_noiseparam.py:5-136."""

    def __init__(self, measurement_model, setting_values, parameter_samples,
                 constants, noise_parameter_index=None, **kwargs):
        OracleOptBayesExpt.__init__(self, measurement_model, setting_values,
                                    parameter_samples, constants, **kwargs)
        self.noise_parameter_index = np.atleast_1d(noise_parameter_index)
        if len(self.noise_parameter_index) != self.n_channels:
            raise RuntimeError("noise_parameter_index is not compatible with"
                               f" {self.n_channels} measurement channels")

    def enforce_parameter_constraints(self):
        """obe_noiseparam.py:57-79 — zero the weight of particles with sigma <= 0."""
        changed = False
        for row in np.asarray(self.parameters)[self.noise_parameter_index]:
            bad = np.nonzero(row <= 0)[0]
            if len(bad) > 0:
                changed = True
                self.particle_weights[bad] = 0
        if changed:
            self.particle_weights = self.particle_weights / np.sum(self.particle_weights)

    def likelihood(self, y_model, measurement_record):
        """obe_noiseparam.py:109-120 — sigma is a per-particle parameter row."""
        y_meas = measurement_record[1]
        sigma = np.asarray(self.parameters)[self.noise_parameter_index]
        lky = 1.0
        for y_m, y, s in zip(y_model, np.atleast_1d(y_meas), sigma):
            lky = lky * gauss_likelihood(y_m, y, s)
        if self.choke is not None:
            return np.power(lky, self.choke)
        return lky

    def yvar_noise_model(self):
        return mean_noise_variance(self.parameters, self.noise_parameter_index,
                                   self.particle_weights)


class OracleOptBayesExptSweeper(OracleOptBayesExptNoiseParameter):
    """demos/sweeper/obe_sweeper.py:9-229 — settings are (start, stop) index pairs on the
    first setting axis; a measurement is a whole sweep.  ``sweep_rng`` stands for that
    module's own module-level generator (used by good_setting / random_setting)."""

    def __init__(self, measurement_model, setting_values, parameter_samples, constants,
                 noise_parameter_index, **kwargs):
        OracleOptBayesExptNoiseParameter.__init__(self, measurement_model, setting_values, parameter_samples,
                                                  constants, noise_parameter_index, **kwargs)
        self.sweep_settings = setting_values[0]
        self.start_stop_subsample = 3
        self.start_stop_indices = self._generate_start_stop_indices()
        self.start_stop_choice_indices = np.arange(len(self.start_stop_indices), dtype=int)
        self.start_stop_values = self.sweep_settings[self.start_stop_indices]
        self.cost_of_new_sweep = 5.
        self.sweep_rng = np.random.default_rng()

    def _generate_start_stop_indices(self):
        """obe_sweeper.py:207-229: every (start, stop) with stop > start on the sub-sampled
        index grid 0, k, 2k, ... plus the last index; start-major order."""
        n = len(self.sweep_settings)
        grid = list(range(0, n, self.start_stop_subsample))
        if grid[-1] != n - 1:
            grid.append(n - 1)
        grid = np.array(grid)
        i, j = np.triu_indices(len(grid), 1)
        return np.stack([grid[i], grid[j]], axis=1)

    def pdf_update(self, measurement_record):
        """obe_sweeper.py:86-100: one NoiseParameter update per point of the sweep."""
        (xs,), ys = measurement_record
        for x, y in zip(xs, ys):
            OracleOptBayesExptNoiseParameter.pdf_update(self, ((x,), y))

    def sweep_cost_estimate(self):
        """obe_sweeper.py:106-120."""
        return self.start_stop_indices[:, 1] - self.start_stop_indices[:, 0] + self.cost_of_new_sweep

    def sweep_utility(self):
        """obe_sweeper.py:122-149: running sum of the point utility, differenced at the ends."""
        cost = self.sweep_cost_estimate()
        proto = np.cumsum(self.utility())
        ends = proto[self.start_stop_indices]
        return (ends[:, 1] - ends[:, 0]) / cost

    def opt_setting(self):
        """obe_sweeper.py:151-167 — returns the (start, stop) index pair."""
        self.last_utility = self.sweep_utility()
        index = int(np.argmax(self.last_utility))
        self.last_setting_index = index
        return self.start_stop_indices[index]

    def good_setting(self):
        """obe_sweeper.py:169-193."""
        self.last_utility = self.sweep_utility()
        p = self.last_utility ** self.pickiness
        p = p / np.sum(p)
        index = int(choice_indices(p, np.array([self.sweep_rng.random()]))[0])
        self.last_setting_index = index
        return self.start_stop_indices[index]
