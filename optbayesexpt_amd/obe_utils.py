"""Demo helpers with the reference's names (optbayesexpt/obe_utils.py:8-113).

These are *callers* of the hot path, not part of it (out of scope, O(1)
work per measurement cycle); they are provided so that the reference's demo scripts
import cleanly after ``import optbayesexpt_amd as optbayesexpt``.
"""
import numpy as np

rng = np.random.default_rng()


class MeasurementSimulator:
    """Simulated noisy measurements: ``model_function(setting, true_params, cons)`` plus
    Gaussian noise of standard deviation ``noise_level``."""

    def __init__(self, model_function, true_params, cons, noise_level):
        self.model_function = model_function
        self.params = true_params
        self.cons = cons
        self.noise_level = noise_level

    def simdata(self, setting, params=None, noise_level=None):
        params = self.params if params is None else params
        noise_level = self.noise_level if noise_level is None else noise_level
        y = np.array(self.model_function(setting, params, self.cons))
        return y + rng.standard_normal(y.shape) * noise_level


def trace_sort(settings, measurements):
    """Bin repeated settings: (unique sorted settings, mean, standard error, count)."""
    settings = np.asarray(settings)
    measurements = np.asarray(measurements)
    order = np.argsort(settings)
    s_sorted, m_sorted = settings[order], measurements[order]
    uniq, start, counts = np.unique(s_sorted, return_index=True, return_counts=True)
    means, errs = [], []
    for b, c in zip(start, counts):
        chunk = m_sorted[b:b + c]
        means.append(np.mean(chunk))
        errs.append(np.std(chunk) / np.sqrt(c))
    return list(uniq), means, errs, list(counts)


try:        # the reference keeps a copy of scipy's estimator for installations without scipy
    from scipy.stats import differential_entropy      # noqa: F401  (obe_utils.py:116-310; obe_base.py:7-10)
except ImportError:      # pragma: no cover
    def differential_entropy(*args, **kwargs):
        raise ImportError("differential_entropy needs scipy (the device utilities pseudo_utility / "
                          "full_kld_utility carry their own estimator)")
