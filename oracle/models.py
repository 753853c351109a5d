"""TEST INFRASTRUCTURE — NumPy model functions for the oracle.

These follow the reference's ``model_function(settings, parameters,
constants)`` calling convention (obe_base.py:50-72): either the settings or
the parameters are arrays, the other is a tuple of scalars, and NumPy
broadcasting does the rest.  The formulas are the ones the reference's demos
use; each cites the demo it comes from.
"""
import numpy as np


def lorentzian(sets, pars, cons):
    """demos/find_peak/sequentialLorentzian.py:53-75 (3 parameters) and
    demos/sweeper/sweeper.py:42-64 (a 4th, unused, noise parameter)."""
    x, = sets
    x0, a, b = pars[0], pars[1], pars[2]
    d, = cons
    return b + a / (((x - x0) / d) ** 2 + 1)


def multi_lorentzian(n_peaks):
    """SURVEY.md §8(d) config-5 model (builder-defined, D6): parameters
    (x0_1..x0_K, a, b[, sigma]), one constant d."""
    def model(sets, pars, cons):
        x, = sets
        a, b = pars[n_peaks], pars[n_peaks + 1]
        d, = cons
        y = b
        for k in range(n_peaks):
            y = y + a / (((x - pars[k]) / d) ** 2 + 1)
        return y
    model.__name__ = f"multi_lorentzian_{n_peaks}"
    return model


def line_ab(sets, pars, cons):
    """tests/test_optbayesexpt.py:11-14: y = a + b x."""
    x, = sets
    a, b = pars[0], pars[1]
    return a + b * x


def line_mb(sets, pars, cons):
    """demos/line_plus_noise/line_plus_noise.py:36-53: y = m x + b (3rd
    parameter is the noise sigma, unused by the model)."""
    x, = sets
    m, b = pars[0], pars[1]
    return m * x + b


def first_parameter(sets, pars, cons):
    """tests/test_zinference.py:21-26: output = parameter 0."""
    return pars[0]


def rabi(sets, pars, cons):
    """demos/pipulse/pipulse.py:18-49 (2 settings, 2 parameters, 3 constants)."""
    pulsetime, delta_f = sets
    b1, f_center = pars[0], pars[1]
    baseline, contrast, t1 = cons
    zz = ((delta_f - f_center) / b1) ** 2
    f_rabi = np.hypot(delta_f - f_center, b1)
    return baseline * (1 - np.exp(-pulsetime / t1) * contrast / 2 *
                       (1 - np.cos(np.pi * 2 * f_rabi * pulsetime)) / (zz + 1))


def coil(sets, pars, cons):
    """demos/lockin/lockin_of_coil.py:63-102 (2 output channels: Re Z, Im Z)."""
    w, = sets
    L, R, C = pars[0], pars[1], pars[2]
    y1 = 1 / (R + 1j * w * L)
    y2 = 1j * w * C
    z = 1 / (y1 + y2)
    return np.array((np.real(z), np.imag(z)))
