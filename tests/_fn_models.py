"""Reference-style model functions (plain Python, the way the reference's demos define them)
used to test models.from_function: restatements of the demo formulas, plus functions the
translator must refuse."""
import numpy as np


def lorentzian(sets, pars, cons):
    """y = b + a / (((x - x0)/d)^2 + 1)   (demos/find_peak/sequentialLorentzian.py:53-75)"""
    # unpack the settings, the parameters (a trailing noise parameter is allowed) and the constants
    x, = sets
    x0, a, b = pars[0], pars[1], pars[2]
    d, = cons
    # the Lorentzian
    return b + a / (((x - x0) / d) ** 2 + 1)


def rabi(sets, pars, cons):
    """Rabi counts (demos/pipulse/pipulse.py:18-49)."""
    pulsetime, delta_f = sets
    b1, f_center = pars
    baseline, contrast, t1 = cons
    zz = ((delta_f - f_center) / b1) ** 2
    f_rabi = np.hypot(delta_f - f_center, b1)
    return baseline * (1 - np.exp(-pulsetime / t1) * contrast / 2
                       * (1 - np.cos(np.pi * 2 * f_rabi * pulsetime)) / (zz + 1))


def peak(x, x0, a, b, d):
    return b + a / (((x - x0) * 2 / d) ** 2 + 1)


def lorentzian_via_helper(sets, pars, cons):
    """The demos often factor the formula out into a helper of the same module."""
    x, = sets
    x0, a, b = pars
    d, = cons
    return peak(x, x0, a, b, d)


def two_channels(sets, pars, cons):
    t = sets[0]
    w, ph = pars[0], pars[1]
    arg = w * t + ph
    return np.array((np.cos(arg), np.sin(arg) + cons[0]))


def with_branch(sets, pars, cons):
    x, = sets
    if x > 1:
        return x
    return pars[0]


def with_complex(sets, pars, cons):
    z = 1 / (pars[0] + 1j * sets[0])
    return np.real(z)


SCALE = 3.0


def with_global(sets, pars, cons):
    return SCALE * sets[0] + pars[0]


def with_loop(sets, pars, cons):
    y = pars[0]
    for k in range(2):
        y = y + sets[0]
    return y
