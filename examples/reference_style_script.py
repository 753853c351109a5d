"""A script written the way the reference's demos are (cf. demos/find_peak/sequentialLorentzian.py:
the model is a plain Python function, the package is imported as ``optbayesexpt``), used to check
the drop-in route: run it with  PYTHONPATH=<repo>:<repo>/compat OBE_AUTO_DEVICE_MODEL=1.
Prints the true and the measured parameters; exits non-zero if the peak was not found."""
import sys

import numpy as np

from optbayesexpt import MeasurementSimulator, OptBayesExpt


def my_model_function(sets, pars, cons):
    """Lorentzian peak on a background."""
    x, = sets
    x0, a, b = pars
    d, = cons
    return b + a / (((x - x0) / d) ** 2 + 1)


n_measure = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
gen = np.random.default_rng(12)
xvals = np.linspace(1.5, 4.5, 200)
sets = (xvals,)
pars = (gen.uniform(2, 4, n_samples), gen.uniform(-2000, -400, n_samples), gen.normal(50000, 1000, n_samples))
cons = (0.1,)
my_obe = OptBayesExpt(my_model_function, sets, pars, cons, scale=False)
true_pars = (2.9, -1100.0, 50000.0)
my_sim = MeasurementSimulator(my_model_function, true_pars, cons, noise_level=500.0)
for i in range(n_measure):
    xmeas = my_obe.good_setting(pickiness=19) if i % 2 else my_obe.opt_setting()
    ymeasure = my_sim.simdata(xmeas)
    my_obe.pdf_update((xmeas, ymeasure, 500.0))
mean, sigma = my_obe.mean(), my_obe.std()
on_device = getattr(my_obe, "_device_model", None) is not None
print("model on the device:", on_device)
for name, t, m, s in zip(("x0", "a", "b"), true_pars, mean, sigma):
    print(f"{name:>3s} = {t:10.3f}; measured {m:10.3f} +/- {s:8.3f}")
sys.exit(0 if abs(mean[0] - true_pars[0]) < 5 * sigma[0] + 1e-9 and sigma[0] < 0.02 else 1)
