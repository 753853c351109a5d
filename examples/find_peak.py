"""Sequential Bayesian experiment design for a Lorentzian peak — the measurement loop of the
reference's demos/find_peak/sequentialLorentzian.py:88-150, written against optbayesexpt_amd
(no plotting).  Only the import and the model object differ from the reference script.

    python examples/find_peak.py [n_measure] [n_samples] [optimal|good]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optbayesexpt_amd as optbayesexpt                     # noqa: E402  (same class names as the reference)


def main(n_measure=200, n_samples=50000, selection="optimal", seed=0, quiet=False):
    rng = np.random.default_rng(seed)
    # the model: y = b + a / (((x - x0)/d)**2 + 1); a formula (compiled once, cached) or the
    # hand-tuned registry entry optbayesexpt.models.lorentzian()
    my_model_function = optbayesexpt.models.lorentzian()

    xvals = np.linspace(1.5, 4.5, 200)
    settings = (xvals,)
    x0_samples = rng.uniform(2, 4, n_samples)
    a_samples = rng.uniform(-2000, -400, n_samples)
    b_samples = rng.normal(50000, 1000, n_samples)
    parameters = (x0_samples, a_samples, b_samples)
    constants = (0.1,)

    my_obe = optbayesexpt.OptBayesExpt(my_model_function, settings, parameters, constants, scale=False)
    my_obe.rng = np.random.default_rng(seed + 1)
    # the simulator's noise comes from the module-level generator of obe_utils, as in the reference
    optbayesexpt.obe_utils.rng = np.random.default_rng(seed + 2)

    true_pars = (rng.uniform(2.5, 3.5), rng.uniform(-2000, -400), 50000.0)
    noise_level = 500.0
    my_sim = optbayesexpt.MeasurementSimulator(my_obe.model_function, true_pars, constants, noise_level=noise_level)

    sig = []
    for i in range(n_measure):
        if selection == "optimal":
            xmeas = my_obe.opt_setting()
        else:
            xmeas = my_obe.good_setting(pickiness=19)
        ymeasure = my_sim.simdata(xmeas)
        my_obe.pdf_update((xmeas, ymeasure, noise_level))
        sig.append(my_obe.std()[0])
    mean, std = my_obe.mean(), my_obe.std()
    if not quiet:
        for name, t, m, s in zip(("x0", "a", "b"), true_pars, mean, std):
            print(f"{name:>3s} = {t:10.3f}; measured {m:10.3f} +/- {s:8.3f}")
        print(f"sigma(x0) after 10 / {n_measure} measurements: {sig[9]:.4f} / {sig[-1]:.5f}")
    return true_pars, mean, std


if __name__ == "__main__":
    args = sys.argv[1:]
    main(int(args[0]) if args else 200, int(args[1]) if len(args) > 1 else 50000,
         args[2] if len(args) > 2 else "optimal")
