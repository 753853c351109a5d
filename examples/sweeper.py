"""A swept measurement — the loop of the reference's demos/sweeper/sweeper.py:66-150 against
optbayesexpt_amd.OptBayesExptSweeper (no plotting): settings are (start, stop) index pairs,
each measurement is a whole sweep, the noise level is one of the unknown parameters.

    python examples/sweeper.py [n_measure] [n_samples] [optimal|good]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import optbayesexpt_amd as optbayesexpt                     # noqa: E402
from optbayesexpt_amd import sweeper as sweeper_module      # noqa: E402  (its module-level rng, like obe_sweeper.rng)


def main(n_measure=2000, n_samples=50000, selection="good", seed=0, quiet=False):
    rng = np.random.default_rng(seed)
    sweeper_module.rng = np.random.default_rng(seed + 2)
    model = optbayesexpt.models.lorentzian()                # parameters (x0, a, b, [sigma]); constant d
    xvals = np.linspace(1.5, 4.5, 100)
    parameters = (rng.uniform(2, 4, n_samples), rng.uniform(400, 2000, n_samples),
                  rng.normal(500, 1000, n_samples), rng.exponential(500, n_samples))
    my_obe = optbayesexpt.OptBayesExptSweeper(model, (xvals,), parameters, (0.1,), scale=False,
                                              utility_method="variance_approx", selection_method=selection,
                                              pickiness=20, noise_parameter_index=3)
    my_obe.rng = np.random.default_rng(seed + 1)
    # the simulator's noise comes from the module-level generator of obe_utils, as in the reference
    optbayesexpt.obe_utils.rng = np.random.default_rng(seed + 2)
    noise_level = 2000.0
    true_pars = [rng.uniform(2.5, 3.5), rng.uniform(400, 2000), 500.0, noise_level]
    my_sim = optbayesexpt.MeasurementSimulator(model, true_pars, (0.1,), noise_level=noise_level)

    iterations, sweeps = 0, 0
    while iterations < n_measure:
        start, stop = my_obe.get_setting()
        sweep_x_values = xvals[start:stop]
        ymeasure = my_sim.simdata((sweep_x_values,))
        my_obe.pdf_update(((sweep_x_values,), ymeasure))
        iterations += len(sweep_x_values)
        sweeps += 1
    mean, std = my_obe.mean(), my_obe.std()
    if not quiet:
        print(f"{sweeps} sweeps, {iterations} points")
        for name, t, m, s in zip(("x0", "a", "b", "sigma"), true_pars, mean, std):
            print(f"{name:>5s} = {t:9.3f}; measured {m:9.3f} +/- {s:7.3f}")
    return true_pars, mean, std


if __name__ == "__main__":
    args = sys.argv[1:]
    main(int(args[0]) if args else 2000, int(args[1]) if len(args) > 1 else 50000,
         args[2] if len(args) > 2 else "good")
