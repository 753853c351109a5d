"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

CPU (NumPy) restatement of the optbayesexpt hot path, used only as the
*checker* by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  Nothing under ``optbayesexpt_amd/`` may import this
package: the product path is the HIP library and fails loudly without it.

Contents: ``obe_oracle.py`` (NumPy restatement of the classes and their arithmetic),
``models.py`` (the demo model formulas), ``csweep.c`` / ``csweep.py`` (plain C + OpenMP
restatement of the Lorentzian full sweep and update: a second oracle and the all-cores CPU
baseline of bench.py).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference (``/root/reference``, optbayesexpt 1.2.0) in the build container,
runs seeded trajectories and writes the ``tests/golden/*.npz`` fixtures;
``tests/test_oracle_golden.py`` checks this restatement against every one of
them, and against the literal expectations of the reference's own unit tests
(``tests/test_particlepdf.py``, ``tests/test_optbayesexpt.py``,
``tests/test_zinference.py::test_infer``).
"""
from .obe_oracle import *  # noqa: F401,F403
