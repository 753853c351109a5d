"""``import optbayesexpt`` for scripts written against the reference package.

Put this directory on ``PYTHONPATH`` (ahead of an installed optbayesexpt) and a script such as
the reference's demos runs on the MI355X classes without an edit::

    PYTHONPATH=/path/to/repo:/path/to/repo/compat OBE_AUTO_DEVICE_MODEL=1 python sequentialLorentzian.py

``OBE_AUTO_DEVICE_MODEL=1`` lets ``OptBayesExpt(my_model_function, ...)`` translate a straight-line
Python model function into the HIP kernels (models.from_function); without it the function stays a
host-callable model.  The names are those the reference exports (optbayesexpt/__init__.py:1-6)
without the TCP server and socket, which are outside this package's scope.
"""
from optbayesexpt_amd import (MeasurementSimulator, OptBayesExpt, OptBayesExptNoiseParameter,      # noqa: F401
                              OptBayesExptSweeper, ParticlePDF, models, obe_base, obe_noiseparam, obe_utils,
                              particlepdf, trace_sort)
from optbayesexpt_amd import __version__                                                            # noqa: F401
