"""optbayesexpt_amd — the MI355X-native hot path of NIST's optbayesexpt.

Same six public names as the reference package (optbayesexpt/__init__.py:1-6) minus the
TCP server/socket (out of scope), plus the device-model registry and the sweeper class of
the reference's demos (demos/sweeper/obe_sweeper.py):

    import optbayesexpt_amd as obe
    my_obe = obe.OptBayesExpt(obe.models.lorentzian(), settings, parameters, constants)

Importing the classes loads libobe_hip.so; there is no CPU fallback.
"""
from . import models                                        # noqa: F401
from .particlepdf import ParticlePDF                        # noqa: F401
from .obe_base import OptBayesExpt                          # noqa: F401
from .obe_noiseparam import OptBayesExptNoiseParameter      # noqa: F401
from .sweeper import OptBayesExptSweeper                    # noqa: F401
from .obe_utils import MeasurementSimulator, trace_sort     # noqa: F401
from .dist import SettingsShard                             # noqa: F401

__version__ = "0.1.0"
