"""Device-model registry (host side).

The reference takes an arbitrary Python callable ``model_function(settings,
parameters, constants)`` (obe_base.py:50-72) and leans on NumPy broadcasting; a HIP
kernel cannot call that.  A :class:`DeviceModel` is a *recognisable* model: it carries
the id of a hand-written gfx950 device function (include/obe_hip.h, ``enum
obe_model_id``) and is itself a callable with the reference's signature, so user code
that needs plain numbers from it — ``MeasurementSimulator(my_obe.model_function, ...)``
in every demo — keeps working unchanged.  That NumPy form is never used by the
classes in this package: all library arithmetic runs in the HIP kernels.

Plain Python callables are still accepted by ``OptBayesExpt`` ("host-callable" mode):
then only the *user's own function* runs on the host, and its outputs are uploaded for
the likelihood / weight update / variance / argmax kernels.
"""
import numpy as np

from ._lib import OBE_MAX_CONSTS, OBE_MAX_DIMS, ObeModelStruct

MODEL_LORENTZ = 1
MODEL_LINE_AB = 2
MODEL_LINE_MB = 3
MODEL_FIRST_PARAM = 4
MODEL_RABI = 5
MODEL_COIL = 6
MODEL_PLUGIN = 100


class DeviceModel:
    """A measurement model that exists as a device function in libobe_hip.

    Attributes:
        model_id, aux: identify the device function.
        n_read: parameter rows the model reads (the particle array may have more,
            e.g. a trailing noise parameter).
        n_setdims, n_channels, n_consts: shape of the model.
    """

    def __init__(self, name, model_id, aux, n_read, n_setdims, n_channels, n_consts, numpy_form,
                 plugin_path=None, safe_sweep=None, range_hint=None, safe_sweep_min_spt=1):
        #: the fast sweep form can leave its exact range only when a lane owns at least this many settings
        #: (obe_sweep_settings_per_lane): 2 for models whose ONLY shared arithmetic is the batched reciprocal
        #: of a lane's settings — a sweep of few settings (one per lane) is then IEEE as it is and needs no
        #: kappa check, no repeat, and no host round trip when nobody asks for its result
        self.safe_sweep_min_spt = int(safe_sweep_min_spt)
        #: optional ``range_hint(settings (S, n), particles (D, N), cons, settings_per_lane)`` ->
        #: True / False: a cheap host-side *prediction* of whether the fast sweep form stays inside
        #: its exact range on this grid and cloud.  False makes the first sweep start with the safe
        #: form instead of discovering it by a poisoned fast attempt; the kernel's own range check
        #: stays the guarantee either way.  (Range of the safe forms themselves: generated models and the
        #: coil invert element by element — any finite denominator; the Lorentzians' safe form inverts the 16
        #: denominators of two particles x 8 settings of one peak together while their product fits a double
        #: and element by element otherwise: the reference's numbers for any |x - x0| / d.)
        self.range_hint = range_hint
        #: path of the per-model plugin library (expression models), else None
        self.plugin_path = plugin_path
        #: the model's fast sweep form poisons batches that leave their exact range and has an
        #: always-IEEE twin for the repeat (include/obe_hip.h: OBE_SWEEP_SAFE)
        self.safe_sweep = bool(plugin_path) if safe_sweep is None else bool(safe_sweep)
        self.name = name
        self.model_id = model_id
        self.aux = aux
        self.n_read = n_read
        self.n_setdims = n_setdims
        self.n_channels = n_channels
        self.n_consts = n_consts
        self._numpy_form = numpy_form
        self.__name__ = name

    def __call__(self, sets, pars, cons):
        """Reference calling convention, for user-side simulation and plotting."""
        return self._numpy_form(sets, pars, cons)

    def __repr__(self):
        return f"DeviceModel({self.name!r})"

    def struct(self, n_params, cons):
        """The ``obe_model`` passed across the C ABI."""
        cons = [float(c) for c in cons]
        if len(cons) < self.n_consts:
            raise ValueError(f"{self.name} needs {self.n_consts} constant(s), got {len(cons)}")
        if len(cons) > OBE_MAX_CONSTS:
            raise ValueError(f"at most {OBE_MAX_CONSTS} constants are supported")
        if not (self.n_read <= n_params <= OBE_MAX_DIMS):
            raise ValueError(f"{self.name} reads {self.n_read} parameters; the particle array has "
                             f"{n_params} (device limit {OBE_MAX_DIMS})")
        m = ObeModelStruct()
        m.id, m.aux = self.model_id, self.aux
        m.n_params, m.n_setdims, m.n_channels = n_params, self.n_setdims, self.n_channels
        m.n_consts = len(cons)
        for i, c in enumerate(cons):
            m.consts[i] = c
        return m


def lorentzian(n_peaks=1):
    """``y = b + sum_k a / (((x - x0_k)/d)**2 + 1)``; parameters ``(x0_1..x0_K, a, b[, ...])``,
    constants ``(d,)``.  K=1 is demos/find_peak/sequentialLorentzian.py:53-75."""
    if not 1 <= n_peaks <= 8:
        raise ValueError("n_peaks must be 1..8")

    def form(sets, pars, cons):
        x, = sets
        a, b = pars[n_peaks], pars[n_peaks + 1]
        d, = cons
        y = b
        for k in range(n_peaks):
            y = y + a / (((x - pars[k]) / d) ** 2 + 1)
        return y

    def in_range(settings, particles, cons, settings_per_lane):
        # K >= 3: the combined form inverts the denominators of a lane's settings together: a tree over
        # prod_k q_k of each, q_k = 1 + ((x - x0_k)/d)^2, range-checked at 1e250 in the kernel
        # (csrc/obe_models.h: batch_div_ge1).  K < 3: two particles' denominators of ONE peak for a lane's
        # settings share a reciprocal (q^(2 spt)); an overflowing product poisons the batch.  Either way the
        # sweep is then repeated with the SAFE form; bounding every q by the extremes of grid and cloud
        # predicts it, so that such a grid starts with the SAFE form
        x, d = np.asarray(settings[0], dtype=np.float64), abs(float(cons[0]))
        x_lo, x_hi = float(np.min(x)), float(np.max(x))
        with np.errstate(all="ignore"):
            per_peak = []
            for k in range(n_peaks):
                lo, hi = float(np.min(particles[k])), float(np.max(particles[k]))
                t = max(abs(x_hi - lo), abs(x_lo - lo), abs(x_hi - hi), abs(x_lo - hi)) / d
                per_peak.append(np.log10(1.0 + t * t))
            if n_peaks >= 3:
                return bool(settings_per_lane * sum(per_peak) < 245.0)
            return bool(settings_per_lane < 2 or 2 * settings_per_lane * max(per_peak) < 245.0)
    # the fast sweep forms (3 and more peaks: the peaks of an evaluation combined into one fraction, range-checked;
    # 1 or 2 peaks: two particles per reciprocal, poisoned by overflow) have an always-IEEE twin for the repeat:
    # the pair form with a per-batch branch to element-by-element reciprocals (csrc/obe_models.h, Lorentz<K>)
    return DeviceModel(f"lorentzian[{n_peaks}]", MODEL_LORENTZ, n_peaks, n_peaks + 2, 1, 1, 1, form,
                       safe_sweep=True, range_hint=in_range, safe_sweep_min_spt=2 if n_peaks < 3 else 1)


def line_ab():
    """``y = p0 + p1*x`` (tests/test_optbayesexpt.py:11-14)."""
    def form(sets, pars, cons):
        x, = sets
        return pars[0] + pars[1] * x
    return DeviceModel("line_ab", MODEL_LINE_AB, 0, 2, 1, 1, 0, form)


def line_mb():
    """``y = p0*x + p1`` (demos/line_plus_noise/line_plus_noise.py:36-53)."""
    def form(sets, pars, cons):
        x, = sets
        return pars[0] * x + pars[1]
    return DeviceModel("line_mb", MODEL_LINE_MB, 0, 2, 1, 1, 0, form)


def first_parameter():
    """``y = p0`` (tests/test_zinference.py:21-26)."""
    def form(sets, pars, cons):
        return pars[0]
    return DeviceModel("first_parameter", MODEL_FIRST_PARAM, 0, 1, 1, 1, 0, form)


def rabi():
    """Rabi counts, settings ``(pulsetime, detuning)``, parameters ``(B1, f_center)``,
    constants ``(baseline, contrast, T1)`` (demos/pipulse/pipulse.py:18-49)."""
    def form(sets, pars, cons):
        pulsetime, delta_f = sets
        b1, f_center = pars[0], pars[1]
        baseline, contrast, t1 = cons
        zz = ((delta_f - f_center) / b1) ** 2
        f_rabi = np.hypot(delta_f - f_center, b1)
        return baseline * (1 - np.exp(-pulsetime / t1) * contrast / 2 *
                           (1 - np.cos(np.pi * 2 * f_rabi * pulsetime)) / (zz + 1))
    return DeviceModel("rabi", MODEL_RABI, 0, 2, 2, 1, 3, form)


def coil():
    """Impedance of a lossy coil with stray capacitance, setting ``(omega,)``, parameters
    ``(L, R, C[, ...])``, two channels (Re Z, Im Z) (demos/lockin/lockin_of_coil.py:63-102)."""
    def form(sets, pars, cons):
        w, = sets
        L, R, C = pars[0], pars[1], pars[2]
        z = 1 / (1 / (R + 1j * w * L) + 1j * w * C)
        return np.array((np.real(z), np.imag(z)))
    return DeviceModel("coil", MODEL_COIL, 0, 3, 1, 2, 0, form, safe_sweep=True)


def from_expression(expression, settings, parameters, constants=(), name=None):
    """A device model from a formula, e.g. ::

        from_expression("b + a / (((x - x0) / d)**2 + 1)",
                        settings=("x",), parameters=("x0", "a", "b"), constants=("d",))

    ``expression`` is one string, or a tuple of strings for a multi-channel model.  Allowed:
    ``+ - * / **``, parentheses, numbers, ``pi``, ``e``, the listed names and
    exp/log/log1p/expm1/sqrt/sin/cos/tan/tanh/sinh/cosh/arctan/arctan2/hypot/abs/minimum/maximum.
    The particle array may carry extra trailing rows (e.g. a noise parameter) that the
    formula does not name.  The first use compiles the kernels for this model with hipcc
    (1.6 s on an MI355X host, 4.4 s on 8 cores: profiles/r06_plugin_build_*.txt) into ``optbayesexpt_amd/lib/plugins/``; later uses load the cached
    library.  The returned object is also callable as ``model(sets, pars, cons)`` (NumPy)."""
    from . import _exprmodel, build
    header, numpy_form, digest = _exprmodel.translate(expression, settings, parameters, constants)
    n_channels = 1 if isinstance(expression, str) else len(expression)
    beyond = _beyond_device_limits(len(parameters), len(settings), n_channels, len(constants))
    if beyond:
        # the reference takes any number of settings, parameters and channels (obe_base.py:174-176, 807-824): a model
        # beyond what the device functions are compiled for runs as a host-callable model — the formula's NumPy form
        # is evaluated on the host, everything around it stays on the device
        import warnings
        warnings.warn(f"expression model kept on the host ({beyond})", RuntimeWarning)
        numpy_form.__name__ = name or f"expression[{digest}]"
        return numpy_form
    lib = build.build_plugin(header, digest)
    return DeviceModel(name or f"expression[{digest}]", MODEL_PLUGIN, 0, len(parameters), len(settings),
                       n_channels, len(constants), numpy_form, plugin_path=lib)


def _beyond_device_limits(n_parameters, n_settings, n_channels, n_consts):
    """'' if a generated model of this shape fits the device interface (include/obe_hip.h), else what does not."""
    from . import _lib
    over = []
    if n_parameters > _lib.OBE_MAX_DIMS:
        over.append(f"{n_parameters} parameters > {_lib.OBE_MAX_DIMS}")
    if n_settings > _lib.OBE_MAX_SETDIMS:
        over.append(f"{n_settings} setting dimensions > {_lib.OBE_MAX_SETDIMS}")
    if n_channels > _lib.OBE_MAX_CHANNELS:
        over.append(f"{n_channels} channels > {_lib.OBE_MAX_CHANNELS}")
    if n_consts > _lib.OBE_MAX_CONSTS:
        over.append(f"{n_consts} constants > {_lib.OBE_MAX_CONSTS}")
    return ", ".join(over)


#: When True (or the environment has OBE_AUTO_DEVICE_MODEL=1), ``OptBayesExpt`` tries
#: ``from_function`` on a plain Python ``model_function`` before settling for host-callable mode.
#: Off by default: the first use of a model compiles its kernels with hipcc (seconds; needs hipcc on the box).
AUTO_TRANSLATE = False


def from_function(model_function, name=None):
    """A device model from the *source* of a reference-style ``model_function(sets, pars, cons)``
    (the way every demo of the reference defines its model), if it is straight-line elementwise
    arithmetic: argument unpacking, local assignments, NumPy / math functions, one ``return`` of
    an expression or of a tuple / ``np.array((...))`` of them (channels).  The function is not
    executed for the translation; afterwards the translated formula is checked against it, bit
    for bit, on random inputs.  Raises ``ValueError`` if the function cannot be translated — the
    caller can then still pass it to ``OptBayesExpt`` as a host-callable model.

    The returned object calls the original function when used as ``model(sets, pars, cons)``."""
    from . import _exprmodel, _fnmodel, build
    exprs, settings, parameters, constants = _fnmodel.expressions_from_function(model_function)
    header, numpy_form, digest = _exprmodel.translate(exprs, settings, parameters, constants)
    _fnmodel.check_against_function(model_function, numpy_form, len(settings), len(parameters), len(constants))
    beyond = _beyond_device_limits(len(parameters), len(settings), len(exprs), len(constants))
    if beyond:
        raise ValueError(f"beyond the device limits ({beyond})")      # (the caller keeps the function on the host)
    lib = build.build_plugin(header, digest)
    label = name or f"function[{getattr(model_function, '__name__', 'model')}:{digest}]"
    dm = DeviceModel(label, MODEL_PLUGIN, 0, len(parameters), len(settings), len(exprs), len(constants),
                     model_function, plugin_path=lib)
    dm.expressions = exprs
    return dm
