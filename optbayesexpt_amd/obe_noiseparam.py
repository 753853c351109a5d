"""OptBayesExptNoiseParameter — measurement noise as an unknown (particle) parameter.

Mirrors optbayesexpt/obe_noiseparam.py:5-136: the likelihood takes sigma from a
parameter row (per particle), the utility's noise variance is the weighted mean of
sigma^2, and after every resample particles with sigma <= 0 get zero weight.  All
three run in libobe_hip kernels (K2 with ``h_noise_rows``, the K3 moment block, K6).
"""
import numpy as np
import torch

from . import _lib
from .obe_base import OptBayesExpt, _overridden
from .particlepdf import ParticlePDF, _ptr


class OptBayesExptNoiseParameter(OptBayesExpt):
    """``noise_parameter_index``: int or tuple, one parameter row per output channel
    (obe_noiseparam.py:45-55)."""

    def __init__(self, measurement_model, setting_values, parameter_samples,
                 constants, noise_parameter_index=None, **kwargs):
        OptBayesExpt.__init__(self, measurement_model, setting_values,
                              parameter_samples, constants, **kwargs)
        self.noise_parameter_index = np.atleast_1d(noise_parameter_index)
        if len(self.noise_parameter_index) != self.n_channels:
            raise RuntimeError(f"noise_parameter_index is not compatible with"
                               f" {self.n_channels} measurement channels")
        self._noise_rows = np.zeros(max(_lib.OBE_MAX_DIMS, self.n_channels), dtype=np.int32)
        rows = np.asarray(self.noise_parameter_index, dtype=np.int64)
        rows = np.where(rows < 0, rows + self.n_dims, rows)          # NumPy negative indexing
        if np.any(rows < 0) or np.any(rows >= self.n_dims):
            raise IndexError("noise_parameter_index out of range")
        self._noise_rows[:self.n_channels] = rows

    # -- likelihood: sigma is a parameter row (obe_noiseparam.py:81-120) --------------
    def _likelihood_inputs(self, measurement_record):
        y_meas = measurement_record[1]
        n, yy, _ = self._record_channels(y_meas, None)
        return n, yy, None, self._noise_rows

    def _likelihood_overridden(self):
        return _overridden(self, "likelihood", OptBayesExpt, OptBayesExptNoiseParameter)

    def likelihood(self, y_model, measurement_record):
        """Per-particle-sigma Gaussian likelihood; same device kernel as the base class,
        with sigma read from the noise parameter rows."""
        return OptBayesExpt.likelihood(self, y_model, measurement_record)

    # -- constraint: sigma > 0 (obe_noiseparam.py:57-79) ------------------------------
    def enforce_parameter_constraints(self):
        """Zero the weight of every particle whose noise parameter is <= 0 and
        renormalise; called by ``pdf_update`` right after a resample.  The same two launches leave the
        first moments of the constrained cloud behind (the next sweep's shift and noise variance need
        them), and nothing is waited for: ``last_constraint_count`` reads the count when asked."""
        self._await_host_moments()        # (a second call in a row re-arms the words the first one delivers into)
        par = self._parameters.tensor()
        w = self._weights.tensor()
        changed = self.__dict__.get("_changed_pinned")
        if changed is None:
            changed = self._changed_pinned = _lib.pinned_array(1, np.int64)
        fused = self._parameters is self._particles
        masked = self.__dict__.get("_masked_by_gather")
        self._masked_by_gather = None
        if fused:
            done = False
            if masked is not None and masked == (self._particles.version, self._weights.version):
                # the gather of the resample that pdf_update() has just run zeroed these weights already and left
                # the partial sums: only the renormalisation + first moments remain (one launch instead of two)
                try:
                    self._lib.call("obe_mask_renorm_moments", _ptr(par), par.shape[1], self.n_dims, self.n_particles,
                                   _ptr(self._mask_partials), _ptr(w), _ptr(self._moments_dev),
                                   self._hargs.ptr_keep(self._moments_host), _lib.host_ptr(changed), _ptr(self._ws),
                                   self._ws_bytes, self._stream())
                    done = True
                except _lib.ObeHipError as exc:
                    if not exc.refused_before_launch:
                        raise
                    # (refused before any launch: the full form below finds the same particles)
            if not done:
                self._lib.call("obe_mask_nonpositive_moments", _ptr(par), par.shape[1], self.n_dims, self.n_particles,
                               _lib.host_ptr(self._noise_rows), self.n_channels, _ptr(w), _ptr(self._moments_dev),
                               self._hargs.ptr_keep(self._moments_host), _lib.host_ptr(changed), _ptr(self._ws),
                               self._ws_bytes, self._stream())
            self._constraint_pending = True
            # (weights may have changed: a new version either way; the moments describe exactly them)
            self._weights.mark_device_written()
            self._mom_host_key = self._mom_dev_key = (self._particles.version, self._weights.version, False)
            # (the host copy is complete once every word of it — and the count — has arrived: armed by the call)
            self._mom_host_wait = ((self._hargs.ptr_keep(self._moments_host), 2 + 4 * self.n_dims),
                                   (_lib.host_ptr(changed), 1))
        else:       # a stale `parameters` alias (set_pdf between updates): the mask alone, on those rows
            self._lib.call("obe_mask_nonpositive", _ptr(par), par.shape[1], self.n_particles,
                           _lib.host_ptr(self._noise_rows), self.n_channels, _ptr(w), _lib.host_ptr(changed),
                           _ptr(self._ws), self._ws_bytes, self._stream())
            self._constraint_pending = False
            if changed[0]:
                self._weights.mark_device_written()

    def _resample_mask_rows(self):
        """The gather of a resample may apply this class's constraint itself when that constraint is certain to
        follow: the resample is the one resample_test() runs and reports through ``just_resampled``, inside this
        class's pdf_update() (which then calls enforce_parameter_constraints() — a resample() reached any other way,
        on its own or from an overriding hook, must leave uniform weights, like the reference's), every hook on that
        path is the class's own, and tuning_parameters['mask_in_gather'] (default True) does not say otherwise."""
        if not self.__dict__.get("_constraint_follows") or not self.__dict__.get("_in_reported_resample") \
                or not self.tuning_parameters.get("mask_in_gather", True) \
                or _overridden(self, "enforce_parameter_constraints", OptBayesExpt, OptBayesExptNoiseParameter) \
                or _overridden(self, "resample_test", ParticlePDF) or _overridden(self, "resample", ParticlePDF) \
                or _overridden(self, "bayesian_update", ParticlePDF):
            return None           # (a replaced hook might resample without reporting it: no constraint would follow)
        return self._noise_rows, self.n_channels

    def pdf_update(self, measurement_record, y_model_data=None):
        """obe_base.py:340-399 (the noise-parameter class inherits it); see _resample_mask_rows."""
        self._constraint_follows = True
        try:
            return OptBayesExpt.pdf_update(self, measurement_record, y_model_data)
        finally:
            self._constraint_follows = False
            masked = self.__dict__.get("_masked_by_gather")
            if masked is not None:
                # a gather masked the weights and no enforce_parameter_constraints() consumed it (an exception between
                # the two): what the reference's resample() leaves behind is uniform weights (particlepdf.py:308-309)
                self._masked_by_gather = None
                if masked == (self._particles.version, self._weights.version):
                    self.particle_weights = np.ones(self.n_particles) / self.n_particles

    @property
    def last_constraint_count(self):
        """Particles the most recent enforce_parameter_constraints() gave zero weight (waits for the
        kernel that counts them if it has not delivered yet)."""
        changed = self.__dict__.get("_changed_pinned")
        if changed is None:
            return 0
        if self.__dict__.get("_constraint_pending"):
            self._lib.call("obe_host_word_wait", _lib.host_ptr(changed), self._stream())
            self._constraint_pending = False
        return int(changed[0])

    # -- noise model: weighted mean of sigma^2 (obe_noiseparam.py:122-136) ------------
    def yvar_noise_model(self):
        """(C, 1) weighted mean of sigma^2, reduced on the device (K3 block)."""
        t, _ = OptBayesExptNoiseParameter._noise_var_device(self, values=True)
        return t.cpu().numpy().reshape((self.n_channels, 1))

    def _noise_token(self):
        # (the noise variance is a function of the cloud alone: nothing else to compare)
        if _overridden(self, "yvar_noise_model", OptBayesExpt, OptBayesExptNoiseParameter) \
                or self._parameters is not self._particles:
            return None
        return "cloud"

    def _noise_var_device(self, values=False):
        """What a sweep takes as its noise variance: the K3 block itself with the noise rows encoded in the
        leading dimension (include/obe_hip.h: OBE_NOISE_FROM_MOMENTS) — the kernel that forms the utility divides
        m2[row] by sum w itself; ``values``: the C variances as a device vector (yvar_noise_model())."""
        if _overridden(self, "yvar_noise_model", OptBayesExptNoiseParameter):
            return OptBayesExpt._noise_var_device(self)
        if self._parameters is self._particles:
            mom = self._moments_on_device()
            if not values and self._device_model is not None:
                code = self.__dict__.get("_noise_ld_code")
                if code is None:
                    # (5 bits per channel, OBE_MAX_CHANNELS channels: rows below 32 — any other object hands the
                    # sweep the variances as values instead)
                    rows = [int(r) for r in self._noise_rows[:self.n_channels]]
                    if max(rows) < 32 and len(rows) <= _lib.OBE_MAX_CHANNELS:
                        code = self._noise_ld_code = -1 - sum(r << (5 * c) for c, r in enumerate(rows))
                    else:
                        code = self._noise_ld_code = 0
                if code:
                    return mom, code
        else:
            # stale alias after set_pdf(): the reference averages the rows of
            # ``parameters`` (old samples) with the current weights
            par, w = self._parameters.tensor(), self._weights.tensor()
            if par.shape[1] != w.shape[0]:
                raise ValueError("parameters and particle_weights have different lengths")
            mom = torch.zeros_like(self._moments_dev)
            self._lib.call("obe_moments", _ptr(par), par.shape[1], self.n_dims, par.shape[1], _ptr(w), 0,
                           _ptr(mom), None, _ptr(self._ws), self._ws_bytes, self._stream())
        self._lib.call("obe_noise_var_from_moments", _ptr(mom), self.n_dims, _lib.host_ptr(self._noise_rows),
                       self.n_channels, _ptr(self._noise_dev), self._stream())
        self._noise_cache = None
        return self._noise_dev, 0
