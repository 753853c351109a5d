"""OptBayesExpt — sequential Bayesian experiment design on one (or a shard of) MI355X.

Constructor, attributes, methods and override points follow the reference class
(optbayesexpt/obe_base.py:21-824).  The hot path named by BASELINE.json — the utility
sweep behind ``opt_setting()`` and the posterior update in ``pdf_update()`` — runs in
the HIP kernels of libobe_hip.so:

=====================================  ==========================================
reference                              here
=====================================  ==========================================
``yvar_from_parameter_draws`` (:463)   ``obe_sweep_utility`` (K1), draws or full
``utility_variance`` (:628)            fused into the K1 finalize pass
``opt_setting`` argmax (:748)          device first-max (K5) [+ RCCL all-gather]
``eval_over_all_parameters`` +         ``obe_bayes_update_model`` (K2, fused)
``likelihood`` + ``bayesian_update``
=====================================  ==========================================

Two kinds of measurement model are accepted:

* a :class:`~optbayesexpt_amd.models.DeviceModel` — everything above runs on the GPU;
* any Python callable with the reference signature ("host-callable" mode) — the
  user's function is evaluated on the host exactly where the reference evaluates it,
  and its outputs are uploaded for the likelihood / update / variance / argmax kernels.

Override points of the reference (``enforce_parameter_constraints``, ``cost_estimate``,
``yvar_noise_model``, ``likelihood``, ``utility_*``) remain ordinary methods that
subclasses may replace with NumPy code; the fused kernels are used only while the
corresponding methods are not overridden.
"""
import os
import warnings

import numpy as np
import torch

from . import _devrng, _lib
from ._sweepstate import Form, Pending, SweepState, Ticket
from . import models as _models
from ._mirror import Mirror, TrackedArray
from .models import DeviceModel
from .particlepdf import ParticlePDF, _P, _ptr

DEFAULT_N_DRAWS = 30            # obe_base.py:19
rng = np.random.default_rng()   # module-level generator of the reference (random_setting)

UTILITY_METHODS = ["variance_approx", "pseudo_utility", "full_kld_utility", "max_min",
                   "variance_full"]
SELECTION_METHODS = ["optimal", "good", "random"]


_OVERRIDDEN = {}


# default of tuning_parameters['speculative_sweep'] (A/B measurements: OBE_SPECULATIVE_SWEEP=0 / 1 / auto)
_SPECULATIVE_DEFAULT = {"0": False, "1": True}.get(os.environ.get("OBE_SPECULATIVE_SWEEP", "auto"), "auto")


# default of tuning_parameters['randoms_ahead'] (A/B measurements: OBE_RANDOMS_AHEAD=0 / 1 / auto).  OFF: measured on
# one box (profiles/r06_randoms_ahead.txt), one rank's c5 cycle 1.270-1.288 ms without, 1.302-1.309 ms with 'auto' —
# a cycle that does not resample pays ~50 us (the chain runs into the sweep, which has no idle capacity to give), a
# cycle that does gains ~5 us (with the random chain out of the way the host's SVD of the covariance bounds the gather).
_RANDOMS_AHEAD_DEFAULT = {"0": False, "1": True, "auto": "auto"}.get(os.environ.get("OBE_RANDOMS_AHEAD", "0"), False)

_SCALARS = (float, int, np.floating, np.integer)


def _overridden(obj, name, *owners):
    """True if ``obj``'s class replaces method ``name`` defined by one of ``owners`` (answered once
    per class: the hooks are looked up several times in every cycle)."""
    if name in obj.__dict__:            # replaced on the instance (obe.cost_estimate = f)
        return True
    key = (type(obj), name, owners)
    impl = getattr(type(obj), name)
    hit = _OVERRIDDEN.get(key)
    if hit is None or hit[0] is not impl:       # (a method patched onto the class later is seen too)
        hit = _OVERRIDDEN[key] = (impl, all(impl is not getattr(o, name) for o in owners))
    return hit[1]


class _LazyState:
    """What ``pdf_update`` returns: behaves like the reference's ``(particles,
    particle_weights)`` tuple (unpacking, indexing, len), but the two host arrays are
    only copied off the device if the caller actually looks at them."""

    def __init__(self, owner):
        self._owner = owner

    def __len__(self):
        return 2

    def __getitem__(self, i):
        return (self._owner.particles, self._owner.particle_weights)[i]

    def __iter__(self):
        yield self._owner.particles
        yield self._owner.particle_weights


class _SweepIO:
    """What SweepState (_sweepstate.py) waits through: the library's watched host words and the device."""

    def __init__(self, owner):
        self._owner = owner

    def wait_words(self, words, n, stream):
        self._owner._lib.call("obe_host_words_wait", words, n, stream)

    @staticmethod
    def still_armed(block):
        words = _lib.audit.raw(block).view(np.int64)          # (a poll of armed words, on purpose)
        armed = _lib.HOST_SENTINEL                                        # (positive as a signed word too)
        return bool(words[0] == armed or words[1] == armed or words[2] == armed)

    def synchronize(self):
        torch.cuda.synchronize(self._owner._device)
        _lib.audit.synchronized()


class OptBayesExpt(ParticlePDF):
    """Sequential Bayesian experiment design (see module docstring).

    Extensions beyond the reference signature (obe_base.py:154-159):

    ``utility_method='variance_full'``
        the full settings x particles sweep: every particle is a draw and the
        predicted variance is weighted by the particle weights (the N_DRAWS -> all
        limit of ``yvar_from_parameter_draws``).
    ``settings_shard``
        a :class:`~optbayesexpt_amd.dist.SettingsShard`; this process then sweeps only
        its contiguous slice of the settings and ``opt_setting`` combines the per-rank
        maxima with one all-gather.  (With a host-callable model the user's function is evaluated on every rank's
        host over the whole grid — identically —, so nothing is sliced or gathered; the replicas still share one
        posterior and one generator.)  A sharded object is ONE experiment in G processes, and several of its
        operations are COLLECTIVE — every rank must perform them, in the same order: construction, every
        sweep (``opt_setting`` / ``good_setting`` / ``utility`` ...), ``random_setting``, ``check_replicas()``
        and **assigning** ``rng`` (rank 0's generator state is broadcast inside the setter and adopted by
        every rank, whatever type or seed the other ranks passed).  A script that reseeds on one rank only, or
        in a different order on different ranks — legal against the reference's API, where ``rng`` is a plain
        attribute (particlepdf.py:142-145) — blocks inside the setter until the communicator's timeout expires
        (``torch.distributed.init_process_group(timeout=...)``) instead of raising at once; a generator that
        was replaced behind the setter's back (``obj._rng = ...``) is reported by the next
        ``check_replicas()`` on every rank.
    ``tuning_parameters['speculative_sweep']`` (``'auto'``, ``True``, ``False``; ``variance_full`` only)
        ``pdf_update()`` enqueues the sweep of the next ``opt_setting()`` behind its update (see
        ``_speculation_wanted``): the same results, one host round trip per cycle less.
    ``tuning_parameters['sweep_shift']`` (``'auto'``, ``'always'``, ``'never'``), ``['fused_moments']``,
    ``['replica_check_every']``
        see ``_sweep_device``, ``pdf_update``, ``check_replicas``.
    """

    #: hysteresis of the unshifted sweep (see _sweep_device).  Measured (tools/kappa_error.py, 262 144
    #: and 1 048 576 particles): the unshifted variance is off by ~1e-15 * kappa relative, so 3000
    #: keeps it 30x inside the 1e-10 parity tolerance
    KAPPA_ENTER, KAPPA_LEAVE = 1000.0, 3000.0
    #: after this many consecutive sweeps that had to be repeated with the model's safe twin, the fast
    #: attempt is skipped (the range violation is a property of the settings grid)
    SAFE_STREAK = 3
    #: ... and tried again once every this many sweeps: a posterior that has narrowed, or a cloud
    #: that was replaced, may be back inside the fast form's range (costs one discarded fast sweep
    #: per SAFE_RETRY sweeps while it is not)
    SAFE_RETRY = 64

    def __init__(self, measurement_model, setting_values, parameter_samples,
                 constants, n_draws=DEFAULT_N_DRAWS, choke=None,
                 use_jit=True, utility_method="variance_approx",
                 selection_method="optimal", pickiness=15,
                 default_noise_std=1.0, settings_shard=None, **kwargs):
        ParticlePDF.__init__(self, parameter_samples, use_jit=use_jit, **kwargs)

        if not isinstance(measurement_model, DeviceModel) and callable(measurement_model) and \
                (_models.AUTO_TRANSLATE or os.environ.get("OBE_AUTO_DEVICE_MODEL") == "1"):
            # opt-in: a plain reference-style function is translated from its source when it is
            # straight-line arithmetic; otherwise it stays a host-callable model
            try:
                measurement_model = _models.from_function(measurement_model)
            except (ValueError, RuntimeError, OSError) as exc:
                warnings.warn(f"model function kept on the host ({exc})", RuntimeWarning)
        self.model_function = measurement_model
        self._device_model = measurement_model if isinstance(measurement_model, DeviceModel) else None
        if self._device_model is not None and self.n_dims > _lib.OBE_MAX_DIMS:
            # more parameter rows than a device model may be given (include/obe_hip.h: OBE_MAX_DIMS): the model's
            # NumPy form is evaluated on the host, as any plain callable is; everything else stays on the device
            warnings.warn(f"{self._device_model.name}: {self.n_dims} parameters exceed the device models' limit of "
                          f"{_lib.OBE_MAX_DIMS}; the model function is evaluated on the host", RuntimeWarning)
            self._device_model = None
        self._mlib = self._lib          # model-dependent entry points; an expression model brings its own
        self.setting_values = setting_values
        #: (S, N_s) all setting combinations, meshgrid indexing='ij' (obe_base.py:174-176)
        self.allsettings = np.array([s.flatten() for s in
                                     np.meshgrid(*setting_values, indexing="ij")])
        self.setting_indices = np.arange(len(self.allsettings[0]), dtype=int)
        self._n_settings = self.allsettings.shape[1]
        self._parameters = self._particles      # alias, refreshed by pdf_update (obe_base.py:185,395)
        self.cons = constants
        self.choke = choke
        self.N_DRAWS = n_draws
        self.pickiness = pickiness
        self.measurement_results = []
        self.last_setting_index = 0

        if self._device_model is not None:
            dm = self._device_model
            if self.allsettings.shape[0] != dm.n_setdims:
                raise ValueError(f"{dm.name} takes {dm.n_setdims} setting(s), got {self.allsettings.shape[0]}")
            self._model_struct = dm.struct(self.n_dims, constants)
            # expression models are served by their own plugin library (same entry points)
            self._mlib = _lib.DeviceBound(_lib.load_plugin(dm.plugin_path), self._device) if dm.plugin_path else self._lib
            self._mlib.call("obe_model_validate", self._model_struct)
            self.n_channels = dm.n_channels
        else:
            self._model_struct = None
            self.n_channels = self._model_output_len()
        #: a host-callable model with more output channels than one kernel launch takes: the likelihood is formed
        #: in groups of channels (obe_likelihood_y), the update takes it as an array
        self._wide_channels = self.n_channels > _lib.OBE_MAX_CHANNELS

        if self.n_channels == 1:
            def wrapped_function(s, p, c):
                return (measurement_model(s, p, c),)
            self._model_function = wrapped_function
        else:
            self._model_function = self.model_function

        # settings on the device; a shard sweeps only [s_begin, s_end)
        self._shard = settings_shard
        self._sharded_sweeps = 0
        self._sync_rng()          # replicas of a sharded object draw from rank 0's (unseeded) generator state
        # the record of a measurement as the library takes it (filled in place, addresses made once)
        self._hargs = _lib.HostArgs()
        self._rec_x = self._hargs.keep(np.zeros(_lib.OBE_MAX_SETDIMS))
        self._rec_y = self._hargs.keep(np.zeros(max(_lib.OBE_MAX_CHANNELS, self.n_channels)))
        self._rec_s = self._hargs.keep(np.ones(max(_lib.OBE_MAX_CHANNELS, self.n_channels)))
        self._hargs.keep(self._host_out)
        # which form / shift the next sweep uses, and the sweep pdf_update() enqueues ahead (_sweepstate.py)
        self._sweeps = SweepState(_SweepIO(self), self)
        if settings_shard is not None and self._device_model is not None:
            self._s_begin, self._s_end = settings_shard.bounds(self._n_settings)
        else:
            # (unsharded — or a host-callable model on a sharded object: the user's function runs on the host of every
            # rank over the WHOLE grid, identically, so there is nothing to slice and nothing to gather; the replicas
            # still share one generator and one posterior, and random_setting() still takes rank 0's draw)
            self._s_begin, self._s_end = 0, self._n_settings
        self._settings_dev = torch.from_numpy(np.ascontiguousarray(self.allsettings, dtype=np.float64)) \
            .to(self._device)
        n_local = self._s_end - self._s_begin
        self._yvar_dev = torch.zeros((self.n_channels, max(n_local, 1)), dtype=torch.float64, device=self._device)
        self._utility_dev = torch.zeros(max(n_local, 1), dtype=torch.float64, device=self._device)
        self._noise_dev = torch.zeros(self.n_channels, dtype=torch.float64, device=self._device)
        self._noise_cache = None
        self._noise_src = None        # bytes of the default_noise_std the cached device value was made from
        self._alloc_scratch()

        self.utility_y_space = np.array([])
        self.set_n_draws(n_draws)
        self.default_noise_std = np.ones((self.n_channels, 1)) * default_noise_std

        if utility_method == "variance_approx":
            _utility = self.utility_variance
        elif utility_method == "variance_full":
            if self._device_model is None:
                raise ValueError("utility_method='variance_full' needs a DeviceModel")
            _utility = self.utility_variance
        elif utility_method == "pseudo_utility":
            _utility = self.utility_pseudo
        elif utility_method == "max_min":
            _utility = self.utility_max_min
        elif utility_method == "full_kld_utility":
            _utility = self.utility_full_kld
        else:
            raise SyntaxError(f"Unknown utility method, {utility_method}. "
                              f"Valid utility methods are: {UTILITY_METHODS}")
        self.utility_method = utility_method
        self.utility = _utility

        if selection_method == "optimal":
            _get_setting = self.opt_setting
        elif selection_method == "good":
            _get_setting = self.good_setting
        elif selection_method == "random":
            _get_setting = self.random_setting
        else:
            raise SyntaxError(f"Unknown selection_method, {selection_method}. "
                              f"Valid selection methods are: {SELECTION_METHODS}")
        self.get_setting = _get_setting

    # ------------------------------------------------------------------ state
    def _scratch_dims(self):
        return getattr(self, "_n_settings", 1), getattr(self, "n_channels", 1)

    @property
    def parameters(self):
        """The parameter samples the model is evaluated on: an alias of ``particles``
        that is refreshed by ``pdf_update`` (obe_base.py:185, 395)."""
        return self._parameters.host()

    @parameters.setter
    def parameters(self, value):
        if isinstance(value, TrackedArray) and value._obe_owner is self._particles \
                and value.shape == self._particles.shape:
            self._parameters = self._particles           # ``self.parameters = self.particles``
        else:
            self._parameters = Mirror(self._device, host=np.asarray(value))

    # -------------------------------------------------- replicas of a sharded object
    def _rng_assigned(self):
        self._sync_rng()

    def _sync_rng(self):
        """A sharded object is one experiment in G processes: every rank must draw the same particles
        and the same nudges, or the replicated clouds drift apart and the arg-max combine mixes
        utilities of different posteriors.  The reference's generator is unseeded
        (particlepdf.py:142-145), so rank 0's generator state is adopted by every rank — here, when the
        object is built, and again whenever ``rng`` is assigned (both are collective on a sharded
        object: every rank runs the same script).  Ranks that were seeded alike are left as they are."""
        shard = self.__dict__.get("_shard")
        if shard is None or not shard.connected():
            return
        bg = getattr(self._rng, "bit_generator", None)
        state = shard.broadcast_object_from_rank0(None if bg is None else bg.state, self._device)
        if state is None or shard.rank == 0:
            return
        if bg is None or bg.state.get("bit_generator") != state.get("bit_generator"):
            self._rng = np.random.Generator(getattr(np.random, state["bit_generator"])())
        self._rng.bit_generator.state = state

    def _replica_digest(self):
        """(generator digest, bits of sum w, bits of sum w^2): equal on all ranks while the replicas agree."""
        bg = getattr(self._rng, "bit_generator", None)
        st = 0
        if bg is not None:
            inner = bg.state.get("state", {})
            for v in (inner.values() if isinstance(inner, dict) else ()):
                for word in np.atleast_1d(np.asarray(v, dtype=object)).reshape(-1):
                    word = int(word)
                    while word:
                        st = (st * 1000003) ^ (word & 0xFFFFFFFF)
                        st &= (1 << 62) - 1
                        word >>= 32
        m = self._moments(False)
        sums = np.array([m[0], m[1]], dtype=np.float64).view(np.int64)
        return np.array([st, sums[0], sums[1]], dtype=np.int64)

    def check_replicas(self):
        """All-gather the digest of this rank's replica and raise — on every rank, they all see the same
        table — if the ranks of a sharded object no longer hold the same experiment.  Called every
        ``tuning_parameters['replica_check_every']`` sharded sweeps (default 64; 0 = never)."""
        shard = self._shard
        if shard is None or not shard.connected():
            return True
        table = shard.all_gather_int64(self._replica_digest(), self._device)
        bad = [r for r in range(shard.world_size) if not np.array_equal(table[r], table[0])]
        if bad:
            what = [name for k, name in enumerate(("generator state", "sum of weights", "sum of squared weights"))
                    if np.any(table[:, k] != table[0, k])]
            raise RuntimeError(f"sharded OptBayesExpt: the replicas on ranks {bad} differ from rank 0 in "
                               f"{', '.join(what)} — every rank must apply the same measurements and draw "
                               "from the same generator (assign `rng` on all ranks, never on one)")
        return True

    def set_n_draws(self, n_draws=None):
        """obe_base.py:274-296."""
        if n_draws == "default":
            self.N_DRAWS = DEFAULT_N_DRAWS
        elif n_draws:
            self.N_DRAWS = n_draws
        if self._device_model is None:     # device models keep the y-space on the GPU
            self.utility_y_space = np.zeros((self.N_DRAWS, self.n_channels, self._n_settings))
        return self.N_DRAWS

    # -------------------------------------------------------- model evaluation
    def eval_over_all_parameters(self, onesettingset):
        """Model for one setting and all parameter samples -> C arrays of N_p
        (obe_base.py:298-320)."""
        if self._device_model is None:
            return self._model_function(onesettingset, self.parameters, self.cons)
        y = self._eval_over_all_parameters_device(onesettingset)
        return y.cpu().numpy()

    def _eval_over_all_parameters_device(self, onesettingset):
        par = self._parameters.tensor()
        y = torch.empty((self.n_channels, par.shape[1]), dtype=torch.float64, device=self._device)
        st = self._setting_array(onesettingset)
        self._mlib.call("obe_eval_over_particles", self._model_struct, _ptr(par), par.shape[1], par.shape[1],
                       _lib.host_ptr(st), _ptr(y), par.shape[1], self._stream())
        return y

    def eval_over_all_settings(self, oneparamset):
        """Model for all settings and one parameter set -> C arrays of N_s
        (obe_base.py:322-338)."""
        if self._device_model is None:
            return self._model_function(self.allsettings, oneparamset, self.cons)
        th = _lib.f64(np.asarray(oneparamset, dtype=np.float64).reshape(-1)[:self.n_dims], None)
        if th.size < self._device_model.n_read:
            raise ValueError("parameter set is shorter than the model's parameter list")
        thp = np.zeros(max(_lib.OBE_MAX_DIMS, th.size))
        thp[:th.size] = th
        y = torch.empty((self.n_channels, self._n_settings), dtype=torch.float64, device=self._device)
        self._mlib.call("obe_eval_over_settings", self._model_struct, _ptr(self._settings_dev),
                       self._n_settings, self._n_settings, _lib.host_ptr(thp), _ptr(y), self._n_settings,
                       self._stream())
        return y.cpu().numpy()

    def _setting_array(self, onesettingset):
        """The setting of a record, zero-padded to OBE_MAX_SETDIMS — in this object's record buffer:
        valid until the next call (every user passes it on or copies it at once)."""
        st = self._rec_x
        if type(onesettingset) is tuple and len(onesettingset) == 1 and isinstance(onesettingset[0], _SCALARS):
            st[0] = onesettingset[0]          # (the one-setting tuple opt_setting() returns: no array round trip)
            st[1:] = 0.0
            return st
        vals = np.asarray(onesettingset, dtype=np.float64).reshape(-1)
        k = min(vals.size, _lib.OBE_MAX_SETDIMS)
        st[:k] = vals[:k]
        st[k:] = 0.0
        return st

    # -------------------------------------------------------------- pdf_update
    def _record_channels(self, y_meas, sigma):
        """zip(y_model, atleast_1d(y_meas), atleast_1d(sigma)) truncation
        (obe_base.py:453-455)."""
        if isinstance(y_meas, _SCALARS) and (sigma is None or isinstance(sigma, _SCALARS)):
            yy, s = self._rec_y, None          # one channel given as plain numbers: no array round trip
            yy[0] = y_meas
            yy[1:] = 0.0
            if sigma is not None:
                s = self._rec_s
                s[0] = sigma
                s[1:] = 1.0
            return 1, yy, s
        y = np.asarray(y_meas, dtype=np.float64).reshape(-1)
        n = min(self.n_channels, y.size)
        s = None
        if sigma is not None:
            sg = np.asarray(sigma, dtype=np.float64).reshape(-1)
            n = min(n, sg.size)
            s = self._rec_s                      # (record buffers: valid until the next call)
            s[:n] = sg[:n]
            s[n:] = 1.0
        yy = self._rec_y
        yy[:n] = y[:n]
        yy[n:] = 0.0
        return n, yy, s

    def _likelihood_inputs(self, measurement_record):
        """(n_lik_channels, y_meas[4], sigma[4] or None, noise_rows[4] or None).  The arrays are this object's
        record buffers (filled in place, addresses made once): valid until the next call — copy them to keep."""
        _, y_meas, sigma = measurement_record
        n, yy, s = self._record_channels(y_meas, sigma)
        return n, yy, s, None

    def _choke_value(self):
        return float("nan") if self.choke is None else float(self.choke)

    def pdf_update(self, measurement_record, y_model_data=None):
        """Bayesian update of the parameter distribution from one measurement
        (obe_base.py:340-399).  Returns ``(particles, particle_weights)``."""
        onesetting = measurement_record[0]
        fused = (self._device_model is not None and y_model_data is None
                 and not _overridden(self, "eval_over_all_parameters", OptBayesExpt)
                 and not self._likelihood_overridden())
        if fused:
            n, yy, s, rows = self._likelihood_inputs(measurement_record)
            par = self._parameters.tensor()
            w = self._weights.tensor()
            if par.shape[1] != w.shape[0]:
                raise ValueError("parameters and particle_weights have different lengths")
            hp = self._hargs.ptr
            args = (self._model_struct, _ptr(par), par.shape[1],
                    self.n_particles, _ptr(w), hp(self._setting_array(onesetting)),
                    hp(yy), None if s is None else hp(s),
                    None if rows is None else hp(rows), n, self._choke_value())
            if self._parameters is self._particles and self.tuning_parameters.get("fused_moments", True) \
                    and not self._strict_sums() and self.n_dims <= _lib.OBE_FAST_DIMS:
                # ... and the first moments of the posterior in the same pass over the cloud: the next
                # sweep's shift, mean(), std() and the noise-parameter variance need no launch of their own
                self._drop_speculative_sweep()
                # (the calls below arm the page-locked block that an un-awaited constraint mask — two updates in
                # a row on a noise-parameter object, no sweep in between — may still be delivering into)
                self._await_host_moments()
                if self._randoms_ahead_wanted():
                    # the random numbers this update's resample would take, enqueued beside the update itself
                    # (kept for the next resample if this update does not resample: particlepdf.py, "randoms ahead")
                    self._randoms_ahead_enqueue()
                if self.utility_method == "variance_full" and not self._sweeps.unavailable \
                        and self.tuning_parameters.get("speculative_sweep", _SPECULATIVE_DEFAULT) is not False:
                    self._update_then_speculate(args)       # (enqueued; what goes behind it is decided while it runs)
                else:
                    self._mlib.call("obe_bayes_update_model_moments", *args, _ptr(self._moments_dev), _ptr(self._ws),
                                    self._ws_bytes, self._hargs.ptr_keep(self._upd_host), self._stream())
                    self._after_weight_update(self._upd_host[1], moments_fresh=True)
            else:
                # (a stale `parameters` alias after set_pdf, obe_base.py:185,395: not the cloud the moments describe;
                # or a small cloud whose sums are formed in np.sum's order: tuning_parameters['strict_sums'])
                self._unfused_update(self._mlib, "obe_bayes_update_model", *args, _ptr(self._ws), self._ws_bytes,
                                     hp(self._host_out), self._stream())
                self._after_weight_update(self._host_out[1])
        else:
            if y_model_data is None:
                y_model_data = self.eval_over_all_parameters(onesetting)
            if self._likelihood_overridden():
                likyhd = self.likelihood(y_model_data, measurement_record)   # user NumPy code
                self.bayesian_update(likyhd)
            elif self._wide_channels:
                # more channels than one launch of the fused y-update takes: the likelihood in groups of channels
                # (a device array), then the update from it
                self.bayesian_update(self._likelihood_device(y_model_data, measurement_record))
            else:
                y_dev = self._y_to_device(y_model_data)
                n, yy, s, rows = self._likelihood_inputs(measurement_record)
                par = self._parameters.tensor()
                w = self._weights.tensor()
                self._unfused_update(self._lib, "obe_bayes_update_y", _ptr(y_dev), y_dev.shape[1], self.n_channels,
                                     _ptr(par), par.shape[1], self.n_particles, _ptr(w), _lib.host_ptr(yy),
                                     None if s is None else _lib.host_ptr(s),
                                     None if rows is None else _lib.host_ptr(rows), n, self._choke_value(),
                                     _ptr(self._ws), self._ws_bytes, _lib.host_ptr(self._host_out), self._stream())
                self._after_weight_update(self._host_out[1])
        self._parameters = self._particles
        if self.just_resampled:
            self.enforce_parameter_constraints()
        # (the cycle pattern the speculative sweep looks for: a full sweep of exactly this cloud comes next)
        self._sweeps.update_finished((self._particles.version, self._weights.version) if fused else None,
                                     self.just_resampled)
        if fused and self.just_resampled and self._speculation_wanted(after_resample=True):
            # the cloud is final (resampled, constrained): its sweep goes out now, behind the gather and the
            # moments, instead of after the caller's way back through opt_setting()
            self._drop_speculative_sweep()
            try:
                self._sweep_device(False, speculate="after_resample")
            except _lib.ObeHipError as exc:
                if not exc.refused_before_launch:
                    raise
                self._sweeps.library_refused()
        return _LazyState(self)

    # ------------------------------------------------------- speculative sweep
    # The reference's cycle is opt_setting -> measure -> pdf_update -> opt_setting ... (obe_base.py:733-756,
    # 340-399): after an update the next thing the device is asked for is, almost always, the sweep over the
    # updated cloud.  Once that pattern has been seen twice in a row, pdf_update() enqueues the update WITHOUT
    # waiting for its sums, enqueues that sweep right behind it, and only then waits for the update: the host
    # round trip of the update, the resample test, the caller's own work between the two calls and the launch
    # overhead of the sweep are hidden behind the sweep's kernels.  The update kernel leaves its resample
    # decision on the device; a sweep behind an update that resamples does nothing (the cloud is about to
    # change) and the real sweep is launched when it is asked for.  Everything is decided again on the host
    # from the delivered values: the speculative result is used only if the update says the sweep ran, the
    # cloud is still the one it swept and every input of the sweep is what a fresh launch would use — the
    # same kernels on the same data, hence the same bits.  tuning_parameters['speculative_sweep']: 'auto'
    # (default), True (from the first update on), False (never).  Measured (MI355X, tools/spec_cycles.py,
    # tools/shard_cycle.py): the plain cycle of 4096 settings x 262 144 particles 0.314 -> 0.288 ms, of one
    # rank's 8192 x 1 048 576 slice 1.877 -> 1.845 ms.
    def _speculation_wanted(self, after_resample=False):
        mode = self.tuning_parameters.get("speculative_sweep", _SPECULATIVE_DEFAULT)
        if not self._sweeps.speculation_wanted(mode, after_resample):      # (the policy: _sweepstate.py)
            return False
        if after_resample and self._parameters is not self._particles:
            return False
        return (self.utility_method == "variance_full" and self._utility_fusable()
                and not _overridden(self, "cost_estimate", OptBayesExpt)
                and self._noise_token() is not None
                and self._sweeps.form is Form.FAST
                and self.N_DRAWS <= self._ws_draws)

    def _randoms_ahead_wanted(self):
        """tuning_parameters['randoms_ahead']: False (default: measured, no gain — see _RANDOMS_AHEAD_DEFAULT), True,
        or 'auto': only where nothing else draws from self.rng between two resamples — the full sweep chosen by
        opt_setting() (the reference-semantics sweep takes N_DRAWS uniforms per cycle, good_setting() one) — and
        once the experiment has shown that it resamples at all."""
        mode = self.tuning_parameters.get("randoms_ahead", _RANDOMS_AHEAD_DEFAULT)
        if mode is False or not self.tuning_parameters["auto_resample"]:
            return False
        if mode is True:
            return True
        return (self.utility_method == "variance_full" and self._sweeps.resample_rate >= 0.1
                and getattr(self.get_setting, "__func__", None) is OptBayesExpt.opt_setting)

    def sweep_state(self):
        """A snapshot of what steers this object's sweeps (diagnostic): the kernel form in use, the shift
        hysteresis, the update -> sweep pattern and the sweep enqueued ahead, by name (_sweepstate.py)."""
        return self._sweeps.describe()

    speculation_state = sweep_state

    # (the names round-4 tests and tools know this state under)
    _sweep_unshifted = property(lambda self: self._sweeps.unshifted,
                                lambda self, v: setattr(self._sweeps, "unshifted", bool(v)))
    _sweep_safe_streak = property(lambda self: self._sweeps.safe_streak,
                                  lambda self, v: setattr(self._sweeps, "safe_streak", int(v)))
    _sweep_safe_run = property(lambda self: self._sweeps.safe_run,
                               lambda self, v: setattr(self._sweeps, "safe_run", int(v)))

    @property
    def _spec(self):
        """The sweep enqueued ahead as round-4 code saw it (None, or a dict with 'ran')."""
        t = self._sweeps.ticket
        if t is None:
            return None
        return dict(t.inputs, words=t.words, block=t.block, record=t.record, stream=t.stream,
                    ran={Pending.RAN: True, Pending.ABORTED: False}.get(self._sweeps.pending))

    def _noise_token(self):
        """What the utility's noise variance depends on besides the cloud (hashable), or None if that cannot
        be told without calling user code."""
        if _overridden(self, "yvar_noise_model", OptBayesExpt):
            return None
        dns = self.default_noise_std
        return dns.tobytes() if isinstance(dns, np.ndarray) else None

    def _update_then_speculate(self, args):
        tp = self.tuning_parameters
        st = self._stream()
        d = self.n_dims
        p_out = self._hargs.ptr_keep(self._upd_host)
        try:
            self._mlib.call("obe_bayes_update_model_moments_enqueue", *args, _ptr(self._moments_dev),
                            _ptr(self._ws), self._ws_bytes, p_out, 1 if tp["auto_resample"] else 0,
                            float(tp["resample_threshold"]), st)
        except _lib.ObeHipError as exc:
            if not exc.refused_before_launch:
                # a launch (or the device) failed AFTER pass A may have multiplied the weights in place: re-running
                # the update would apply the likelihood twice and hide the real error
                raise
            # refused before anything was launched (status -1: no arrival counter for this stream, a workspace
            # without the spare tail word): the weights are untouched — the plain form, now and from now on
            self._sweeps.library_refused()
            self._mlib.call("obe_bayes_update_model_moments", *args, _ptr(self._moments_dev), _ptr(self._ws),
                            self._ws_bytes, p_out, st)
            self._after_weight_update(self._upd_host[1], moments_fresh=True)
            return
        self._weights.mark_device_written()
        key = (self._particles.version, self._weights.version)
        self._mom_dev_key = key + (False,)        # on the device, in stream order: what the sweep reads
        if self._speculation_wanted():
            try:
                self._sweep_device(False, speculate=True)
            except _lib.ObeHipError as exc:
                if not exc.refused_before_launch:
                    raise
                self._sweeps.library_refused()        # (nothing of the sweep was enqueued: refused before any launch)
        self._lib.call("obe_host_words_wait", p_out, 5 + 4 * d, st)
        self._sweeps.update_delivered(resampled=self._upd_host[4 + 4 * d] != 0.0)
        self._mom_host_key = key + (False,)
        self._sumsq, self._sumsq_key = float(self._upd_host[1]), self._weights.version
        if tp["auto_resample"]:
            self.resample_test()

    def _sweep_inputs(self, shifted):
        """Everything the result of a full sweep of this object depends on besides the kernels: compared
        between the sweep enqueued ahead and the one being asked for (both must read the same)."""
        return dict(cloud=(self._particles.version, self._weights.version), shifted=bool(shifted),
                    noise=self._noise_token(), settings=(self._s_begin, self._s_end),
                    alias=self._parameters is self._particles,
                    cost_hook=_overridden(self, "cost_estimate", OptBayesExpt))

    def _drop_speculative_sweep(self):
        """Forget a speculative sweep nobody asked for (what is waited for: SweepState.drop)."""
        if self._sweeps.ticket is not None:
            self._sweeps.drop(self._stream().value)

    def _take_speculative_sweep(self, shifted):
        """The result of the speculative sweep if it is the sweep being asked for: (best, index, kappa) for an
        unsharded object, the device record for a sharded one; None (and the speculation forgotten) if not."""
        if self._sweeps.ticket is None:
            return None
        # (a stale `parameters` alias or a cost hook installed since makes the inputs differ: dropped)
        t = self._sweeps.take(self._sweep_inputs(shifted), self._stream().value)
        if t is None:
            return None
        if t.words is None:
            return t.record
        block = t.block
        return float(block[0]), int(block.view(np.int64)[1]), float(block[2])

    def _likelihood_overridden(self):
        return _overridden(self, "likelihood", OptBayesExpt)

    def _y_to_device(self, y_model_data):
        if isinstance(y_model_data, torch.Tensor):
            return y_model_data.to(self._device, torch.float64).reshape(self.n_channels, -1).contiguous()
        rows = [np.broadcast_to(np.asarray(r, dtype=np.float64), (self.n_particles,)) for r in y_model_data]
        return torch.from_numpy(np.array(rows[:self.n_channels], dtype=np.float64)).to(self._device)

    def enforce_parameter_constraints(self):
        """Stub for subclasses (obe_base.py:401-416); called after each resample."""
        pass

    def likelihood(self, y_model, measurement_record):
        """Gaussian likelihood of the measurement for every parameter sample
        (obe_base.py:418-461), computed on the device from ``y_model`` (C x N_p)."""
        return self._likelihood_device(y_model, measurement_record).cpu().numpy()

    def _likelihood_device(self, y_model, measurement_record):
        y_dev = self._y_to_device(y_model)
        n, yy, s, rows = self._likelihood_inputs(measurement_record)
        par = self._parameters.tensor()
        out = torch.empty(y_dev.shape[1], dtype=torch.float64, device=self._device)
        self._lib.call("obe_likelihood_y", _ptr(y_dev), y_dev.shape[1], self.n_channels, _ptr(par),
                       par.shape[1], y_dev.shape[1], _lib.host_ptr(yy),
                       None if s is None else _lib.host_ptr(s),
                       None if rows is None else _lib.host_ptr(rows), n, self._choke_value(), _ptr(out),
                       self._stream())
        return out

    # ----------------------------------------------------------------- utility
    def yvar_noise_model(self):
        """Measurement-noise variance used by the utility (obe_base.py:542-564)."""
        return self.default_noise_std ** 2

    def y_var_noise_model(self):
        return self.yvar_noise_model()

    def cost_estimate(self):
        """Cost of a measurement per setting (obe_base.py:566-577)."""
        return 1.0

    def _noise_var_device(self, values=False):
        """(tensor, ld): noise variance on the device — one value per channel (ld = 0)
        or, for an overriding yvar_noise_model() that returns per-setting values, a
        (C, n_local) array (ld = n_local).  (``values``: see the noise-parameter class, whose sweeps take the
        variance straight from the moment block.)"""
        if not _overridden(self, "yvar_noise_model", OptBayesExpt) and self._noise_cache is not None:
            dns = self.default_noise_std
            if isinstance(dns, np.ndarray) and dns.tobytes() == self._noise_src:
                return self._noise_dev, 0        # the class's own model of an unchanged default_noise_std
        nv = np.asarray(self.yvar_noise_model(), dtype=np.float64)
        c, ns = self.n_channels, self._n_settings
        per_channel = nv.size == 1 or nv.shape in ((c,), (c, 1))
        if per_channel:
            flat = np.array(np.broadcast_to(nv.reshape(-1), (c,)), dtype=np.float64)
            key = flat.tobytes()
            if self._noise_cache != key:
                self._noise_dev.copy_(torch.from_numpy(flat))
                self._noise_cache = key
            dns = self.default_noise_std
            # (the shortcut above is only for the class's own model: a replaced yvar_noise_model leaves no source)
            own = not _overridden(self, "yvar_noise_model", OptBayesExpt)
            self._noise_src = dns.tobytes() if own and isinstance(dns, np.ndarray) else None
            return self._noise_dev, 0
        full = np.broadcast_to(nv, (c, ns))[:, self._s_begin:self._s_end]
        t = torch.from_numpy(np.array(full)).to(self._device)
        return t, t.shape[1]

    def _cost_device(self, whole_grid=False):
        if not _overridden(self, "cost_estimate", OptBayesExpt):
            return None, 1.0
        cost = self.cost_estimate()
        if np.ndim(cost) == 0:
            return None, float(cost)
        c = np.array(np.broadcast_to(np.asarray(cost, dtype=np.float64), (self._n_settings,)))
        if whole_grid:
            return torch.from_numpy(c).to(self._device), 1.0
        return torch.from_numpy(c[self._s_begin:self._s_end].copy()).to(self._device), 1.0

    def _utility_fusable(self):
        return (self._device_model is not None
                and getattr(self.utility, "__func__", None) is OptBayesExpt.utility_variance
                and not _overridden(self, "yvar_from_parameter_draws", OptBayesExpt)
                and not _overridden(self, "eval_over_all_settings", OptBayesExpt))

    def _sweep_device(self, want_best, speculate=False):
        """K1 + K5 on this rank's settings slice.  Leaves yvar/utility on the device;
        returns (best value, best *global* index) if ``want_best``.  ``speculate``: pdf_update()'s launch of
        the sweep it expects to be asked for next (see _speculation_wanted) — enqueued, nothing read."""
        full = self.utility_method == "variance_full"
        sharded = self._shard is not None     # every sweep of a sharded object gathers: ranks stay in lockstep
        state = self._sweeps
        if full and not speculate:
            # update -> full sweep of exactly that cloud, twice in a row: the next update speculates
            state.full_sweep_requested((self._particles.version, self._weights.version))
        elif not speculate:
            self._drop_speculative_sweep()
        idx = None
        n_draws = 0
        if not full:
            if self.N_DRAWS > self._ws_draws:       # more draws than particles (N_DRAWS is a public attribute):
                self._alloc_scratch()               # the packed draws of the sweep need the room
            # consumes N_DRAWS uniforms (randdraw); its check of sum(w) waits for the sweep's own sync
            idx = self._draw_indices(self.N_DRAWS, defer_validation=True)
            n_draws = self.N_DRAWS
        n_local = self._s_end - self._s_begin
        res = self.__dict__.get("_sweep_result")
        if res is None:               # persistent host landing zone of the sweep result, pointers made once:
            block = _lib.pinned_array(4)          # {best, index bits, kappa} in three consecutive words
            best, best_idx, kappa = block[0:1], block.view(np.int64)[1:2], block[2:3]
            res = self._sweep_result = (best, best_idx, kappa, _lib.host_ptr(best), _lib.host_ptr(best_idx),
                                        _lib.host_ptr(kappa), block)
        best, best_idx, kappa, p_best, p_best_idx, p_kappa, block = res
        s_ptr = _P(self._settings_dev.data_ptr() + 8 * self._s_begin)

        if sharded and not speculate:
            every = int(self.tuning_parameters.get("replica_check_every", 64) or 0)
            self._sharded_sweeps += 1
            if every > 0 and self._sharded_sweeps % every == 0:
                self.check_replicas()
        result = {}
        # Nobody waits for this sweep: its caller wants the utility on the device (good_setting,
        # utility_variance), a draws-mode sweep is always shifted (kappa decides nothing) and the model's
        # fast form cannot leave its range.  Then the launch returns no host result and does not
        # synchronise; the deferred check of sum(w) happens at the caller's own synchronisation.
        checked = self._sweep_needs_range_check(n_draws)
        lazy = not want_best and not full and not sharded and not checked
        # (sharded: the 32-byte result record at the END of the workspace — include/obe_hip.h: OBE_WS_RESULT_TAIL —,
        # which the update calls between a sweep enqueued ahead and its collection do not touch)
        tail = self._ws[-6:-2] if sharded else None

        def launch(shifted, safe=False, speculative=False):
            # sharded: no host read here — the 32-byte result record is all-gathered from
            # device memory and read back once, together with the other ranks' records
            p, w = self._pw_tensors()
            cost_t, cost_s = self._cost_device()
            noise, noise_ld = self._noise_var_device()
            # last, so that the moments kernels and the sweep are enqueued back to back (every idle
            # microsecond before the sweep kernel also costs clock ramp-up inside it)
            mom = self._moments_on_device()
            no_host = sharded or lazy
            stream = self._stream()
            self._mlib.call("obe_sweep_utility", self._model_struct, s_ptr, self._n_settings, n_local,
                           _ptr(p), p.shape[1], self.n_particles, _ptr(w),
                           None if idx is None else _ptr(idx), n_draws, _ptr(mom),
                           (_lib.OBE_SWEEP_SHIFTED if shifted else 0) | (_lib.OBE_SWEEP_SAFE if safe else 0)
                           | (0 if not speculative else _lib.OBE_SWEEP_NOWAIT if speculative == "after_resample"
                              else _lib.OBE_SWEEP_SPECULATIVE),
                           _ptr(noise), noise_ld, None if cost_t is None else _ptr(cost_t), cost_s,
                           _ptr(self._yvar_dev), _ptr(self._utility_dev),
                           None if no_host else p_best,
                           None if no_host else p_best_idx,
                           None if no_host else p_kappa,
                           _ptr(self._ws), self._ws_bytes, stream)
            if speculative:
                # (waited for on the stream it was launched on; the sweep of a resampled cloud has nothing to guess)
                state.enqueued(Ticket(self._sweep_inputs(shifted), None if sharded else p_best, block,
                                      tail if sharded else None, stream, stream.value),
                               certain=speculative == "after_resample")
            else:
                deliver(None if not sharded else tail)

        def deliver(record):
            if lazy:
                result["best"] = None
                kappa[0] = 0.0
            elif sharded:
                val, gidx, k = self._shard.combine_records(record, self._n_settings)
                result["best"] = (val, gidx)
                kappa[0] = k
            else:
                result["best"] = (float(best[0]), int(best_idx[0]) + self._s_begin)

        # (sharded: kappa is the worst over all ranks, so every rank takes the same branch and the
        # collectives stay in step)
        # Shift policy (full sweep only).  The unshifted kernel saves one FP64 instruction
        # per evaluation (12 % of the sweep) but loses accuracy in proportion to
        # kappa = (mean of y)^2 / var, which every sweep reports: measured ~1e-15 * kappa
        # relative (worst case eps*kappa*sqrt(N)).  It is used only while the previous sweep saw
        # kappa < KAPPA_ENTER, and a sweep that comes back with kappa > KAPPA_LEAVE is repeated
        # with the shift: the variance is always good to a few 1e-12.  (SweepState.sweep_reported_kappa)
        mode = self.tuning_parameters.get("sweep_shift", "auto")
        shifted = state.shifted_for_next_sweep(mode, full)
        safe = False
        if speculate:
            launch(shifted, speculative=speculate)
            return None
        self._apply_range_hint()
        form = state.form_for_next_sweep()
        if lazy:
            launch(True)
            self.last_sweep = dict(shifted=True, kappa=float("nan"), safe=False)     # kappa was not read back
            return None
        if form is Form.FAST:
            taken = self._take_speculative_sweep(shifted) if full else None
            if taken is None:
                launch(shifted)
            else:
                deliver(taken if sharded else None)
            self._check_pending_total()
            poisoned = checked and bool(np.isnan(kappa[0]))
            if state.sweep_reported_kappa(mode, full, shifted, float(kappa[0])):
                shifted = True
                if not poisoned:          # (a poisoned sweep is repeated below anyway, with the twin)
                    launch(True)
                    poisoned = checked and bool(np.isnan(kappa[0]))
            if poisoned:
                # a model's branch-free batched divisions left their exact range somewhere (or the
                # model really produces NaN): repeat with its always-in-range twin
                safe = shifted = True
                launch(True, safe=True)
                state.fast_form_left_its_range()
            else:
                state.fast_form_held()
        else:
            # the fast form has left its range SAFE_STREAK sweeps in a row: that is a property of the
            # settings grid (its span against the model's width), not of one cloud — stop paying for a
            # fast attempt that is thrown away
            self._drop_speculative_sweep()
            safe = shifted = True
            launch(True, safe=True)
            self._check_pending_total()
        self.last_sweep = dict(shifted=shifted, kappa=float(kappa[0]), safe=safe)
        if want_best:
            return result["best"]
        return None

    def _settings_per_lane(self, n_draws=0):
        """Settings one lane of the sweep kernel owns on a slice of this job: at most (n_draws = 0, what a full
        sweep gets), or in a reference-semantics sweep of n_draws draws (1 on the one-workgroup path).  A sharded
        object answers with the LARGEST figure over all ranks' slices (they differ by at most one setting, which
        can straddle a threshold): everything decided from it — whether kappa can say "out of range" at all — is
        then the same decision on every rank, and the ranks' collectives stay in step."""
        cache = self.__dict__.setdefault("_spt_local", {})
        spt = cache.get(n_draws)
        if spt is None:
            lengths = {max(self._s_end - self._s_begin, 1)}
            if self._shard is not None:
                base, extra = divmod(self._n_settings, self._shard.world_size)       # (dist.shard_bounds)
                lengths = {max(base, 1), base + 1} if extra else {max(base, 1)}
            spt = cache[n_draws] = max(int(self._mlib.cdll.obe_sweep_settings_per_lane_for(n, n_draws))
                                       for n in lengths)
        return spt

    def _sweep_needs_range_check(self, n_draws=0):
        """Whether this sweep's fast form can leave its exact range at all (then kappa is read back and a NaN
        repeats the sweep with the model's twin): the model has such a pair of forms and a lane owns enough
        settings for its batched arithmetic to exist (DeviceModel.safe_sweep_min_spt) — on ANY rank of a
        sharded job, whose kappa is the worst over all ranks."""
        dm = self._device_model
        return bool(dm is not None and dm.safe_sweep and self._settings_per_lane(n_draws) >= dm.safe_sweep_min_spt)

    def _apply_range_hint(self):
        """Models with a ``range_hint`` (models.py) predict from the settings grid and the extremes
        of a cloud the host holds (the prior, set_pdf, user-written particles) whether their fast
        sweep form stays in range: if not, the sweep starts with the safe form instead of finding
        out by a poisoned fast attempt (the kernel's range check remains the guarantee; a wrong
        'in range' costs one repeated sweep, a wrong 'out of range' is retried after SAFE_RETRY).

        A sharded object must take this decision identically on every rank (the form decides how many sweeps, and
        with them how many all-gathers, a request takes): it looks only at clouds whose values were WRITTEN by
        host code — which every rank of an SPMD script did alike —, never at one that merely happens to have been
        read back on this rank, and at the whole settings grid instead of its own slice."""
        hint = getattr(self._device_model, "range_hint", None)
        pm = self._particles
        state = self._sweeps
        sharded = self._shard is not None
        if hint is None or not pm._host_valid or state.range_hint_key == pm.version:
            return
        if sharded and not pm.host_born:
            return
        state.range_hint_key = pm.version
        n_local = self._s_end - self._s_begin
        if n_local <= 0 and not sharded:
            return
        spt = self._settings_per_lane()
        settings = self.allsettings if sharded else self.allsettings[:, self._s_begin:self._s_end]
        state.range_hint(hint(settings, pm._host, self.cons, spt))

    def yvar_from_parameter_draws(self):
        """Variance of the model output over parameter draws, per setting: (C, N_s)
        (obe_base.py:463-489; all particles, weighted, for 'variance_full')."""
        if self._device_model is not None and not _overridden(self, "eval_over_all_settings", OptBayesExpt):
            self._sweep_device(False)
            return self._gather_settings(self._yvar_dev)
        # host-callable model: the user's function fills the y-space, the device reduces it
        paramsets = self.randdraw(self.N_DRAWS).T
        for i, oneparamset in enumerate(paramsets):
            self.utility_y_space[i] = self.eval_over_all_settings(oneparamset)
        ysp = torch.from_numpy(np.ascontiguousarray(self.utility_y_space)).to(self._device)
        yvar = torch.empty((self.n_channels, self._n_settings), dtype=torch.float64, device=self._device)
        self._lib.call("obe_yspace_variance", _ptr(ysp), self.N_DRAWS, self.n_channels, self._n_settings,
                       _ptr(yvar), self._stream())
        return yvar.cpu().numpy()

    def utility_variance(self):
        """Variance-approximation utility per setting, (N_s,) (obe_base.py:628-655)."""
        if self._utility_fusable() and not _overridden(self, "utility_variance", OptBayesExpt):
            self._sweep_device(False)
            return self._gather_settings(self._utility_dev.reshape(1, -1))[0]
        var_p = self.yvar_from_parameter_draws()
        return self._utility_from_host_yvar(var_p)

    def _utility_from_host_yvar(self, var_p):
        """sum_c var_p / var_n / cost on the device from a (C, N_s) variance: a host array over
        all settings, or a device tensor over this rank's settings slice (then the slices'
        utilities are gathered, so every rank returns the full (N_s,) vector)."""
        local = isinstance(var_p, torch.Tensor)
        if local:
            yv = var_p
            n = self._s_end - self._s_begin
        else:
            # a variance computed on the host covers ALL settings — on every rank of a sharded object alike (user
            # code runs identically on the replicas): the utility of the whole grid is formed here, nothing is gathered
            n = self._n_settings
            yv = torch.from_numpy(np.array(np.broadcast_to(np.asarray(var_p, dtype=np.float64),
                                                           (self.n_channels, n)))).to(self._device)
        noise, noise_ld = self._noise_var_device(values=True)
        if not local and noise_ld > 0 and noise_ld != n:
            raise NotImplementedError("a host-side variance with per-setting noise values on a sliced settings axis: "
                                      "return the variance as a device tensor of this rank's slice, or build the "
                                      "object without settings_shard")
        cost_t, cost_s = self._cost_device(whole_grid=not local)
        util = torch.empty(max(n, 1), dtype=torch.float64, device=self._device)
        if n > 0:
            self._lib.call("obe_utility_argmax", _ptr(yv), self.n_channels, n, _ptr(noise), noise_ld,
                           None if cost_t is None else _ptr(cost_t), cost_s,
                           _ptr(util), None, None, _ptr(self._ws), self._ws_bytes, self._stream())
        if not local:
            host = util[:n].cpu().numpy()
            self._check_pending_total()
            return host
        return self._gather_settings(util[:n].reshape(1, -1))[0]

    # ---- the y-space utilities (obe_base.py:491-535, 602-626, 657-720) ----
    def _yspace_device(self):
        """utility_y_space on the device, (N_DRAWS, C, n): the model over this rank's settings
        (all of them unless sharded) for N_DRAWS fresh weighted draws (consumes N_DRAWS uniforms
        of self.rng — the same ones on every rank of a sharded object)."""
        nd, c = self.N_DRAWS, self.n_channels
        if self._device_model is not None and not _overridden(self, "eval_over_all_settings", OptBayesExpt):
            idx = self._draw_indices(nd)
            p = self._particles.tensor()
            n = self._s_end - self._s_begin
            ysp = torch.empty((nd, c, max(n, 1)), dtype=torch.float64, device=self._device)
            if n > 0:
                s_ptr = _P(self._settings_dev.data_ptr() + 8 * self._s_begin)
                self._mlib.call("obe_eval_draws", self._model_struct, s_ptr, self._n_settings, n, _ptr(p),
                                p.shape[1], self.n_particles, _ptr(idx), nd, _ptr(ysp), self._stream())
            return ysp if n else ysp[:, :, :0]
        # (a host-callable model: the whole grid on every rank — __init__ left the settings axis unsliced)
        ns = self._n_settings
        paramsets = self.randdraw(nd).T              # host-callable model: the user's function fills it
        for i, oneparamset in enumerate(paramsets):
            self.utility_y_space[i] = self.eval_over_all_settings(oneparamset)
        return torch.from_numpy(np.ascontiguousarray(self.utility_y_space)).to(self._device)

    def _column_entropy(self, ysp, as_variance):
        """scipy.stats.differential_entropy(ysp, axis=0) per column, on the device."""
        nd = ysp.shape[0]
        cols = ysp[0].numel()
        m = int(np.floor(np.sqrt(nd) + 0.5))
        if not 2 <= 2 * m < nd:
            raise ValueError(f"Window length ({m}) must be positive and less "
                             f"than half the sample size ({nd}).")
        out = torch.empty(ysp.shape[1:], dtype=torch.float64, device=self._device)
        if cols:
            scratch = torch.empty(nd * cols, dtype=torch.float64, device=self._device)
            self._lib.call("obe_yspace_entropy", _ptr(ysp), nd, cols, 1 if as_variance else 0, _ptr(scratch),
                           _ptr(out), self._stream())
        return out

    def yvar_from_entropy(self):
        """Variance of the normal distribution with the model outputs' differential entropy,
        per setting (obe_base.py:491-518)."""
        return self._gather_settings(self._yvar_from_entropy_device())

    def _yvar_from_entropy_device(self):
        return self._column_entropy(self._yspace_device(), True)

    def yvar_max_min(self):
        """(max - min)^2 of the model outputs over the draws (obe_base.py:520-535)."""
        return self._gather_settings(self._yvar_max_min_device())

    def _yvar_max_min_device(self):
        ysp = self._yspace_device()
        out = torch.empty(ysp.shape[1:], dtype=torch.float64, device=self._device)
        if out.numel():
            self._lib.call("obe_yspace_maxmin", _ptr(ysp), ysp.shape[0], ysp[0].numel(), _ptr(out), self._stream())
        return out

    def utility_max_min(self):
        """obe_base.py:602-626."""
        if _overridden(self, "yvar_max_min", OptBayesExpt):
            return self._utility_from_host_yvar(self.yvar_max_min())
        return self._utility_from_host_yvar(self._yvar_max_min_device())

    def utility_pseudo(self):
        """obe_base.py:657-686."""
        if _overridden(self, "yvar_from_entropy", OptBayesExpt):
            return self._utility_from_host_yvar(self.yvar_from_entropy())
        return self._utility_from_host_yvar(self._yvar_from_entropy_device())

    def utility_full_kld(self):
        """exp(H(y + noise) - H(noise)) - 1, shape (C, N_s) (obe_base.py:688-720).  The
        N_DRAWS*C standard normals come from the module-level ``rng`` as in the reference."""
        nd, c = self.N_DRAWS, self.n_channels
        ysp = self._yspace_device()
        nva = self._rank0_values(rng.normal(0, 1.0, nd * c))
        nvb = nva.reshape((c, nd))
        noisevalues = np.ascontiguousarray((nvb * np.sqrt(self.yvar_noise_model())).T)     # (N_d, C) glue
        noise_dev = torch.from_numpy(noisevalues).to(self._device)
        n = ysp.shape[2]                                                # this rank's settings
        util = torch.empty((c, n), dtype=torch.float64, device=self._device)
        if n:
            self._lib.call("obe_yspace_add_noise", _ptr(ysp), nd, c, n, _ptr(noise_dev), self._stream())
            h_y = self._column_entropy(ysp, False)                          # (C, n)
            h_n = self._column_entropy(noise_dev.reshape(nd, c, 1), False)  # (C, 1)
            self._lib.call("obe_kld_utility", _ptr(h_y), c, n, _ptr(h_n), _ptr(util), self._stream())
        return self._gather_settings(util)

    def _gather_settings(self, local):
        """Host (rows, N_s) array from this rank's (rows, n_local) device slice."""
        if self._shard is None or self._s_end - self._s_begin == self._n_settings and self._device_model is None:
            host = local.cpu().numpy()
            self._check_pending_total()       # (the copy synchronised: a deferred check of sum(w) can run)
            return host
        return self._shard.gather_rows(local, self._n_settings)

    # --------------------------------------------------------------- selection
    def utility(self):
        """The utility of every setting, (N_s,).  Like the reference's placeholder of this name
        (obe_base.py:579-600) it is replaced per object in ``__init__`` by the method that
        ``utility_method`` names (``utility_variance`` by default)."""
        raise NotImplementedError("utility is bound in __init__ (utility_method)")

    def get_setting(self):
        """The next setting by the ``selection_method`` chosen at construction: ``opt_setting``,
        ``good_setting`` or ``random_setting`` (obe_base.py:722-731; bound per object in ``__init__``)."""
        raise NotImplementedError("get_setting is bound in __init__ (selection_method)")

    def opt_setting(self):
        """The setting with maximum utility (obe_base.py:733-756)."""
        if self._utility_fusable():
            val, bestindex = self._sweep_device(True)      # global over all ranks when sharded
        else:
            utility = np.asarray(self.utility(), dtype=np.float64)       # overridden / y-space utility
            self.last_utility = utility
            u = torch.from_numpy(np.ascontiguousarray(utility).reshape(-1)).to(self._device)
            best = np.zeros(1)
            best_idx = np.zeros(1, dtype=np.int64)
            self._lib.call("obe_argmax", _ptr(u), u.numel(), _lib.host_ptr(best), _lib.host_ptr(best_idx),
                           _ptr(self._ws), self._ws_bytes, self._stream())
            bestindex = int(best_idx[0])
        bestvalues = self.allsettings[:, bestindex]
        self.last_setting_index = bestindex
        return tuple(bestvalues)

    def good_setting(self, pickiness=None):
        """A setting drawn with probability ~ utility**pickiness (obe_base.py:758-789)."""
        if pickiness is None:
            pickiness = self.pickiness
        if self._utility_fusable():
            self._sweep_device(False)
            u = self._utility_dev
            if self._shard is not None:
                # every rank assembles the full utility vector (N_s doubles over xGMI) and then
                # draws from it exactly as a single GPU would: same sums, same index everywhere
                full = self._gather_settings(self._utility_dev.reshape(1, -1))[0]
                u = torch.from_numpy(np.ascontiguousarray(full)).to(self._device)
        else:
            u = torch.from_numpy(np.ascontiguousarray(self.utility(), dtype=np.float64).reshape(-1)).to(self._device)
        n = u.numel()
        bufs = self.__dict__.get("_good_bufs")
        if bufs is None or bufs[0].numel() != n:
            # scratch of the selection, made once; the chosen index lands in page-locked host memory,
            # which the search kernel writes through its device address (no copy back)
            # (through the address the device sees it under: obe_host_device_pointer)
            idx_host = _lib.pinned_array(1, np.int64)
            bufs = self._good_bufs = (torch.empty(n, dtype=torch.float64, device=self._device),
                                      torch.empty(n, dtype=torch.float64, device=self._device),
                                      idx_host, _lib.device_ptr_of_pinned(self._lib, idx_host), _lib.host_ptr(idx_host))
        prob, cdf, idx_host, idx_dev, idx_hptr = bufs
        self._lib.call("obe_power_normalize", _ptr(u), n, float(pickiness), _ptr(prob), _ptr(self._ws),
                       self._ws_bytes, self._stream())
        uni = np.atleast_1d(self.rng.random())
        # CDF of the selection probabilities + the search for one uniform (passed by value); sum(p) lands in
        # page-locked memory for numpy's validation of p (Generator.choice, obe_base.py:785)
        total = self._total_pinned
        stream = self._stream()
        self._lib.call("obe_host_word_arm", idx_hptr)        # the index is the search kernel's last word: watched, not synchronised
        total_ptr = self._total_ptrs[1]
        self._lib.call("obe_host_word_arm", total_ptr)       # ... and so is sum(p), a store of its own to a line of its own:
        self._lib.call("obe_draw_indices", _ptr(prob), n, 0, 0, _ptr(cdf), _lib.host_ptr(uni), 1,
                       idx_dev, total_ptr, _ptr(self._ws), self._ws_bytes, stream)
        self._lib.call("obe_host_word_wait", idx_hptr, stream)
        self._lib.call("obe_host_word_wait", total_ptr, stream)   # (the order of two stores is not the order of their arrival)
        self._check_pending_total()
        try:
            self._validate_total(float(total[1]))      # all-zero / NaN utilities: p = 0/0, ValueError in the reference
        except ValueError:
            # numpy validates p before it draws: give the uniform back (one raw value of a PCG64 stream)
            if _devrng.pcg64_state(self.rng) is not None:
                self.rng.bit_generator.advance((1 << 128) - 1)
            raise
        goodindex = int(idx_host[0])
        self.last_setting_index = goodindex
        return tuple(self.allsettings[:, goodindex])

    def random_setting(self):
        """A uniformly random setting (obe_base.py:791-805)."""
        settingindex = int(self._rank0_values(rng.choice(self.setting_indices))[0])
        self.last_setting_index = settingindex
        return self.allsettings[:, settingindex]

    def _rank0_values(self, values):
        """Draws the reference takes from its *module-level* generators (obe_base.py:18:
        random_setting, the full_kld noise; obe_sweeper.py:3-6) are not reproducible across the
        processes of a sharded object: every process has its own unseeded module generator, ranks
        would pick different settings and the replicated clouds would drift apart.  Every rank draws
        (its generator moves as in an unsharded run) and rank 0's values are used everywhere — one
        tiny broadcast; a run whose rank 0 seeds the module generator like an unsharded run
        reproduces it.  Unsharded: the values pass through."""
        values = np.atleast_1d(np.asarray(values))
        if self._shard is None:
            return values
        return self._shard.broadcast_from_rank0(values, self._device)

    def _model_output_len(self):
        """Number of output channels of a host-callable model, by a trial evaluation
        (obe_base.py:807-824)."""
        trial_rng = np.random.default_rng()
        settingindex = trial_rng.choice(self.setting_indices)
        one_setting = self.allsettings[:, settingindex]
        one_param_set = self.randdraw(n_draws=1)
        singleshot = self.model_function(one_setting, one_param_set, self.cons)
        return len(np.atleast_1d(singleshot))
