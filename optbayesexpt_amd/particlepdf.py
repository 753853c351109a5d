"""ParticlePDF — the weighted-particle posterior, resident on one MI355X.

Same constructor, attributes and methods as the reference class
(optbayesexpt/particlepdf.py:12-345); every numerical step is a HIP kernel in
libobe_hip.so reached through ctypes.  The host keeps only what must stay on the host
for stream parity with the reference: the NumPy ``Generator`` (``self.rng``) that
produces the uniforms and normals of a resample, and the D x D SVD of the nudge
covariance (LAPACK, exactly as ``Generator.multivariate_normal`` does it).
"""
import ctypes
import warnings

import numpy as np
import torch

from . import _devrng, _lib
from ._mirror import Mirror

_P = ctypes.c_void_p


SQRT_EPS = float(np.sqrt(np.finfo(np.float64).eps))   # numpy's tolerance on sum(p) in choice()

#: buffer sets of random chains enqueued ahead (ParticlePDF._randoms_ahead_enqueue) whose kernels may still run
_AHEAD_IN_FLIGHT = []


def _still_armed(pin_i):
    words = _lib.audit.raw(pin_i).view(np.int64)           # (a poll of armed words, on purpose)
    return bool(words[0] == _lib.HOST_SENTINEL or words[1] == _lib.HOST_SENTINEL)


def _ptr(t):
    return _P(t.data_ptr())


class ParticlePDF:
    """A probability distribution represented by weighted samples ("particles").

    Arguments and keyword arguments are those of the reference (particlepdf.py:79-80);
    ``use_jit`` is accepted and ignored (the HIP kernels replace the numba hooks).
    ``device`` (extension) selects the GPU; default: the current torch device.

    Extension, ``tuning_parameters['strict_cdf']`` (default ``False``): build the
    resampling CDF in np.cumsum's serial rounding order (bit-identical CDF, ~ms at 1e6
    particles) instead of the parallel blocked scan.
    Extension, ``tuning_parameters['strict_sums']`` (``'auto'`` — on up to 4096 particles —, ``True``, ``False``):
    the update's sum(t) and the sum(w^2) of the resample test are formed in the order np.sum adds (particlepdf.py:138,
    243; csrc/obe_update.hip: numpy_order_sum_kernel), so that the normalised weights are the reference's bits —
    its own unit tests compare them with assert_array_equal.  One workgroup: for small clouds.
    Extension, ``tuning_parameters['resample_method']`` (default ``'multinomial'``, the
    reference's ``rng.choice``): ``'systematic'`` draws the N new particles at the stratified
    CDF points (i + u0)/N from ONE uniform of ``self.rng`` (lower resampling variance; the
    nudge normals follow as usual).
    """

    def __init__(self, prior, a_param=0.98, resample_threshold=0.5,
                 auto_resample=True, scale=True, use_jit=True, device=None):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("optbayesexpt_amd needs a HIP device (MI355X / gfx950); "
                               "there is no CPU fallback")
        self._device = torch.device(device) if device is not None else \
            torch.device("cuda", torch.cuda.current_device())
        if self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        self._device_index = self._device.index
        # kernels are launched with this object's device current, whatever the caller's is
        self._lib = _lib.DeviceBound(lib, self._device)

        #: dict: a_param / resample_threshold / auto_resample / scale, read at call time
        self.tuning_parameters = {"a_param": a_param,
                                  "resample_threshold": resample_threshold,
                                  "auto_resample": auto_resample,
                                  "scale": scale}
        self._set_cloud(prior, None)
        #: bool: True if the last bayesian_update() resampled
        self.just_resampled = False
        #: numpy Generator feeding randdraw() and resample(); reseed by assignment
        self.rng = np.random.default_rng()
        self._host_out = _lib.pinned_array(16)       # page-locked: the folding kernels write their scalars here

    # ------------------------------------------------------------------ state
    def _set_cloud(self, samples, weights):
        host = np.asarray(samples)
        if host.ndim != 2:
            raise ValueError("prior/samples must be n_dims x n_particles")
        self._particles = Mirror(self._device, host=host)
        self.n_particles = host.shape[-1]
        self.n_dims = host.shape[0]
        if self.n_dims > _lib.OBE_CLOUD_MAX_DIMS:
            raise ValueError(f"at most {_lib.OBE_CLOUD_MAX_DIMS} parameters are supported on the device")
        self._weights = Mirror(self._device, host=np.ones(self.n_particles) / self.n_particles)
        self._alloc_scratch()
        if weights is not None:
            # weights / np.sum(weights) (particlepdf.py:171) as a device pass: uniform
            # weights times the given array, normalised by its sum
            self._host_out = _lib.pinned_array(16)       # page-locked: the folding kernels write their scalars here
            self._weights = Mirror(self._device, host=np.ones(self.n_particles))
            lik = torch.from_numpy(np.array(weights, dtype=np.float64)).to(self._device)
            w = self._weights.tensor()
            self._unfused_update(self._lib, "obe_bayes_update_lik", _ptr(lik), self.n_particles, _ptr(w), _ptr(self._ws),
                                 self._ws_bytes, _lib.host_ptr(self._host_out), self._stream())
            self._weights.mark_device_written()

    #: clouds up to this size sum the update in np.sum's order unless tuning_parameters['strict_sums'] says otherwise
    STRICT_SUMS_MAX = 4096

    def _strict_sums(self):
        mode = self.tuning_parameters.get("strict_sums", "auto")
        return self.n_particles <= self.STRICT_SUMS_MAX if mode == "auto" else bool(mode)

    def _unfused_update(self, lib, name, *args):
        """One of the unfused update calls (obe_bayes_update_model / _y / _lik), in np.sum's order of additions
        when this object asks for it (the library's switch is per thread: set for this call, cleared behind it)."""
        if not self._strict_sums():
            return lib.call(name, *args)
        lib.cdll.obe_strict_sums(1)
        try:
            return lib.call(name, *args)
        finally:
            lib.cdll.obe_strict_sums(0)

    def _scratch_dims(self):
        """(n_settings, n_channels) the workspace must cover; OptBayesExpt overrides."""
        return 1, 1

    def _alloc_scratch(self):
        n, d = self.n_particles, self.n_dims
        n_settings, n_channels = self._scratch_dims()
        # the sweep packs its draws into the workspace: all particles, or N_DRAWS of them
        self._ws_draws = max(n, int(getattr(self, "N_DRAWS", 0) or 0))
        libs = {id(x): x for x in (self._lib, getattr(self, "_mlib", self._lib))}.values()   # a plugin knows its
        sizes = [(n, n_settings), (self._ws_draws, n_settings)]                                # model's packed width
        n_local = getattr(self, "_s_end", n_settings) - getattr(self, "_s_begin", 0)
        if 0 < n_local < n_settings:       # a settings shard plans its sweep for its own slice
            sizes += [(n, n_local), (self._ws_draws, n_local)]
        nbytes = max(lib.workspace_bytes(a, b, n_channels, d) for lib in libs for a, b in sizes)
        self._ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=self._device)
        self._ws_bytes = self._ws.numel() * 8
        self._moments_dev = torch.zeros(self._lib.moments_len(d), dtype=torch.float64, device=self._device)
        # page-locked landing zone of the fused update: [0] sum t, [1] sum w'^2, [2:] the K3 block —
        # the host copy of the moments is that tail, whichever call fills it
        self._upd_host = _lib.pinned_array(2 + self._lib.moments_len(d))
        self._moments_host = self._upd_host[2:]
        self._mom_host_key = None      # (particles version, weights version, has_cov) of the host copy
        self._mom_dev_key = None       # same, for the device copy
        self._cdf_dev = torch.empty(n, dtype=torch.float64, device=self._device)
        self._cdf_key = None           # (weights version, strict)
        # [0] the asynchronous sum(w) of a small draw, [1] sum(p) of good_setting(): page-locked, each watched on its own
        self._total_pinned = _lib.pinned_array(8)
        self._total_ptrs = (_P(self._total_pinned.ctypes.data), _P(self._total_pinned.ctypes.data + 8))
        self._pending_total = None     # (generator state before the draw,) while that sum is unchecked
        self._sumsq_key = None         # weights version for which _sumsq is valid
        self._sumsq = None
        self.last_draw_indices_device = None

    @property
    def rng(self):
        """The numpy ``Generator`` behind ``randdraw()`` and ``resample()`` (particlepdf.py:142-145):
        a public attribute of the reference that callers reseed by assignment."""
        return self._rng

    @rng.setter
    def rng(self, value):
        self._rng = value
        self._rng_assigned()

    def _rng_assigned(self):
        """Hook: OptBayesExpt makes the replicas of a sharded object adopt rank 0's generator."""

    @property
    def particles(self):
        """``n_dims x n_particles`` ndarray (host mirror of the device array)."""
        return self._particles.host()

    @particles.setter
    def particles(self, value):
        value = np.asarray(value)
        if value.ndim != 2:
            raise ValueError("particles must be n_dims x n_particles")
        resized = (value.shape[0], value.shape[-1]) != (self.n_dims, self.n_particles)
        self._particles.set_host(value)
        self.n_particles = value.shape[-1]
        self.n_dims = value.shape[0]
        if resized:
            # the workspace, CDF and moment blocks are sized for the cloud; the weights keep their
            # length (as in the reference) and a mismatch is reported by the next kernel call
            if self.n_dims > _lib.OBE_CLOUD_MAX_DIMS:
                raise ValueError(f"at most {_lib.OBE_CLOUD_MAX_DIMS} parameters are supported on the device")
            self._alloc_scratch()

    @property
    def particle_weights(self):
        """ndarray of probability weights; in-place writes are tracked and uploaded."""
        return self._weights.host()

    @particle_weights.setter
    def particle_weights(self, value):
        self._weights.set_host(np.asarray(value))

    @property
    def _particle_indices(self):
        return np.arange(self.n_particles, dtype="int")

    def _stream(self):
        """The torch current stream of the object's device, as the C ABI wants it."""
        try:            # the raw query (0.3 us) instead of building a torch.cuda.Stream object (2 us)
            return _P(torch._C._cuda_getCurrentRawStream(self._device_index))
        except AttributeError:
            return _P(torch.cuda.current_stream(self._device).cuda_stream)

    def _pw_tensors(self):
        p, w = self._particles.tensor(), self._weights.tensor()
        if p.shape[1] != w.shape[0]:
            raise ValueError("particles and particle_weights have different lengths")
        return p, w

    # ---------------------------------------------------------------- set_pdf
    def set_pdf(self, samples, weights=None):
        """Re-initialise the distribution (particlepdf.py:147-171)."""
        samples = np.asarray(samples)
        if weights is not None:
            if len(weights) != samples.shape[-1]:
                raise ValueError("Length of weights does not match the number of particles.")
            weights = np.asarray(weights, dtype=np.float64)
        self._set_cloud(samples, weights)

    # ---------------------------------------------------------------- moments
    def _moments(self, want_cov):
        """Host copy of the K3 output block (computes it if stale); synchronises."""
        key = (self._particles.version, self._weights.version)
        hk = self._mom_host_key
        self._await_host_moments()
        if hk is not None and hk[:2] == key and (hk[2] or not want_cov):
            return self._moments_host
        p, w = self._pw_tensors()
        # the first moments are already there, on the device and on the host (the last update left them, or
        # an earlier mean()/std()): only the covariance pass is missing
        have_first = want_cov and hk is not None and hk[:2] == key and self._mom_dev_key is not None \
            and self._mom_dev_key[:2] == key
        self._lib.call("obe_moments", _ptr(p), p.shape[1], self.n_dims, self.n_particles, _ptr(w),
                       (2 if have_first else 1) if want_cov else 0, _ptr(self._moments_dev),
                       _lib.host_ptr(self._moments_host), _ptr(self._ws), self._ws_bytes, self._stream())
        self._mom_host_key = self._mom_dev_key = key + (bool(want_cov),)
        return self._moments_host

    def _await_host_moments(self):
        """A call that delivers the K3 block to the host without being waited for (the constraint mask with its
        first moments) leaves the armed words to wait for here, before the host copy is read."""
        pending = self.__dict__.get("_mom_host_wait")
        if pending is not None:
            self._mom_host_wait = None
            for ptr, n_words in pending:
                self._lib.call("obe_host_words_wait", ptr, n_words, self._stream())

    def _moments_on_device(self):
        """Device copy of the first moments, current for the present cloud (no sync)."""
        key = (self._particles.version, self._weights.version)
        dk = self._mom_dev_key
        if dk is None or dk[:2] != key:
            p, w = self._pw_tensors()
            self._lib.call("obe_moments", _ptr(p), p.shape[1], self.n_dims, self.n_particles, _ptr(w),
                           0, _ptr(self._moments_dev), None, _ptr(self._ws), self._ws_bytes, self._stream())
            self._mom_dev_key = key + (False,)
        return self._moments_dev

    def mean(self):
        """Weighted mean, shape ``(n_dims,)`` (particlepdf.py:173-183)."""
        d = self.n_dims
        return self._moments(False)[2:2 + d].copy()

    def covariance(self):
        """Weighted covariance ``(n_dims, n_dims)`` (particlepdf.py:185-198)."""
        d = self.n_dims
        m = self._moments(True)
        return m[2 + 4 * d:2 + 4 * d + d * d].reshape((d, d)).copy()

    def std(self):
        """Per-parameter standard deviation (particlepdf.py:200-214)."""
        d = self.n_dims
        return self._moments(False)[2 + 3 * d:2 + 4 * d].copy()

    # ------------------------------------------------------------ Bayes update
    def bayesian_update(self, likelihood):
        """weights <- normalised(weights * likelihood), then the resample test
        (particlepdf.py:216-234).  ``likelihood`` is an ndarray or a device tensor."""
        if isinstance(likelihood, torch.Tensor):
            lik = likelihood.to(device=self._device, dtype=torch.float64).contiguous()
        else:
            lik = torch.from_numpy(np.array(
                np.broadcast_to(np.asarray(likelihood, dtype=np.float64), (self.n_particles,)))).to(self._device)
        w = self._weights.tensor()
        self._unfused_update(self._lib, "obe_bayes_update_lik", _ptr(lik), self.n_particles, _ptr(w), _ptr(self._ws),
                             self._ws_bytes, _lib.host_ptr(self._host_out), self._stream())
        self._after_weight_update(self._host_out[1])

    def _after_weight_update(self, sum_w2, moments_fresh=False):
        self._weights.mark_device_written()
        if moments_fresh:     # the update also left the first moments of the new weights (device and host)
            self._mom_host_key = self._mom_dev_key = (self._particles.version, self._weights.version, False)
        self._sumsq, self._sumsq_key = float(sum_w2), self._weights.version
        if self.tuning_parameters["auto_resample"]:
            self.resample_test()

    def _sum_w2(self):
        if self._sumsq_key != self._weights.version:
            w = self._weights.tensor()
            self._lib.call("obe_weight_sums", _ptr(w), self.n_particles, _ptr(self._ws), self._ws_bytes,
                           _lib.host_ptr(self._host_out), self._stream())
            self._sumsq, self._sumsq_key = float(self._host_out[0]), self._weights.version
        return self._sumsq

    def resample_test(self):
        """Resample if the effective particle number is low (particlepdf.py:236-258)."""
        s2 = float(self._sum_w2())
        n_eff = 1.0 / s2 if s2 != 0.0 else float("inf")        # np.float64(1) / np.float64(0) = inf
        self.last_n_eff = n_eff
        if n_eff < 0.1 * self.n_particles:
            warnings.warn("\nParticle filter rejected > 90 % of particles. "
                          f"N_eff = {n_eff:.2f}. "
                          "Particle impoverishment may lead to errors.",
                          RuntimeWarning)
            self._resample_reported()
        elif n_eff / self.n_particles < self.tuning_parameters["resample_threshold"]:
            self._resample_reported()
        else:
            self.just_resampled = False

    def _resample_reported(self):
        """The resample of resample_test(): the only one whose outcome is reported through ``just_resampled`` — and
        therefore the only one a caller (OptBayesExpt.pdf_update) is certain to follow with its parameter
        constraints (see OptBayesExptNoiseParameter._resample_mask_rows)."""
        self._in_reported_resample = True
        try:
            self.resample()
        finally:
            self._in_reported_resample = False
        self.just_resampled = True

    # --------------------------------------------------------------- resample
    def _cdf(self):
        strict = bool(self.tuning_parameters.get("strict_cdf", False))
        key = (self._weights.version, strict)
        if self._cdf_key != key:
            w = self._weights.tensor()
            if self._cdf_dev.numel() != self.n_particles:
                self._cdf_dev = torch.empty(self.n_particles, dtype=torch.float64, device=self._device)
            self._lib.call("obe_weight_cdf", _ptr(w), self.n_particles, 1 if strict else 0,
                           _ptr(self._cdf_dev), _lib.host_ptr(self._host_out), _ptr(self._ws),
                           self._ws_bytes, self._stream())
            self._validate_total(self._host_out[0])
            self._cdf_key = key
        return self._cdf_dev

    def _device_rng_ok(self, n_uniform, n_normal):
        """Whether self.rng can be continued on the device for this many draws: a Generator over PCG64,
        a request large enough to pay for the launches, tuning_parameters['device_rng'] not False."""
        return self._device_rng_state(n_uniform, n_normal) is not None

    def _device_rng_state(self, n_uniform, n_normal):
        """(state dict, uint64[4]) of self.rng if it can be continued on the device for this many draws, else
        None (reading the state costs ~6 us: resample() hands it on to the pipelined path)."""
        if not self.tuning_parameters.get("device_rng", True) or n_uniform + n_normal < _devrng.MIN_DEVICE_DRAWS:
            return None
        return _devrng.pcg64_state(self.rng)

    def _device_stream(self, n_uniform, n_normal):
        """A device-side continuation of self.rng for this many draws, or None (then self.rng is
        called on the host, as the reference does)."""
        if not self._device_rng_ok(n_uniform, n_normal):
            return None
        return _devrng.DeviceStream(self._lib, self._device, self._stream(), self.rng, n_uniform, n_normal)

    @staticmethod
    def _validate_total(total):
        """numpy's validation of p in Generator.choice (ValueError in the reference), in numpy's order and
        words.  ``total`` is what the CDF kernels report: sum(p), or -inf when some entry is negative."""
        if total != total:
            raise ValueError("Probabilities contain NaN")
        if total == float("-inf"):
            raise ValueError("Probabilities are not non-negative")
        if abs(total - 1.0) > SQRT_EPS:
            raise ValueError("Probabilities do not sum to 1. See Notes section of docstring for more information.")

    def _check_pending_total(self):
        """After a stream synchronisation: the deferred validation of the weights a small draw
        used.  On failure the generator is put back where it was before the draw — numpy
        validates p before it consumes any uniforms."""
        if self._pending_total is None:
            return
        (state,), self._pending_total = self._pending_total, None
        try:
            self._lib.call("obe_host_word_wait", self._total_ptrs[0], self._stream())
            self._validate_total(float(self._total_pinned[0]))
        except ValueError:
            if state is not None:
                self.rng.bit_generator.state = state
            self._cdf_key = None
            raise

    def _draw_indices(self, n_draws, stream=None, defer_validation=False):
        """Device int64 indices of ``n_draws`` weighted draws: the uniforms
        Generator.choice would take from self.rng, CDF search on the device.

        ``defer_validation``: the caller synchronises the stream soon anyway and then calls
        ``_check_pending_total()`` — the draw itself then never waits for the device."""
        if self._weights.shape[0] != self.n_particles:
            raise ValueError("a and p must have same size")
        own = stream is None
        if own:
            stream = self._device_stream(n_draws, 0)
        if stream is None and n_draws <= 64:
            # (the index buffer of small draws is made once per size: last_draw_indices_device is "the
            # most recent draw", and every consumer reads it on the same stream before the next one)
            bufs = self.__dict__.setdefault("_small_idx", {})
            idx = bufs.get(n_draws)
            if idx is None:
                idx = bufs[n_draws] = torch.empty(n_draws, dtype=torch.int64, device=self._device)
            # small draw: uniforms as kernel arguments, CDF + search in one call, no synchronisation
            strict = bool(self.tuning_parameters.get("strict_cdf", False))
            key = (self._weights.version, strict)
            fresh = self._cdf_key == key
            w = self._weights.tensor()
            if self._cdf_dev.numel() != self.n_particles:
                self._cdf_dev = torch.empty(self.n_particles, dtype=torch.float64, device=self._device)
                fresh = False
            state = None
            if not fresh and hasattr(self.rng, "bit_generator"):
                state = self.rng.bit_generator.state
            u = np.atleast_1d(self.rng.random(n_draws))
            if not fresh:
                # (watched like every other result word: a store issued earlier is not guaranteed to reach the host before
                # one issued later, by the same kernel or the next — _check_pending_total waits for this one itself)
                self._lib.call("obe_host_word_arm", self._total_ptrs[0])
            self._lib.call("obe_draw_indices", _ptr(w), self.n_particles, 1 if strict else 0, 1 if fresh else 0,
                           _ptr(self._cdf_dev), _lib.host_ptr(u), n_draws, _ptr(idx),
                           None if fresh else self._total_ptrs[0], _ptr(self._ws), self._ws_bytes,
                           self._stream())
            if not fresh:
                self._cdf_key = key
                self._pending_total = (state,)
                if not defer_validation:
                    torch.cuda.current_stream(self._device).synchronize()
                    _lib.audit.synchronized()
                    self._check_pending_total()
            self.last_draw_indices_device = idx
            return idx
        idx = torch.empty(n_draws, dtype=torch.int64, device=self._device)
        cdf = self._cdf()
        if stream is not None:
            u_dev = stream.uniforms()
            if own:
                stream.finish_uniform_only()
        else:
            u_dev = torch.from_numpy(np.atleast_1d(self.rng.random(n_draws))).to(self._device)
        self._lib.call("obe_cdf_search", _ptr(cdf), self.n_particles, _ptr(u_dev), n_draws, _ptr(idx),
                       _ptr(self._ws), self._ws_bytes, self._stream())
        self.last_draw_indices_device = idx
        return idx

    def randdraw(self, n_draws=1):
        """``n_dims x n_draws`` weighted random draws (particlepdf.py:312-345)."""
        if n_draws == 0:          # rng.choice(size=0): nothing drawn, nothing consumed, an (n_dims, 0) array
            return np.empty((self.n_dims, 0))
        idx = self._draw_indices(n_draws)
        p = self._particles.tensor()
        out = torch.empty((self.n_dims, n_draws), dtype=torch.float64, device=self._device)
        self._lib.call("obe_gather_columns", _ptr(p), p.shape[1], self.n_dims, self.n_particles, _ptr(idx),
                       n_draws, _ptr(out), n_draws, self._stream())
        return out.cpu().numpy()

    @property
    def last_draw_indices(self):
        """Host copy of the particle indices of the most recent draw (diagnostic)."""
        t = self.last_draw_indices_device
        return None if t is None else t.cpu().numpy()

    def resample(self):
        """Multinomial resample + Gaussian nudge (+ contraction if ``scale``)
        (particlepdf.py:260-310).  RNG order as in the reference: N uniforms, then
        N x D standard normals."""
        n, d = self.n_particles, self.n_dims
        method = self.tuning_parameters.get("resample_method", "multinomial")
        if method == "multinomial":                       # the reference: N i.i.d. uniforms
            state = self._device_rng_state(n, n * d)
            # (the one-call pipeline exists for the widths its kernels are compiled for; a wider cloud takes the
            # step-by-step path, whose kernels tile over the rows: include/obe_hip.h, OBE_FAST_DIMS)
            if state is not None and self.tuning_parameters.get("pipelined_resample", True) and d <= _lib.OBE_FAST_DIMS:
                return self._resample_pipelined(state)
            rstream = self._device_stream(n, n * d)       # exact continuation of self.rng, or None
            idx = self._draw_indices(n, rstream)
        elif method == "systematic":                      # extension: ONE uniform, draws at (i + u0)/N
            cdf = self._cdf()
            u0 = float(self.rng.random())
            idx = torch.empty(n, dtype=torch.int64, device=self._device)
            self._lib.call("obe_systematic_indices", _ptr(cdf), n, u0, n, _ptr(idx), self._stream())
            rstream = self._device_stream(0, n * d)
        else:
            raise ValueError(f"unknown resample_method {method!r} (multinomial or systematic)")
        m = self._moments(True)                       # pre-resample weights (:290-291)
        factor, mean = self._nudge_factor(m)
        if rstream is not None:
            z_dev = rstream.normals()                 # (n*d,) row-major (n, d); advances self.rng
        else:
            z_dev = torch.from_numpy(self.rng.standard_normal((n, d))).to(self._device)
        self._resample_apply(idx, z_dev, factor, mean)

    def _nudge_factor(self, m, defer_check=False):
        """(F, mean) for the nudge of resample(): Generator.multivariate_normal(method='svd') draws
        x = z @ (u * sqrt(s)).T from the SVD of (1 - a^2) cov.  ``defer_check``: (F, mean, check) — numpy's
        validity test of the covariance only warns, so the pipelined resample runs it (``check()``) after
        the gather has been launched instead of before (25 of the 40-55 us this function takes)."""
        d = self.n_dims
        mean = m[2:2 + d].copy()
        cov = m[2 + 4 * d:2 + 4 * d + d * d].reshape((d, d))
        a = self.tuning_parameters["a_param"]
        newcov = (1 - a ** 2) * cov
        u, s, vh = np.linalg.svd(newcov)

        def check():
            # numpy's check_valid='warn': allclose(dot(v.T * s, v), cov, rtol = atol = 1e-8), spelled out
            # (np.allclose itself costs ~25 us of interpreter in every resample)
            back = np.dot(vh.T * s, vh)
            with np.errstate(invalid="ignore"):
                ok = bool(np.all(np.abs(back - newcov) <= 1e-8 + 1e-8 * np.abs(newcov))) \
                    and bool(np.all(np.isfinite(newcov)))
            if not ok:
                warnings.warn("covariance is not symmetric positive-semidefinite.", RuntimeWarning)

        factor = np.ascontiguousarray(u * np.sqrt(s))
        if defer_check:
            return factor, mean, check
        check()
        return factor, mean

    def _resample_mask_rows(self):
        """Hook: (int32 rows array, n) if a positivity constraint on those parameter rows is KNOWN to follow this
        resample (OptBayesExptNoiseParameter inside pdf_update()): the gather then zeroes those weights itself."""
        return None

    def _resample_apply(self, idx, z_dev, factor, mean, aos=None, new=None):
        """Gather + nudge.  ``aos``: the (N, D) copy of the old cloud that obe_resample_begin already made;
        ``new``: the (D, N) tensor for the new cloud if the caller has allocated it already."""
        n, d = self.n_particles, self.n_dims
        old = self._particles.tensor()
        if new is None:
            new = torch.empty((d, n), dtype=torch.float64, device=self._device)
        w = self._weights.tensor()
        mask = self._resample_mask_rows() if aos is not None else None
        self._masked_by_gather = None
        if mask is not None:
            partials = self.__dict__.get("_mask_partials")
            if partials is None:          # {sum w, count} per workgroup of the gather: the object's own, not the workspace
                partials = self._mask_partials = torch.empty(2 * 2048, dtype=torch.float64, device=self._device)
            self._lib.call("obe_resample_particles_aos_masked", _ptr(aos), d, n, _ptr(idx), _ptr(z_dev),
                           _lib.host_ptr(factor), _lib.host_ptr(mean), float(self.tuning_parameters["a_param"]),
                           1 if self.tuning_parameters["scale"] else 0, _ptr(new), n, _ptr(w), _lib.host_ptr(mask[0]),
                           mask[1], _ptr(partials), self._stream())
        elif aos is not None:
            self._lib.call("obe_resample_particles_aos", _ptr(aos), d, n, _ptr(idx), _ptr(z_dev),
                           _lib.host_ptr(factor), _lib.host_ptr(mean), float(self.tuning_parameters["a_param"]),
                           1 if self.tuning_parameters["scale"] else 0, _ptr(new), n, _ptr(w), self._stream())
        else:
            self._lib.call("obe_resample_particles", _ptr(old), old.shape[1], d, n, _ptr(idx), _ptr(z_dev),
                           _lib.host_ptr(factor), _lib.host_ptr(mean), float(self.tuning_parameters["a_param"]),
                           1 if self.tuning_parameters["scale"] else 0, _ptr(new), n, _ptr(w), _ptr(self._ws),
                           self._ws_bytes, self._stream())
        self._particles = Mirror(self._device, tensor=new)
        self._weights.mark_device_written()
        self.last_resample_indices_device = idx
        if mask is not None:      # (valid while nobody touches the cloud or the weights: enforce_parameter_constraints checks)
            self._masked_by_gather = (self._particles.version, self._weights.version)

    def _resample_buffers(self, n, d):
        """Scratch of the pipelined resample, made once per cloud shape: raw PCG64 values, uniforms,
        normals, the ziggurat workspace, page-locked landing zones and their addresses."""
        b = self.__dict__.get("_rs_bufs")
        if b is not None and b["shape"] == (n, d) and b["n_raw"] >= n + n * d + b["margin"]:
            return b
        margin = max((b or {}).get("margin", 0), (n * d) // 24 + 4096)       # ~2.2 % is consumed extra
        n_raw = n + n * d + margin
        dev = self._device
        zig_bytes = int(self._lib.cdll.obe_ziggurat_workspace_bytes(n_raw - n))
        mlen = self._lib.moments_len(d)
        pin_f, pin_i = _lib.pinned_array(mlen + 8), _lib.pinned_array(2, np.int64)
        b = self._rs_bufs = dict(
            shape=(n, d), margin=margin, n_raw=n_raw,
            uni=torch.empty(n, dtype=torch.float64, device=dev),
            normals=torch.empty(n * d, dtype=torch.float64, device=dev),
            zig_ws=torch.empty(zig_bytes // 8 + 1, dtype=torch.float64, device=dev), tables=_devrng._tables(dev),
            pin_f=pin_f, pin_i=pin_i, p_f=_lib.host_ptr(pin_f), p_i=_lib.host_ptr(pin_i),
            # the (N, D) copy of the pre-resample cloud the gather reads (made inside obe_resample_begin, while
            # the host factorises the covariance); small or one-parameter clouds gather from the (D, N) array
            aos=torch.empty(n * d, dtype=torch.float64, device=dev) if d >= 2 and n >= 65536 else None,
            # the drawn indices, two buffers used in turn: last_resample_indices_device of one resample stays
            # what it was through the next one
            idx=(torch.empty(n, dtype=torch.int64, device=dev), torch.empty(n, dtype=torch.int64, device=dev)),
            flip=0)
        return b

    # ---- the random numbers of the NEXT resample, enqueued ahead of it (round 6) ----------------------------------
    # resample() takes N uniforms and N x D normals from self.rng (particlepdf.py:272, 296-301): numbers that depend
    # on the generator state and the cloud's shape alone, and whose chain (92 us at 524 288 x 10) is what the gather
    # of a pipelined resample ends up waiting for.  pdf_update() therefore enqueues that chain when it STARTS — beside
    # the update's own latency-bound kernels and the host round trips — and a resample that finds the generator
    # exactly where the chain was started from uses its output; the numbers are kept through cycles that do not
    # resample, and thrown away (and the habit dropped after two misses) as soon as anything else has moved the
    # generator.  The same kernels on the same state: the same bits, the same generator bookkeeping.
    # tuning_parameters['randoms_ahead']: False (the default — built, bit-identical, measured and NOT a gain at the
    # BASELINE sizes: profiles/r06_randoms_ahead.txt), True, 'auto' (where nothing else draws from self.rng between
    # resamples: OptBayesExpt with the full sweep and opt_setting).
    def _randoms_ahead_enqueue(self):
        """Enqueue the random chain of the next resample from the generator's PRESENT state, unless one is already
        waiting for exactly that state.  Never raises: any refusal switches the feature off for this object."""
        n, d = self.n_particles, self.n_dims
        if d > _lib.OBE_FAST_DIMS or not self.tuning_parameters.get("pipelined_resample", True) \
                or self.tuning_parameters.get("resample_method", "multinomial") != "multinomial" \
                or self.__dict__.get("_ahead_misses", 0) >= 2:
            return False
        state = self._device_rng_state(n, n * d)
        if state is None:
            return False
        _, h_state = state
        pre = self.__dict__.get("_ahead")
        if pre is not None:
            if pre["shape"] == (n, d) and np.array_equal(pre["h_state"], h_state):
                return True                              # still good: the generator has not moved
            self._randoms_ahead_drop(miss=True)
            if self._ahead_misses >= 2:
                return False
        b = self._resample_buffers(n, d)
        try:
            self._lib.call("obe_resample_randoms_enqueue", _lib.host_ptr(h_state), n, d, b["n_raw"], _ptr(b["uni"]),
                           _ptr(b["tables"]), _ptr(b["normals"]), _ptr(b["zig_ws"]), b["zig_ws"].numel() * 8, b["p_i"],
                           self._stream())
        except _lib.ObeHipError as exc:
            if not exc.refused_before_launch:
                raise
            self._ahead_misses = 2                       # (no side streams: never again for this object)
            return False
        self._ahead = dict(shape=(n, d), h_state=h_state.copy(), bufs=b, stream=self._stream())
        # (the chain runs on a stream torch knows nothing about: its buffers must not go back to the allocator — with
        # this object, say — before it has ended; the module keeps them until their result words have arrived)
        _AHEAD_IN_FLIGHT[:] = [e for e in _AHEAD_IN_FLIGHT if _still_armed(e["pin_i"])] + [b]
        return True

    def _randoms_ahead_drop(self, miss=False):
        """Forget the chain enqueued ahead (the generator moved, the cloud changed shape): its kernels are waited
        for — they deliver {consumed, found} when they end — before anything arms those words or reuses the buffers."""
        pre = self.__dict__.get("_ahead")
        if pre is None:
            return
        self._ahead = None
        self._lib.call("obe_host_words_wait", pre["bufs"]["p_i"], 2, pre["stream"])
        if miss:
            self._ahead_misses = self.__dict__.get("_ahead_misses", 0) + 1

    def _resample_pipelined(self, state=None):
        """resample() with the caller's generator continued on the device.  ONE library call
        (obe_resample_begin) enqueues CDF, uniforms, search, covariance and the ziggurat normals back to
        back; the host waits — by watching the page-locked words of the covariance block — for
        the covariance, factorises it while the normals are still being generated, launches the gather
        + nudge and then reads the generator bookkeeping the same way (no stream synchronisation, so
        the gather runs on while the caller goes on).  Same kernels, same numbers and the same
        generator state as the step-by-step path."""
        n, d = self.n_particles, self.n_dims
        if self._weights.shape[0] != n:
            raise ValueError("a and p must have same size")
        st, h_state = state if state is not None else _devrng.pcg64_state(self._rng)
        strict = bool(self.tuning_parameters.get("strict_cdf", False))
        key = (self._weights.version, strict)
        p, w = self._pw_tensors()
        if self._cdf_dev.numel() != n:
            self._cdf_dev = torch.empty(n, dtype=torch.float64, device=self._device)
            self._cdf_key = None
        b = self._resample_buffers(n, d)
        # the chain enqueued ahead, if it started from exactly this generator state (and wrote these buffers)
        pre = self.__dict__.get("_ahead")
        ahead = pre is not None and pre["bufs"] is b and pre["shape"] == (n, d) and np.array_equal(pre["h_state"], h_state) \
            and pre["stream"].value == self._stream().value
        if pre is not None and not ahead:
            self._randoms_ahead_drop(miss=True)
        elif ahead:
            self._ahead, self._ahead_misses = None, 0
            self._ahead_hits = self.__dict__.get("_ahead_hits", 0) + 1        # (diagnostic)
        mlen = self._lib.moments_len(d)
        b["flip"] ^= 1
        idx = b["idx"][b["flip"]]
        mkey = (self._particles.version, self._weights.version)
        have_first = self._mom_host_key is not None and self._mom_host_key[:2] == mkey \
            and self._mom_dev_key is not None and self._mom_dev_key[:2] == mkey
        self._await_host_moments()
        stream = self._stream()
        self._lib.call("obe_resample_begin", _ptr(p), p.shape[1], d, n, _ptr(w), None if ahead else _lib.host_ptr(h_state),
                       1 if strict else 0, 1 if self._cdf_key == key else 0, 1 if have_first else 0,
                       b["n_raw"], _ptr(self._cdf_dev), _ptr(b["uni"]), _ptr(idx), _ptr(b["tables"]),
                       _ptr(b["normals"]), _ptr(b["zig_ws"]), b["zig_ws"].numel() * 8, _ptr(self._moments_dev),
                       b["p_f"], b["p_i"], None if b["aos"] is None else _ptr(b["aos"]), _ptr(self._ws),
                       self._ws_bytes, stream)
        pin_f = b["pin_f"]
        first = 2 + 4 * d                              # (a covariance-only pass delivers only the covariance)
        lo = first if have_first else 0
        # (the new cloud's storage, while the covariance is still on its way: off the host's critical path)
        new_cloud = torch.empty((d, n), dtype=torch.float64, device=self._device)
        # every word of the block is watched (the call armed them): the covariance is there, the normals still run
        self._lib.call("obe_host_words_wait", _P(pin_f.ctypes.data + 8 * (1 + lo)), mlen - lo, stream)
        self._lib.call("obe_host_words_wait", b["p_f"], 1, stream)     # sum(w), from an earlier kernel
        self._validate_total(float(pin_f[0]))         # (raises before any generator state has moved)
        self._cdf_key = key
        self._moments_host[lo:mlen] = pin_f[1 + lo:1 + mlen]
        self._mom_host_key = self._mom_dev_key = mkey + (True,)
        factor, mean, check_covariance = self._nudge_factor(self._moments_host, defer_check=True)
        self.last_draw_indices_device = idx
        before = self._particles
        self._resample_apply(idx, b["normals"], factor, mean, aos=b["aos"], new=new_cloud)
        check_covariance()
        self._lib.call("obe_host_words_wait", b["p_i"], 2, stream)     # {raw consumed, normals found}
        consumed, found = int(b["pin_i"][0]), int(b["pin_i"][1])
        if self._lib.cdll.obe_ziggurat_check(consumed, found, n * d, b["n_raw"] - n, 0) != 0:
            # unlucky stream (never seen): the raw buffer was too short for N D normals — draw them
            # again from a longer one and repeat the gather from the old cloud
            b["margin"] = 2 * b["margin"] + 65536
            rstream = _devrng.DeviceStream(self._lib, self._device, stream, self._rng, n, n * d)
            rstream.margin = b["margin"]
            rstream._generate()
            z_dev = rstream.normals()                  # (advances the generator past uniforms + normals)
            self._particles = before
            self._resample_apply(idx, z_dev, factor, mean, aos=b["aos"])
        else:
            _devrng.advance(self._rng, st, n + consumed)

    @staticmethod
    def _normalized_product(weight_array, likelihood_array):
        """Kept for API compatibility (particlepdf.py:347-360); the product path is
        ``bayesian_update``."""
        raise NotImplementedError("use ParticlePDF.bayesian_update(); the normalised product "
                                  "runs on the device")
