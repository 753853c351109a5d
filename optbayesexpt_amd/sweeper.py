"""OptBayesExptSweeper — instruments that sweep a setting.

Mirrors the reference's demos/sweeper/obe_sweeper.py:9-229 (a subclass of
OptBayesExptNoiseParameter that ships with the demos, not with the package): the
"settings" offered to the experiment are (start, stop) index pairs on the first setting
axis, a measurement record is a whole sweep ``((x_values,), y_values)``, and the utility
of a sweep is the point utility integrated between its ends over the sweep's cost.

Device side: the point utility stays where K1 left it; ``obe_cumsum`` forms its running
sum, ``obe_interval_utility`` differences it at the (start, stop) rows (O(N_s^2 / 18) of
them: 15 M at 16384 settings) and ``obe_argmax`` / the CDF search pick the pair.  The
per-point updates of a sweep run through the K2 kernels one after another exactly as the
reference does (the resample test sits between the points).

The reference subclass itself also runs unchanged on top of this package's
OptBayesExptNoiseParameter (it only uses ``self.utility()``, ``super().pdf_update`` and
NumPy); this class is the same thing with the composition on the device.
"""
import numpy as np
import torch

from . import _lib
from .obe_base import OptBayesExpt, _LazyState, _overridden
from .obe_noiseparam import OptBayesExptNoiseParameter
from .particlepdf import ParticlePDF, _ptr

#: module-level generator for good_setting / random_setting, like the reference module's
#: own ``rng`` (obe_sweeper.py:3-6)
rng = np.random.default_rng()


class OptBayesExptSweeper(OptBayesExptNoiseParameter):
    """Same constructor and attributes as obe_sweeper.py:72-84:
    ``sweep_settings``, ``start_stop_subsample`` (3), ``start_stop_indices`` (n_pairs, 2),
    ``start_stop_choice_indices``, ``start_stop_values``, ``cost_of_new_sweep`` (5.0)."""

    def __init__(self, model_function, setting_values, parameter_samples, constants,
                 noise_parameter_index, **kwargs):
        OptBayesExptNoiseParameter.__init__(self, model_function, setting_values, parameter_samples,
                                            constants, noise_parameter_index, **kwargs)
        self.sweep_settings = setting_values[0]
        self.start_stop_subsample = 3
        self.start_stop_indices = self._generate_start_stop_indices()
        self.start_stop_choice_indices = np.arange(len(self.start_stop_indices), dtype=int)
        self.start_stop_values = self.sweep_settings[self.start_stop_indices]
        self.cost_of_new_sweep = 5.
        self._pairs_key = None
        self._pairs_dev = self._pair_ws = None
        self._cum_dev = torch.empty(self._n_settings, dtype=torch.float64, device=self._device)
        self._sweep_utility_dev = None

    # ------------------------------------------------------------------ inference
    def pdf_update(self, measurement_record):
        """One Bayesian update per point of the sweep (obe_sweeper.py:86-100); the record is
        ``((x_values,), y_values)``.

        With a device model and none of the per-point hooks overridden the points are enqueued
        in batches (``obe_bayes_update_sweep``): the resample test between the points runs on
        the device, and the host synchronises once per batch instead of once per point.  A point
        that calls for a resample ends its batch; the resample (and the constraint hook) run
        on the host path exactly as in the one-by-one loop, then the rest of the sweep follows."""
        (setting_values,), result_values = measurement_record
        points = list(zip(setting_values, result_values))
        self.last_sweep_batches = []            # (points submitted, points applied) per device batch; empty: point by point
        if not points:
            return None
        batch = self._sweep_batch_inputs(points)
        if batch is None:
            out = None
            for setting, result in points:
                out = OptBayesExptNoiseParameter.pdf_update(self, ((setting,), result))
            return out
        xs, ys, n_lik = batch
        par, w = self._parameters.tensor(), self._weights.tensor()
        if par.shape[1] != w.shape[0]:
            raise ValueError("parameters and particle_weights have different lengths")
        pos, chunk, out = 0, self.SWEEP_BATCH_MIN, self._host_out
        while pos < len(points):
            k = min(chunk, len(points) - pos)
            par, w = self._parameters.tensor(), self._weights.tensor()
            self._unfused_update(self._mlib, "obe_bayes_update_sweep", self._model_struct, _ptr(par), par.shape[1], self.n_particles,
                            _ptr(w), _lib.host_ptr(xs[pos:]), _lib.host_ptr(ys[pos:]), None,
                            _lib.host_ptr(self._noise_rows), n_lik, self._choke_value(), k,
                            1 if self.tuning_parameters["auto_resample"] else 0,
                            float(self.tuning_parameters["resample_threshold"]),
                            _ptr(self._ws), self._ws_bytes, _lib.host_ptr(out), self._stream())
            applied = int(out[3])
            self.last_sweep_batches.append((k, applied))
            self._after_weight_update(out[1])       # the same test on the host: resamples if it is due
            self._parameters = self._particles
            if self.just_resampled:
                self.enforce_parameter_constraints()
            pos += applied
            chunk = min(self.SWEEP_BATCH_MAX, 2 * chunk) if applied == k else self.SWEEP_BATCH_MIN
        return _LazyState(self)

    #: points per device batch: doubled after every batch that ran to its end, reset by a resample
    SWEEP_BATCH_MIN, SWEEP_BATCH_MAX = 8, 64

    def _sweep_batch_inputs(self, points):
        """(settings (M, OBE_MAX_SETDIMS), y (M, OBE_MAX_CHANNELS), n_lik) for the batched
        update, or None when the sweep has to go point by point (host-callable model, or a
        subclass overriding one of the hooks that the per-point path calls)."""
        if self._device_model is None or len(points) < 2 \
                or _overridden(self, "eval_over_all_parameters", OptBayesExpt) or self._likelihood_overridden() \
                or _overridden(self, "bayesian_update", ParticlePDF) or _overridden(self, "resample_test", ParticlePDF) \
                or _overridden(self, "_likelihood_inputs", OptBayesExptNoiseParameter):
            return None
        xs = np.zeros((len(points), _lib.OBE_MAX_SETDIMS))
        ys = np.zeros((len(points), _lib.OBE_MAX_CHANNELS))
        n_lik = None
        for i, (setting, result) in enumerate(points):
            xs[i] = self._setting_array((setting,))
            n, ys[i], _ = self._record_channels(result, None)
            if n_lik is None:
                n_lik = n
            elif n != n_lik:
                return None
        return xs, ys, n_lik

    # -------------------------------------------------------------------- utility
    def cost_estimate(self):
        # point costs are uniform along a sweep (obe_sweeper.py:102-104)
        return 1.0

    def sweep_cost_estimate(self):
        """Sweep length + the cost of setting a sweep up (obe_sweeper.py:106-120)."""
        return self.start_stop_indices[:, 1] - self.start_stop_indices[:, 0] + self.cost_of_new_sweep

    @staticmethod
    def _fingerprint(a):
        """Cheap change detector for a large host array: identity, shape and a strided sample
        (replace the array, as the constructor does, rather than editing single rows of a
        multi-million-row table in place)."""
        flat = a.reshape(-1)
        step = max(1, flat.size // 4096)
        return (id(a), a.shape, flat[::step].tobytes(), flat[-2:].tobytes())

    def _pair_tensors(self):
        """(start, stop) rows on the device — re-uploaded only when the host array changed —
        and the sweep costs: None while sweep_cost_estimate() is the class's own (the kernel
        then forms stop - start + cost_of_new_sweep itself), else the overriding method's array."""
        pairs = self.start_stop_indices
        key = self._fingerprint(pairs)
        if key != self._pairs_key:
            p = np.ascontiguousarray(pairs, dtype=np.int64)
            if p.ndim != 2 or p.shape[1] != 2 or len(p) == 0:
                raise ValueError("start_stop_indices must be an (n_pairs, 2) index array")
            self._pairs_dev = torch.from_numpy(p).to(self._device)
            n = self._lib.workspace_bytes(max(len(p), self._n_settings), 1, 1, 1)
            self._pair_ws = torch.empty(n // 8 + 1, dtype=torch.float64, device=self._device)
            self._sweep_utility_dev = torch.empty(len(p), dtype=torch.float64, device=self._device)
            self._pairs_key = key
        if not _overridden(self, "sweep_cost_estimate", OptBayesExptSweeper):
            return self._pairs_dev, None
        n_pairs = self._pairs_dev.shape[0]
        cost = np.array(np.broadcast_to(np.asarray(self.sweep_cost_estimate(), dtype=np.float64), (n_pairs,)))
        return self._pairs_dev, torch.from_numpy(cost).to(self._device)

    def _sweep_utility_device(self):
        """The utility of every (start, stop) pair, left on the device."""
        if self._utility_fusable():
            self._sweep_device(False)                    # K1: point utility on the device
            u = self._utility_dev
            if self._shard is not None:                  # every rank needs the whole running sum
                full = self._gather_settings(self._utility_dev.reshape(1, -1))[0]
                u = torch.from_numpy(np.ascontiguousarray(full)).to(self._device)
        else:                                            # overridden / y-space point utility
            u = torch.from_numpy(np.ascontiguousarray(self.utility(), dtype=np.float64).reshape(-1)) \
                .to(self._device)
        if u.numel() != self._n_settings:
            raise ValueError("the point utility must have one value per setting")
        pairs, cost = self._pair_tensors()
        ws, wsb = self._pair_ws, self._pair_ws.numel() * 8
        strict = 1 if self.tuning_parameters.get("strict_cdf", False) else 0
        self._lib.call("obe_cumsum", _ptr(u), self._n_settings, strict, _ptr(self._cum_dev), _ptr(ws), wsb,
                       self._stream())
        self._lib.call("obe_interval_utility", _ptr(self._cum_dev), self._n_settings, _ptr(pairs), pairs.shape[0],
                       None if cost is None else _ptr(cost), float(self.cost_of_new_sweep),
                       _ptr(self._sweep_utility_dev), self._stream())
        return self._sweep_utility_dev

    def sweep_utility(self):
        """Utility of every (start, stop) pair, (n_pairs,) (obe_sweeper.py:122-149)."""
        host = self._sweep_utility_device().cpu().numpy()
        self._check_pending_total()      # the copy synchronised: a lazy draws-mode sweep's check of sum(w)
        return host

    # ------------------------------------------------------------------ selection
    def opt_setting(self):
        """The (start, stop) index pair with maximum utility (obe_sweeper.py:151-167)."""
        u = self._sweep_utility_device()
        best = np.zeros(1)
        best_idx = np.zeros(1, dtype=np.int64)
        self._lib.call("obe_argmax", _ptr(u), u.numel(), _lib.host_ptr(best), _lib.host_ptr(best_idx),
                       _ptr(self._pair_ws), self._pair_ws.numel() * 8, self._stream())
        self._check_pending_total()      # (obe_argmax waited for its host results)
        index = int(best_idx[0])
        self.last_setting_index = index
        return self.start_stop_indices[index]

    def good_setting(self):
        """A (start, stop) pair drawn with probability ~ utility**pickiness
        (obe_sweeper.py:169-193); consumes one uniform of this module's ``rng`` (rank 0's value on a
        sharded object, see ``_rank0_values``)."""
        u = self._sweep_utility_device()
        n = u.numel()
        ws, wsb = self._pair_ws, self._pair_ws.numel() * 8
        prob = torch.empty(n, dtype=torch.float64, device=self._device)
        cdf = torch.empty(n, dtype=torch.float64, device=self._device)
        self._lib.call("obe_power_normalize", _ptr(u), n, float(self.pickiness), _ptr(prob), _ptr(ws), wsb,
                       self._stream())
        uni = self._rank0_values(rng.random())
        idx = torch.empty(1, dtype=torch.int64, device=self._device)
        self._lib.call("obe_draw_indices", _ptr(prob), n, 0, 0, _ptr(cdf), _lib.host_ptr(uni), 1, _ptr(idx), None,
                       _ptr(ws), wsb, self._stream())
        index = int(idx.cpu()[0])
        self._check_pending_total()      # (the copy synchronised)
        self.last_setting_index = index
        return self.start_stop_indices[index]

    def random_setting(self):
        """A uniformly random (start, stop) pair (obe_sweeper.py:195-205)."""
        index = int(self._rank0_values(rng.choice(self.start_stop_choice_indices))[0])
        self.last_setting_index = index
        return self.start_stop_indices[index]

    def _generate_start_stop_indices(self):
        """All [start, stop] pairs with stop > start on the sub-sampled index grid
        0, k, 2k, ... plus the last index, start-major (obe_sweeper.py:207-229)."""
        raw_length = len(self.sweep_settings)
        grid = np.arange(0, raw_length, self.start_stop_subsample)
        if grid[-1] != raw_length - 1:
            grid = np.append(grid, raw_length - 1)
        i, j = np.triu_indices(len(grid), 1)
        return np.stack([grid[i], grid[j]], axis=1)

