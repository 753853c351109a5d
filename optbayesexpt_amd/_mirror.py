"""Host-visible views of device-resident state.

The reference exposes ``particles`` / ``particle_weights`` / ``parameters`` as plain
ndarrays that user code reads, slices and *writes in place*
(``self.particle_weights[bad] = 0``: obe_noiseparam.py:71, tests/test_zinference.py:40,
demos/lockin/lockin_of_coil.py:128) or rebinds (tests/test_particlepdf.py:128).  Here the
device tensor is the working copy and the ndarray is a lazily synchronised mirror:

* a kernel wrote the tensor      -> the host copy is stale, refreshed on next access;
* user code wrote the host array -> ``TrackedArray`` notices and the tensor is
  re-uploaded before the next kernel reads it.

No arithmetic happens here; it is bookkeeping plus hipMemcpy (through torch).
"""
import itertools

import numpy as np
import torch

# process-wide stamps: a (particles, weights) version pair never repeats, even across
# Mirror objects, so it can key cached reductions
_STAMP = itertools.count(1)


class TrackedArray(np.ndarray):
    """Read-only-flagged ndarray view of a mirror's host buffer that lets the two in-place idioms
    of the reference's callers through — item assignment (``w[bad] = 0``) and ufuncs with ``out=``
    (``w *= 2``) — and reports them to the owning mirror.  Views (slices, rows) keep the link;
    copies and arithmetic results are plain writable arrays.

    Because the view is flagged read-only, every *other* way of writing into it (``np.copyto``,
    ``np.put``/``putmask``, ``ndarray.fill``/``sort``/``partition``, ``.flat[...] =``,
    ``np.nan_to_num(copy=False)``) raises numpy's "destination is read-only" ValueError instead of
    silently leaving the device copy stale.  A view of a buffer the mirror has since replaced (the
    cloud was updated on the device after the view was taken) is a snapshot, as an array kept across
    ``pdf_update`` is in the reference: writes go to the snapshot only."""

    _obe_owner = None

    def __array_finalize__(self, obj):
        owner = getattr(obj, "_obe_owner", None)
        # only views of the mirrored buffer stay linked
        self._obe_owner = owner if (owner is not None and self.base is not None) else None

    def _touch(self):
        owner = self._obe_owner
        if owner is not None and owner._host is not None and np.may_share_memory(self, owner._host):
            owner.mark_host_written()

    def _unlocked(self):
        """A plain writable ndarray over the same memory (allowed: the mirror's buffer, the
        ultimate base of every view, is itself writable)."""
        alias = self.view(np.ndarray)
        if self._obe_owner is not None:
            alias.flags.writeable = True
        return alias

    def __setitem__(self, key, value):
        self._unlocked()[key] = value
        self._touch()

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        plain_in = tuple(np.asarray(x) if isinstance(x, TrackedArray) else x for x in inputs)
        touched = []
        if out is not None:
            touched = [o for o in out if isinstance(o, TrackedArray)]
            kwargs["out"] = tuple(o._unlocked() if isinstance(o, TrackedArray) else o for o in out)
        result = getattr(ufunc, method)(*plain_in, **kwargs)
        for o in touched:
            o._touch()
        if out is not None and len(out) == 1 and touched:
            return out[0]
        return result

    def __reduce__(self):   # pickles as a plain array
        return np.asarray(self).__reduce__()


class Mirror:
    """A float64 array living on the GPU with a lazily synchronised host copy."""

    def __init__(self, device, host=None, tensor=None):
        self.device = device
        self.version = next(_STAMP)
        self._host = None
        self._tensor = None
        if host is not None:
            self.set_host(host)
        elif tensor is not None:
            self.set_tensor(tensor)
        else:
            raise ValueError("Mirror needs a host array or a device tensor")

    # -- host side -----------------------------------------------------------
    def set_host(self, array):
        arr = np.array(array, dtype=np.float64, order="C", copy=True)
        self.shape = arr.shape
        self._host = arr
        self._host_valid = True
        self._dev_valid = False
        self.host_born = True          # this version's values were written by host code (not by a kernel)
        self.version = next(_STAMP)

    def host(self):
        if not self._host_valid:
            self._host = self._tensor.cpu().numpy()      # D2H, synchronises
            self._host_valid = True
        view = self._host.view(TrackedArray)
        view._obe_owner = self
        view.flags.writeable = False        # writes go through TrackedArray's tracked paths or raise
        return view

    def mark_host_written(self):
        self._dev_valid = False
        self.host_born = True
        self.version = next(_STAMP)

    # -- device side ---------------------------------------------------------
    def set_tensor(self, tensor):
        assert tensor.dtype == torch.float64 and tensor.is_contiguous()
        self.shape = tuple(tensor.shape)
        self._tensor = tensor
        self._dev_valid = True
        self._host_valid = False
        self.host_born = False
        self.version = next(_STAMP)

    def tensor(self):
        """Device tensor, uploading the host copy first if user code changed it."""
        if not self._dev_valid:
            src = torch.from_numpy(np.ascontiguousarray(self._host))
            if self._tensor is None or tuple(self._tensor.shape) != tuple(src.shape):
                self._tensor = torch.empty(src.shape, dtype=torch.float64, device=self.device)
            self._tensor.copy_(src)                        # H2D
            self._dev_valid = True
        return self._tensor

    def mark_device_written(self):
        """A kernel modified the tensor in place."""
        self._dev_valid = True
        self._host_valid = False
        self.host_born = False
        self.version = next(_STAMP)
