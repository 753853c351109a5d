"""OBE_CHECK_DELIVERY=1 — an audit of how results reach the host (debug mode, host side only).

Kernels of libobe_hip deliver their few result scalars by writing page-locked host memory themselves, and the
host waits by WATCHING those words (csrc/obe_common.h: arm_host_words / wait_host_words) instead of
synchronising the stream.  That protocol is only correct if every word that is read was waited for, and three
races of rounds 4-5 were exactly violations of it (tests/test_gpu_delivery_audit.py names them).  With the mode
on, the protocol is checked by construction instead of by soak runs:

* every page-locked landing zone (``_lib.pinned_array``) is registered, one state per 8-byte word;
* a call that will deliver into a zone without waiting (the enqueue forms, ``obe_host_word(s)_arm``) marks the
  words it delivers ARMED — the table below restates, per entry point, which host words a call arms
  (include/obe_hip.h) —, a wait marks the words it covered DELIVERED (and checks that none of them still holds
  the armed bit pattern), a device synchronisation everything;
* a Python read of an ARMED word raises ``DeliveryError`` (the arrays handed out are an ndarray subclass whose
  ``__getitem__`` looks the touched words up);
* a zone that is handed back to the allocator — or re-registered at the same address — while words of it are
  ARMED is a violation too (recorded, raised by the next library call: finalisers cannot raise).

Off (the default) every hook is a no-op method of ``_NoAudit``.
"""
import bisect
import os

import numpy as np

try:
    _byte_bounds = np.lib.array_utils.byte_bounds          # numpy >= 2
except AttributeError:                                      # pragma: no cover
    _byte_bounds = np.byte_bounds

SENTINEL = 0x7ff8c0dec0dec0de          # csrc/obe_common.h: kHostSentinel
OBE_SWEEP_SPECULATIVE, OBE_SWEEP_NOWAIT = 8, 16


class DeliveryError(RuntimeError):
    """A host result word was read, or its memory released, before it had been waited for."""


class _NoAudit:
    on = False

    def wrap(self, a):
        return a

    def zone_created(self, addr, nbytes, keeper):
        pass

    def zone_released(self, keeper):
        pass

    def zone_freed(self, keeper, drained):
        pass

    def after_call(self, name, args):
        pass

    def synchronized(self):
        pass

    def raw(self, a):
        return a


class CheckedArray(np.ndarray):
    """A view of a landing zone whose element reads are checked against the zone's word states."""

    def __getitem__(self, key):
        res = np.ndarray.__getitem__(self, key)
        audit.check_read(self, key, res)
        return res


def _addr(x):
    if x is None:
        return None
    v = getattr(x, "value", x)
    return None if v is None else int(v)


def _int(x):
    return int(getattr(x, "value", x))


class _Zone:
    __slots__ = ("addr", "nbytes", "armed", "released")

    def __init__(self, addr, nbytes):
        self.addr, self.nbytes = addr, nbytes
        self.armed = np.zeros((nbytes + 7) // 8, dtype=bool)
        self.released = False


class _Audit(_NoAudit):
    on = True

    def __init__(self):
        self.starts = []          # sorted zone start addresses
        self.zones = {}           # start -> _Zone
        self.violations = []
        self.counts = dict(zones=0, armed=0, waited=0, reads=0, syncs=0)

    # ---- zones ---------------------------------------------------------------------------------------
    def wrap(self, a):
        return a.view(CheckedArray)

    def raw(self, a):
        """The plain ndarray behind a checked view: for code that POLLS armed words on purpose."""
        return np.asarray(a).view(np.ndarray)

    def zone_created(self, addr, nbytes, keeper):
        old = self.zones.get(addr)
        if old is not None and old.armed.any():
            self.violations.append(f"landing zone at {addr:#x} handed out again while {int(old.armed.sum())} word(s) of "
                                   "its previous owner were still armed (a kernel of a dead object may write into it)")
        if old is None:
            bisect.insort(self.starts, addr)
        self.zones[addr] = _Zone(addr, nbytes)
        self.counts["zones"] += 1

    def zone_released(self, keeper):
        z = self.zones.get(keeper.data_ptr())
        if z is not None:
            z.released = True

    def zone_freed(self, keeper, drained):
        addr = keeper.data_ptr()
        z = self.zones.get(addr)
        if z is None:
            return
        if not drained and z.armed.any():
            self.violations.append(f"landing zone at {addr:#x} freed with {int(z.armed.sum())} armed word(s) and no "
                                   "device synchronisation in between")
            return                       # (stays registered: a re-use of the address is then reported as well)
        del self.zones[addr]
        self.starts.pop(bisect.bisect_left(self.starts, addr))

    def _find(self, addr):
        i = bisect.bisect_right(self.starts, addr) - 1
        if i < 0:
            return None
        z = self.zones[self.starts[i]]
        return z if addr < z.addr + z.nbytes else None

    def _mark(self, addr, n_words, armed):
        if addr is None or n_words <= 0:
            return
        z = self._find(addr)
        if z is None:
            return                       # (not one of this package's landing zones: a test's own buffer)
        w0 = (addr - z.addr) // 8
        z.armed[w0:w0 + n_words] = armed
        self.counts["armed" if armed else "waited"] += 1

    # ---- library calls -------------------------------------------------------------------------------
    def after_call(self, name, args):
        if self.violations:
            v, self.violations = self.violations, []
            raise DeliveryError("; ".join(v))
        rule = _RULES.get(name)
        if rule is not None:
            rule(self, args)

    def waited(self, addr, n_words):
        if addr is None:
            return
        words = np.ctypeslib.as_array((np.ctypeslib.ctypes.c_uint64 * n_words).from_address(addr))
        if np.any(words == SENTINEL):
            raise DeliveryError(f"obe_host_words_wait returned with {int(np.sum(words == SENTINEL))} of {n_words} "
                                f"word(s) at {addr:#x} still armed")
        self._mark(addr, n_words, False)

    def synchronized(self):
        for z in self.zones.values():
            z.armed[:] = False
        self.counts["syncs"] += 1

    # ---- reads ---------------------------------------------------------------------------------------
    def check_read(self, arr, key, res):
        self.counts["reads"] += 1
        if isinstance(res, np.ndarray):
            if res.size == 0:
                return
            lo, hi = _byte_bounds(res)
        elif arr.ndim == 1 and isinstance(key, (int, np.integer)):
            lo = arr.ctypes.data + (int(key) % arr.shape[0]) * arr.strides[0]
            hi = lo + arr.itemsize
        else:
            lo, hi = _byte_bounds(arr)
        z = self._find(lo)
        if z is None:
            return
        w0, w1 = (lo - z.addr) // 8, (hi - z.addr + 7) // 8
        if z.armed[w0:w1].any():
            bad = (np.nonzero(z.armed[w0:w1])[0] + w0).tolist()
            raise DeliveryError(f"read of word(s) {bad} of the landing zone at {z.addr:#x} that were armed and have not "
                                "been waited for since (obe_host_words_wait on exactly those words, or a device "
                                "synchronisation)")


# ---- which host words a call arms / delivers (include/obe_hip.h) -----------------------------------------
def _rule_arm(a, args):
    a._mark(_addr(args[0]), _int(args[1]) if len(args) > 1 else 1, True)


def _rule_arm1(a, args):
    a._mark(_addr(args[0]), 1, True)


def _rule_wait(a, args):
    a.waited(_addr(args[0]), _int(args[1]))


def _rule_wait1(a, args):
    a.waited(_addr(args[0]), 1)


def _rule_update_enqueue(a, args):
    d = int(args[0].n_params)
    a._mark(_addr(args[14]), 5 + 4 * d, True)


def _rule_sweep(a, args):
    nowait = _int(args[11]) & (OBE_SWEEP_SPECULATIVE | OBE_SWEEP_NOWAIT)
    for k in (18, 19, 20):
        a._mark(_addr(args[k]), 1, bool(nowait))         # (the synchronous form has waited for each of them itself)


def _rule_resample_begin(a, args):
    d = _int(args[2])
    mlen = 2 + 4 * d + d * d
    lo = 2 + 4 * d if _int(args[8]) else 0
    f64 = _addr(args[18])
    if not _int(args[7]):
        a._mark(f64, 1, True)                            # sum(w) of a CDF made by this call
    else:
        a._mark(f64, 1, False)                           # (written by the host: 1.0)
    a._mark(f64 + 8 * (1 + lo), mlen - lo, True)
    if _addr(args[5]) is not None:                       # (NULL state: the randoms were enqueued ahead, h_i64 armed then)
        a._mark(_addr(args[19]), 2, True)


def _rule_draw_indices(a, args):
    if not _int(args[3]) and _addr(args[8]) is not None:
        a._mark(_addr(args[8]), 1, True)                 # sum(p): delivered by the call's kernels, never waited for by it
    a._mark(_addr(args[7]), _int(args[6]), True)         # the indices, when d_idx is the device view of a landing zone


def _rule_mask_moments(h_mom, h_changed, n_dims):
    def rule(a, args):
        d = _int(args[n_dims])
        a._mark(_addr(args[h_mom]), 2 + 4 * d, True)
        a._mark(_addr(args[h_changed]), 1, True)
    return rule


def _delivered(index, words):
    def rule(a, args):
        a._mark(_addr(args[index]), words(args) if callable(words) else words, False)
    return rule


_RULES = {
    "obe_host_word_arm": _rule_arm1,
    "obe_host_words_arm": _rule_arm,
    "obe_host_word_wait": _rule_wait1,
    "obe_host_words_wait": _rule_wait,
    "obe_bayes_update_model_moments_enqueue": _rule_update_enqueue,
    "obe_sweep_utility": _rule_sweep,
    "obe_resample_begin": _rule_resample_begin,
    "obe_draw_indices": _rule_draw_indices,
    "obe_resample_randoms_enqueue": lambda a, args: a._mark(_addr(args[9]), 2, True),
    "obe_mask_nonpositive_moments": _rule_mask_moments(8, 9, 2),
    "obe_mask_renorm_moments": _rule_mask_moments(7, 8, 2),
    # synchronous forms: they wait for their own words before they return
    "obe_bayes_update_model": _delivered(13, 2),
    "obe_bayes_update_model_moments": _delivered(14, lambda args: 4 + 4 * int(args[0].n_params)),
    "obe_bayes_update_lik": _delivered(5, 2),
    "obe_bayes_update_y": _delivered(14, 2),
    "obe_mask_nonpositive": _delivered(6, 1),
    "obe_weight_sums": _delivered(4, 2),
    "obe_weight_cdf": _delivered(4, 1),
    "obe_utility_argmax": lambda a, args: (a._mark(_addr(args[8]), 1, False), a._mark(_addr(args[9]), 1, False)),
    "obe_argmax": lambda a, args: (a._mark(_addr(args[2]), 1, False), a._mark(_addr(args[3]), 1, False)),
}


# The argument positions the rules above rely on, by the parameter names of include/obe_hip.h
# (tests/test_capi_symbols.py::test_audit_rules_address_the_parameters_they_name holds the two together: a changed
# signature fails there instead of silently auditing the wrong argument).
RULE_PARAMETERS = {
    "obe_host_word_arm": {0: "h_pinned_word"},
    "obe_host_words_arm": {0: "h_pinned_words", 1: "n_words"},
    "obe_host_word_wait": {0: "h_pinned_word"},
    "obe_host_words_wait": {0: "h_pinned_words", 1: "n_words"},
    "obe_bayes_update_model_moments_enqueue": {0: "m", 14: "h_pinned_out"},
    "obe_sweep_utility": {11: "shifted", 18: "h_best", 19: "h_best_idx", 20: "h_kappa"},
    "obe_resample_begin": {2: "n_dims", 5: "h_pcg_state4", 7: "cdf_is_fresh", 8: "have_first_moments", 18: "h_f64",
                           19: "h_i64"},
    "obe_draw_indices": {3: "cdf_is_fresh", 6: "n_draws", 7: "d_idx", 8: "h_total_pinned"},
    "obe_resample_randoms_enqueue": {9: "h_i64"},
    "obe_mask_nonpositive_moments": {2: "n_dims", 8: "h_moments", 9: "h_changed"},
    "obe_mask_renorm_moments": {2: "n_dims", 7: "h_moments", 8: "h_changed"},
    "obe_bayes_update_model": {13: "h_out"},
    "obe_bayes_update_model_moments": {0: "m", 14: "h_out"},
    "obe_bayes_update_lik": {5: "h_out"},
    "obe_bayes_update_y": {14: "h_out"},
    "obe_mask_nonpositive": {6: "h_changed"},
    "obe_weight_sums": {4: "h_out"},
    "obe_weight_cdf": {4: "h_total"},
    "obe_utility_argmax": {8: "h_best", 9: "h_best_idx"},
    "obe_argmax": {2: "h_best", 3: "h_best_idx"},
}

audit = _Audit() if os.environ.get("OBE_CHECK_DELIVERY") == "1" else _NoAudit()

if audit.on and os.environ.get("OBE_AUDIT_REPORT"):
    import atexit
    import json

    def _report(path=os.environ["OBE_AUDIT_REPORT"]):
        # (appended: a test run starts several processes — ranks, tools — that all audit)
        with open(path, "a") as f:
            f.write(json.dumps(dict(audit.counts, pid=os.getpid(), pending_violations=audit.violations)) + "\n")
    atexit.register(_report)
