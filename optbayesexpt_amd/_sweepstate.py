"""The state of one object's sweep path, in one place and with names.

``OptBayesExpt`` steers three things from cycle to cycle (the reference has none of them: its cycle is
``opt_setting`` -> measure -> ``pdf_update``, obe_base.py:733-756, 340-399, with nothing carried over but the
particles):

* which FORM of the model's sweep kernel the next sweep starts with (:class:`Form`): the fast one, whose
  branch-free batched divisions poison a sweep that leaves their exact range, or its always-in-range twin;
* whether the next full sweep accumulates its variance about a per-setting SHIFT (hysteresis on the
  cancellation factor kappa that every sweep reports);
* the sweep that ``pdf_update()`` ENQUEUES AHEAD of being asked for it (:class:`Pattern`, :class:`Pending`):
  when it is worth trying, what it swept, whether its update let it run, and what has to be waited for before
  its result words may be armed again.

Nothing here touches the device: results are waited for through an ``io`` object with two methods
(``wait_words(words, n, stream)``, ``still_armed(block)``, ``synchronize()``), the real one calling the
library, the one of tests/test_host_logic.py a recorder.  Every decision is a method with a name; the object
can be printed (``describe()``) when a run is being debugged.
"""
import enum


class Form(enum.Enum):
    FAST = "fast"      #: the fast form is tried first; a poisoned result (kappa = NaN) is repeated with the twin
    SAFE = "safe"      #: pinned to the twin: the fast form has left its range SAFE_STREAK sweeps in a row


class Pattern(enum.Enum):
    COLD = 0           #: no update -> sweep-of-that-cloud cycle seen (or the pattern just broke)
    WARM = 1           #: one such cycle
    STEADY = 2         #: two or more in a row: 'auto' speculation may start


class Pending(enum.Enum):
    NONE = "none"              #: nothing enqueued ahead
    ENQUEUED = "enqueued"      #: behind an update whose resample decision has not been delivered yet
    RAN = "ran"                #: the update did not resample: the sweep ran (or runs), its words will arrive
    ABORTED = "aborted"        #: the update resampled: the sweep's kernels did nothing, its words stay armed


class Ticket:
    """What was enqueued ahead: ``inputs`` — everything the sweep's result depends on, compared with what a
    fresh launch would use —, where its result lands (``words``/``block``: page-locked host words, or None for
    a sharded object, whose 32-byte ``record`` stays on the device) and the stream it was launched on."""

    __slots__ = ("inputs", "words", "block", "record", "stream", "stream_id")

    def __init__(self, inputs, words, block, record, stream, stream_id):
        self.inputs, self.words, self.block, self.record = inputs, words, block, record
        self.stream, self.stream_id = stream, stream_id


class SweepState:
    def __init__(self, io, limits):
        """``limits``: anything with KAPPA_ENTER, KAPPA_LEAVE, SAFE_STREAK, SAFE_RETRY (the owning object: read
        when used, so that assigning one of them on an object — they are class attributes — takes effect)."""
        self.io = io
        self.limits = limits
        # form
        self.safe_streak = 0          # consecutive sweeps that had to be repeated with the twin
        self.safe_run = 0             # sweeps since the fast form was last tried (while pinned)
        self.range_hint_key = None    # particles version the model's range_hint last looked at
        # shift
        self.unshifted = False        # the next full sweep of 'auto' mode runs unshifted
        # speculation
        self.streak = 0               # update -> sweep-of-that-cloud cycles in a row
        self.updated_cloud = None     # cloud the last fused update left behind (None: none / already swept)
        self.resample_rate = 0.0      # running share of updates that resampled
        self.unavailable = False      # the library refused the enqueue forms: never again for this object
        self.pending = Pending.NONE
        self.ticket = None

    # ------------------------------------------------------------------ form of the sweep kernel
    @property
    def form(self):
        return Form.FAST if self.safe_streak < self.limits.SAFE_STREAK else Form.SAFE

    def form_for_next_sweep(self):
        """Called once per (non-speculative) sweep.  While pinned to the twin, every ``safe_retry``-th sweep
        probes the fast form once more (a posterior that has narrowed may be back in range; a failure pins
        it again at once)."""
        if self.safe_streak >= self.limits.SAFE_STREAK:
            self.safe_run += 1
            if self.safe_run >= self.limits.SAFE_RETRY:
                self.safe_streak, self.safe_run = self.limits.SAFE_STREAK - 1, 0
        return self.form

    def fast_form_left_its_range(self):
        self.safe_streak += 1

    def fast_form_held(self):
        self.safe_streak = 0

    def range_hint(self, in_range):
        """A model's host-side prediction for a new cloud: False pins the twin without a poisoned attempt,
        True releases a pin for one fast attempt, None says nothing."""
        if in_range is False:
            self.safe_streak, self.safe_run = self.limits.SAFE_STREAK, 0
        elif in_range is True and self.safe_streak >= self.limits.SAFE_STREAK:
            self.safe_streak = self.limits.SAFE_STREAK - 1

    # ------------------------------------------------------------------ variance shift
    def shifted_for_next_sweep(self, mode, full):
        return (not full) or mode == "always" or (mode == "auto" and not self.unshifted)

    def sweep_reported_kappa(self, mode, full, shifted, kappa):
        """Hysteresis of 'auto': a shifted sweep with kappa < kappa_enter lets the next one run unshifted; an
        unshifted one that comes back with kappa > kappa_leave (or NaN) is not kept — returns True: repeat it
        shifted."""
        if not (full and mode == "auto"):
            return False
        if shifted:
            self.unshifted = bool(kappa < self.limits.KAPPA_ENTER)
            return False
        if not kappa <= self.limits.KAPPA_LEAVE:
            self.unshifted = False
            return True
        return False

    # ------------------------------------------------------------------ the update -> sweep pattern
    @property
    def pattern(self):
        return Pattern.STEADY if self.streak >= 2 else (Pattern.WARM if self.streak == 1 else Pattern.COLD)

    def update_finished(self, cloud, resampled):
        """End of pdf_update(): ``cloud`` = (particles version, weights version) if the fused update ran on
        the object's own cloud, else None."""
        self.updated_cloud = cloud
        self.resample_rate = 0.8 * self.resample_rate + (0.2 if resampled else 0.0)

    def full_sweep_requested(self, cloud):
        """A caller asks for the full sweep of ``cloud``: one more update -> sweep cycle if it is the cloud the
        last update left, else the pattern starts over."""
        seen, self.updated_cloud = self.updated_cloud, None
        self.streak = self.streak + 1 if seen == cloud else 0

    def speculation_wanted(self, mode, after_resample=False):
        """The policy half of the decision (the owner adds what only it can know: hooks, utility method).
        'auto': after two update -> sweep cycles in a row and while fewer than half of the recent updates
        resampled (a sweep behind a resampling update is launched for nothing); the sweep enqueued AFTER a
        resample has nothing to guess, the cloud is final."""
        if mode is False or mode == "never" or self.unavailable:
            return False
        if mode is True:
            return True
        return self.pattern is Pattern.STEADY and (after_resample or self.resample_rate < 0.5)

    # ------------------------------------------------------------------ the sweep enqueued ahead
    def enqueued(self, ticket, certain=False):
        """A sweep went out ahead of its request; ``certain``: not behind an undecided update (the sweep of a
        freshly resampled cloud): it runs."""
        self.ticket = ticket
        self.pending = Pending.RAN if certain else Pending.ENQUEUED

    def update_delivered(self, resampled):
        """The update's host words are in: its device-side resample test decided the fate of the sweep."""
        if self.pending is Pending.ENQUEUED:
            self.pending = Pending.ABORTED if resampled else Pending.RAN

    def take(self, inputs, stream_id):
        """The ticket of the sweep enqueued ahead if it IS the sweep being asked for now — it ran, on exactly
        these inputs, and its result can still be collected on the stream it was launched on; None (and the
        speculation forgotten, the pattern broken) if not.  For page-locked results the words are waited for
        here; words that are still armed after the stream drained mean the sweep never ran: None."""
        t = self.ticket
        if t is None:
            return None
        ok = (self.pending is Pending.RAN and t.inputs == inputs
              and (t.words is not None or t.stream_id == stream_id))
        if not ok:
            self.drop(stream_id)
            return None
        self.ticket, self.pending = None, Pending.NONE
        if t.words is None:
            return t
        self.io.wait_words(t.words, 3, t.stream)
        if self.io.still_armed(t.block):
            return None
        return t

    def drop(self, stream_id=None):
        """Forget a sweep nobody asked for.  If it ran, its kernels may still be running and WILL write the
        result words: they are waited for before anything arms those words again (a sharded object's record
        stays on the device: only a change of stream needs a synchronisation).  The pattern broke: two plain
        cycles before the next attempt."""
        t, was = self.ticket, self.pending
        self.ticket, self.pending = None, Pending.NONE
        if t is None:
            return
        if was is Pending.RAN:
            if t.words is not None:
                self.io.wait_words(t.words, 3, t.stream)
            elif t.stream_id != stream_id:
                self.io.synchronize()
            self.streak = 0

    def library_refused(self):
        self.unavailable = True

    # ------------------------------------------------------------------ introspection
    def describe(self):
        return dict(form=self.form.value, safe_streak=self.safe_streak, safe_run=self.safe_run,
                    unshifted=self.unshifted, pattern=self.pattern.name.lower(), streak=self.streak,
                    resample_rate=self.resample_rate, pending=self.pending.value, unavailable=self.unavailable)
