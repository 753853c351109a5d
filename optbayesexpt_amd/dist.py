"""Settings-axis sharding across the GPUs of one node (one process per GPU).

Every setting's utility is independent given the particle cloud, so
rank r sweeps the contiguous slice ``[r*N_s/G, (r+1)*N_s/G)`` of the flattened settings
and the cloud (34 MB at 1M particles) is replicated: each rank applies the same Bayes
update and the same resample — rank 0's generator state is broadcast when a sharded object
is built and whenever its ``rng`` is assigned (the reference's generator is unseeded,
particlepdf.py:142-145), so replicas stay bit-identical and no particle data ever crosses
xGMI; a periodic all-gather of a digest of (generator state, sum w, sum w^2) raises on
every rank if they ever differ.  The only data-path collective is the arg-max combine of
``opt_setting``: one all-gather of a 32-byte record per rank ``(best value, local index,
kappa, 0)`` straight from device memory over RCCL (``torch.distributed`` backend "nccl"), followed by a local first-max —
the message is latency-bound (tens of microseconds), irrelevant next to a
multi-millisecond sweep.  With backend "gloo" the same code runs on CPU tensors, which
is how the N>1 logic is tested without GPUs.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_settings, rank, world_size):
    """Contiguous, balanced partition: the first ``n % world`` ranks get one extra."""
    base, extra = divmod(int(n_settings), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def first_max(values, indices):
    """np.argmax tie rule on (value, global index) pairs: NaN wins, then the largest
    value, ties broken by the lowest global index.  Returns the winning position."""
    best = 0
    for k in range(1, len(values)):
        a_nan, b_nan = np.isnan(values[k]), np.isnan(values[best])
        if a_nan != b_nan:
            take = a_nan
        elif a_nan and b_nan:
            take = indices[k] < indices[best]
        elif values[k] != values[best]:
            take = values[k] > values[best]
        else:
            take = indices[k] < indices[best]
        if take:
            best = k
    return best


class SettingsShard:
    """This process's slice of the settings axis and its collectives."""

    def __init__(self, rank=None, world_size=None, group=None):
        self.group = group
        if rank is None or world_size is None:
            if not dist.is_initialized():
                raise RuntimeError("SettingsShard needs torch.distributed to be initialised "
                                   "(or explicit rank/world_size)")
            rank, world_size = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.world_size = int(rank), int(world_size)
        self._record_bufs = {}        # device -> (all-gather receive buffer, page-locked host copy)
        #: None, or a list that bench.py hangs here while it times cycles: every arg-max combine appends
        #: (device microseconds between "this rank's record is ready" and "everybody's has arrived" — events
        #: on the launch stream around the all-gather —, host microseconds of the whole call incl. the wait
        #: for this rank's own sweep)
        self.timing = None
        #: True: a world of ONE still goes through the backend's collectives (all-gather of the record from the
        #: workspace view into the page-locked landing zone, row gathers, broadcasts) instead of the local
        #: shortcuts — how the RCCL code path is executed on a one-GPU box (tests, `bench.py --force-dist`)
        self.always_collective = False
        self._starts = {}             # n_settings -> first global index of every rank's slice

    def bounds(self, n_settings):
        return shard_bounds(n_settings, self.rank, self.world_size)

    def connected(self):
        """True when there are other ranks to talk to (a shard built with an explicit rank / world_size in
        a process without torch.distributed — one slice of a sweep computed on its own — has none)."""
        return (self.world_size > 1 or self.always_collective) and dist.is_available() and dist.is_initialized()

    def _through_backend(self):
        return self.always_collective and dist.is_available() and dist.is_initialized()

    def _comm_device(self, device):
        backend = dist.get_backend(self.group)
        return torch.device(device) if backend == "nccl" else torch.device("cpu")

    def combine_records(self, record, n_settings):
        """Global first-maximum from each rank's 32-byte device record {value, local index
        (int64 bits), kappa, 0} (include/obe_hip.h: OBE_WS_RESULT_OFFSET): one all-gather
        straight from device memory, one copy to the host.  Returns (value, global index,
        worst kappa over ALL ranks) — the same triple on every rank, so that decisions taken
        from it (repeat the sweep with the variance shift?) keep the ranks' collectives in step."""
        w = self.world_size
        g = np.asarray(self._gather_records(record), dtype=np.float64).reshape(w, 4)    # (torch tensor or ndarray)
        vals = g[:, 0]
        local = np.ascontiguousarray(g[:, 1]).view(np.int64)
        starts = self._starts.get(n_settings)
        if starts is None:
            starts = self._starts[n_settings] = np.array([shard_bounds(n_settings, r, w)[0] for r in range(w)],
                                                         dtype=np.int64)
        gidx = local + starts
        k = first_max(vals, gidx)
        kappas = g[:, 2]
        worst = float("nan") if np.any(np.isnan(kappas)) else float(np.max(kappas))
        return float(vals[k]), int(gidx[k]), worst

    def _gather_records(self, record):
        """(world, 4) host array of every rank's record (valid until the next call: the page-locked landing zone
        itself — combine_records() consumes it at once)."""
        w = self.world_size
        if w == 1 and not self._through_backend():
            return record.cpu().numpy().reshape(1, 4)
        dev = self._comm_device(record.device)        # nccl: stay on the GPU; gloo: host tensors
        bufs = self._record_bufs
        if dev not in bufs:                           # receive buffer + page-locked landing zone + its numpy view, made once
            host = torch.empty(4 * w, dtype=torch.float64)
            if dev.type == "cuda":
                host = host.pin_memory()
            bufs[dev] = (torch.empty(4 * w, dtype=torch.float64, device=dev), host, host.numpy().reshape(w, 4))
        gathered, host, host_np = bufs[dev]
        timing = self.timing
        if timing is not None:
            import time
            t0 = time.perf_counter()
            if dev.type == "cuda":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
        # (the record is a contiguous float64 view of the sweep's workspace on this rank's device already)
        src = record if record.device == dev and record.is_contiguous() else record.contiguous().to(dev)
        dist.all_gather_into_tensor(gathered, src, group=self.group)
        if dev.type != "cuda":
            if timing is not None:
                us = 1e6 * (time.perf_counter() - t0)
                timing.append((us, us))
            return gathered.numpy().reshape(w, 4)
        if timing is not None:
            e1.record()
        host.copy_(gathered, non_blocking=True)       # one asynchronous copy, one wait
        torch.cuda.current_stream(dev).synchronize()
        if timing is not None:
            timing.append((1e3 * e0.elapsed_time(e1), 1e6 * (time.perf_counter() - t0)))
        return host_np

    def broadcast_from_rank0(self, values, device="cpu"):
        """Rank 0's host array on every rank (same shape and dtype everywhere): used for random
        draws that must be taken once for the whole job."""
        values = np.ascontiguousarray(values)
        if self.world_size == 1 and not self._through_backend():
            return values
        dev = self._comm_device(device)
        t = torch.from_numpy(values.copy()).to(dev)
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(t, src=src, group=self.group)
        return t.cpu().numpy()

    def broadcast_object_from_rank0(self, obj, device="cpu"):
        """Rank 0's picklable object on every rank (generator states: a dict with 128-bit integers)."""
        if self.world_size == 1 and not self._through_backend():
            return obj
        box = [obj if self.rank == 0 else None]
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dev = self._comm_device(device)
        dist.broadcast_object_list(box, src=src, group=self.group, device=dev)
        return box[0]

    def all_gather_int64(self, values, device="cpu"):
        """(world, len(values)) host array of every rank's int64 values (replica checks)."""
        v = np.ascontiguousarray(values, dtype=np.int64)
        if self.world_size == 1 and not self._through_backend():
            return v.reshape(1, -1)
        dev = self._comm_device(device)
        gathered = torch.empty(self.world_size * v.size, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gathered, torch.from_numpy(v.copy()).to(dev), group=self.group)
        return gathered.cpu().numpy().reshape(self.world_size, v.size)

    @staticmethod
    def make_record(value, local_index, kappa=0.0, device="cpu"):
        """A record as the sweep kernels leave it in the workspace (for tests / host paths)."""
        rec = torch.zeros(4, dtype=torch.float64)
        rec[0] = value
        rec[1:2].view(torch.int64)[0] = int(local_index)
        rec[2] = kappa
        return rec.to(device)

    def gather_rows(self, local, n_settings):
        """All ranks' (rows, n_local) slices assembled into a host (rows, n_settings) array."""
        rows = local.shape[0]
        if self.world_size == 1 and not self._through_backend():
            return local.cpu().numpy()
        dev = self._comm_device(local.device)
        widest = shard_bounds(n_settings, 0, self.world_size)
        pad = widest[1] - widest[0]
        buf = torch.zeros((rows, pad), dtype=torch.float64, device=dev)
        b, e = self.bounds(n_settings)
        buf[:, :e - b] = local[:, :e - b].to(dev)
        gathered = torch.empty(self.world_size * rows * pad, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(gathered, buf.reshape(-1), group=self.group)
        gathered = gathered.cpu().numpy().reshape(self.world_size, rows, pad)
        out = np.empty((rows, n_settings))
        for r in range(self.world_size):
            rb, re = shard_bounds(n_settings, r, self.world_size)
            out[:, rb:re] = gathered[r, :, :re - rb]
        return out
