#!/usr/bin/env python
"""bench.py — model-evals/s per opt_setting()+pdf_update() cycle, fp64 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c5]

One "step" is one measurement cycle of the hot path on synthetic data:
``x = obe.opt_setting()`` (full settings x particles utility sweep, argmax) followed by
``obe.pdf_update((x, y, sigma))`` (Bayes update, N_eff test, resample when it triggers).
Default workload = BASELINE.json configs[2] ("c3": Lorentzian 3-param, 65 536 settings x
1 048 576 particles, the config the metric's target is quoted on; it fits one GPU).
With N > 1 (launched by torch.distributed.run, one process per GPU) the settings axis
is sharded — configs[3] — and the per-rank maxima are combined with one RCCL
all-gather; total work is fixed, so ``scaling`` is "strong".

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      — the dominant kernel (K1 sweep): achieved FP64 TFLOP/s from HIP events
                  around the kernel on its launch stream, against the FP64 vector peak;
                  the algorithmic HBM figure the north_star asks for rides along, and
                  ``valu_issue`` prices the kernel in FP64 issue slots counted in its ISA
                  (tools/count_isa.py) against the chip's issue rate — the flop roofline
                  treats every slot as an FMA, the kernel's mix is half mul/add.
  roofline_update — the HBM-bound Bayes update with the posterior's first moments (K2 + K3 pass 1, the two
                  launches of pdf_update()), bytes / time.
  cpu_baseline  — the NumPy oracle on one host core, on a bounded sub-grid (N = 1 only);
                  for c1 the oracle class itself through whole reference-semantics cycles.
  cpu_baseline_allcores — the plain C + OpenMP restatement on every host core (full-sweep configs).
  published_workload — the loop behind the only timing the reference publishes for this path (200 settings x
                  30 draws, 50 000 particles: 4.37 ms per cycle, hardware unstated), through this package.
  other_configs — (default c3 run, N = 1) the cycle loops of c1, c2 and c5 after the timed region, each with its
                  own bounded CPU leg.
  rccl          — (N > 1, or --force-dist) what the communicator itself reports: its world size, an all-gather of
                  the rank ids, the measured arg-max combine (in the timed cycles and on an idle stream),
                  every rank's cycle / K1 / everything-else milliseconds and the CPU affinity each rank gave itself
                  (the cores of its GPU's NUMA node, before torch was imported).
  projection    — (N > 1) what ONE rank's measured cycle on one GPU predicted for this config and world size
                  (profiles/shard_projection.json, tools/shard_cycle.py) next to what this run measured.
With --gpus N > 1 and no launcher around it (no WORLD_SIZE in the environment) the script starts its N
ranks itself, as fresh child processes, before anything here touches the GPU.
"""
import argparse
import ctypes
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X datasheet FP64 vector = 256 CU x 128 flop/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
FLOP_PER_EVAL = {"lorentzian": 10, "lorentzian7": 46}
# FP64 VALU issue slots per evaluation in the kernel's pair loop, counted in the gfx950 ISA
# (tools/count_isa.py; v_rcp_f64 = 4 slots): (unshifted, shifted)
ISSUE_SLOTS_PER_EVAL = {"lorentzian": (7.3125, 8.3125), "lorentzian7": (34.875, 35.875)}   # 7 peaks: 34.625 FP64 + 0.25 v_cndmask
# the form a sweep falls back to when the fast one leaves its range (last_sweep["safe"]): 7 peaks one by
# one, two particles per reciprocal, always shifted
ISSUE_SLOTS_SAFE_FORM = {"lorentzian7": 40.2}
VALU_ISSUE_PEAK = 256 * 4 * 16 * 2.4e9                   # lane-instructions/s: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz

CONFIGS = {
    # name: (n_settings, n_particles, model, description)
    "c1": (201, 5000, "lorentzian", "Lorentzian 3-param, 201 settings x 5 000 particles, reference semantics "
                                    "(N_DRAWS = 30 weighted draws) — the reference's own CPU-sized case"),
    "c2": (4096, 262144, "lorentzian", "Lorentzian 3-param, 4 096 settings x 262 144 particles"),
    "c3": (65536, 1048576, "lorentzian", "Lorentzian 3-param, 65 536 settings x 1 048 576 particles"),
    "c5": (16384, 524288, "lorentzian7", "7-Lorentzian sum, 10 params, 16 384 settings x 524 288 particles, "
                                         "OptBayesExptNoiseParameter"),
}


def make_workload(cfg, seed=20240424):
    """Synthetic inputs of SURVEY.md §8(d)."""
    ns, n_p, model, _ = CONFIGS[cfg]
    g = np.random.default_rng(seed)
    settings = (np.linspace(1.5, 4.5, ns),)
    if model == "lorentzian":
        prior = np.array([g.uniform(2, 4, n_p), g.uniform(-2000, -400, n_p), g.normal(50000, 1000, n_p)])
        true = (3.0, -1000.0, 50000.0)
        sigma = 500.0
    else:
        prior = np.vstack([g.uniform(2, 4, (7, n_p)), g.uniform(400, 2000, (1, n_p)),
                           g.normal(500, 1000, (1, n_p)), g.exponential(500, (1, n_p))])
        true = (2.2, 2.5, 2.8, 3.1, 3.4, 3.7, 3.9, 1000.0, 500.0, 500.0)
        sigma = 500.0
    return settings, prior, (0.1,), true, sigma


def build_obe(cfg, shard, settings, prior, cons):
    import optbayesexpt_amd as obe
    model = CONFIGS[cfg][2]
    if model == "lorentzian":
        return obe.OptBayesExpt(obe.models.lorentzian(1), settings, prior, cons, scale=False,
                                utility_method="variance_approx" if cfg == "c1" else "variance_full",
                                default_noise_std=500.0, settings_shard=shard)
    return obe.OptBayesExptNoiseParameter(obe.models.lorentzian(7), settings, prior, cons, scale=False,
                                          utility_method="variance_full", noise_parameter_index=9,
                                          settings_shard=shard)


def cpu_baseline(cfg, settings, prior, cons, true, sigma, target_s=8.0):
    """The oracle (NumPy, one core) on a bounded sample of the same workload: the full
    particle cloud against a sub-grid of evenly spaced settings for the sweep, plus the
    full update.  A short probe sizes the sub-grid for ~``target_s`` seconds of CPU work.
    Throughput = algorithmic evals / time: the evals/s of the reference-style NumPy path
    on this host (it scales linearly in the number of settings)."""
    import oracle
    from oracle import models as om
    fn = om.lorentzian if CONFIGS[cfg][2] == "lorentzian" else om.multi_lorentzian(7)
    ns, n_p = CONFIGS[cfg][0], CONFIGS[cfg][1]
    w = np.full(n_p, 1.0 / n_p)
    if cfg == "c1":
        # reference semantics: the oracle class itself through whole opt_setting + pdf_update
        # cycles (N_DRAWS = 30 weighted draws, resamples included), like the timed GPU loop
        o = oracle.OracleOptBayesExpt(fn, settings, prior.copy(), cons, scale=False, default_noise_std=sigma)
        o.rng = np.random.default_rng(1)
        sim = np.random.default_rng(2)
        n_cycles, t0 = 0, time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            while time.perf_counter() - t0 < min(target_s, 3.0):
                x = o.opt_setting()
                o.pdf_update((x, float(fn(x, true, cons)) + sigma * sim.standard_normal(), sigma))
                n_cycles += 1
        dt = time.perf_counter() - t0
        return {"value": n_cycles * (ns * o.N_DRAWS + n_p) / dt, "unit": "model-evals/s", "cores": 1, "kind": "port",
                "sample": f"{n_cycles} opt_setting + pdf_update cycles of the oracle class ({ns} settings x "
                          f"{o.N_DRAWS} draws + {n_p}-particle update each), {dt:.1f} s, {1e3 * dt / n_cycles:.3f} ms per cycle",
                "host_cpus": os.cpu_count()}

    # evenly spaced sub-grids of 64 settings each, one full cycle per sub-grid (sweep of the sub-grid over the
    # whole cloud + the full update), until the time budget is used: the sample is bounded by the clock, not
    # by an extrapolation from a probe (round 3's probe undershot: 20.8 s for a 10 s target)
    step = min(64 if target_s >= 6.0 else 8, ns)       # (a short leg takes smaller blocks: the clock is checked per block)
    n_blocks = ns // step
    n_used, t0 = 0, time.perf_counter()
    for b in range(0, n_blocks, max(1, n_blocks // 32)):           # up to 32 blocks, evenly spaced over the grid
        block = slice(b * step, (b + 1) * step)
        sub = (np.ascontiguousarray(settings[0][block]),)
        yvar = oracle.yvar_full_sweep(fn, oracle.flatten_settings(sub), prior, w, cons, chunk=4096)
        util = oracle.utility_from_yvar(yvar, sigma ** 2, 1.0)
        x = (sub[0][int(np.argmax(util))],)
        lik = oracle.gauss_likelihood(fn(x, prior, cons), float(fn(x, true, cons)), sigma)
        oracle.effective_particles(oracle.normalized_product(w, lik))
        n_used += step
        if time.perf_counter() - t0 >= target_s:
            break
    dt = time.perf_counter() - t0
    n_cycles = n_used // step
    evals = n_used * n_p + n_cycles * n_p
    return {"value": evals / dt, "unit": "model-evals/s", "cores": 1, "kind": "port",
            "sample": f"{n_used} of {ns} settings (in {n_cycles} evenly spaced blocks of {step}) x all {n_p} particles "
                      f"(two-pass weighted variance, chunked) + {n_cycles} full {n_p}-particle updates, {dt:.1f} s; "
                      f"NumPy ufuncs are single-threaded",
            "host_cpus": os.cpu_count()}


def cpu_baseline_allcores(cfg, settings, prior, cons, true, sigma, target_s=6.0):
    """The same cycle as a plain-C restatement (oracle/csweep.c) on every host core
    (OpenMP over settings): what the host of this box can do at best, next to the faithful
    single-threaded NumPy figure.  Lorentzian configs only."""
    from oracle import csweep
    ns, n_p, model = CONFIGS[cfg][0], CONFIGS[cfg][1], CONFIGS[cfg][2]
    k = 1 if model == "lorentzian" else 7
    w = np.full(n_p, 1.0 / n_p)

    step = max(64, 8 * csweep.threads())
    n_used, n_cycles, t0 = 0, 0, time.perf_counter()
    for lo in range(0, ns - step + 1, step):
        sub = np.ascontiguousarray(settings[0][lo:lo + step])
        yvar = csweep.lorentz_yvar(sub, prior, w, cons[0], k)
        x = sub[int(np.argmax(yvar))]
        csweep.lorentz_update(x, 49500.0, sigma, prior, w, cons[0], k)
        n_used += step
        n_cycles += 1
        if time.perf_counter() - t0 >= target_s:
            break
    dt = time.perf_counter() - t0
    return {"value": (n_used * n_p + n_cycles * n_p) / dt, "unit": "model-evals/s", "cores": csweep.threads(),
            "kind": "port", "implementation": "plain C + OpenMP (oracle/csweep.c), two-pass weighted variance",
            "sample": f"{n_used} of {ns} settings (blocks of {step}) x all {n_p} particles + {n_cycles} full updates, {dt:.1f} s"}


def published_workload(cycles=1500, warm=100):
    """The only timing the reference publishes for this path (docs/manual_demos.rst:
    337-352): 3000 cycles of good_setting() + pdf_update() + std(), Lorentzian 3-parameter, 200 settings
    x 30 draws, 50 000 particles (demos/numba/numbaLorentzian.py) took 13.109 s without numba and
    8.322 s with it, on unstated hardware.  The same loop through this package, timed here; a
    different workload from the headline metric's, so it is reported beside it, not as vs_baseline."""
    import optbayesexpt_amd as obe_pkg
    g = np.random.default_rng(0)
    n, ns = 50000, 200
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    o = obe_pkg.OptBayesExpt(obe_pkg.models.lorentzian(), (np.linspace(1.5, 4.5, ns),), prior, (0.1,), scale=False)
    o.rng = np.random.default_rng(1)
    sim = np.random.default_rng(2)
    fn, true, sigma = o.model_function, (3.0, -1000.0, 50000.0), 500.0
    resamples, t0 = 0, 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for c in range(cycles + warm):
            if c == warm:
                t0 = time.perf_counter()
            x = o.good_setting(pickiness=19)
            o.pdf_update((x, float(fn(x, true, (0.1,))) + sigma * sim.standard_normal(), sigma))
            o.std()
            resamples += c >= warm and bool(o.just_resampled)
    ms = 1e3 * (time.perf_counter() - t0) / cycles
    evals = ns * o.N_DRAWS + n
    return {"workload": "demos/numba/numbaLorentzian.py loop: good_setting(pickiness=19) + pdf_update + std per cycle, "
                        "200 settings x 30 draws, 50 000 particles", "cycles": cycles, "resamples": int(resamples),
            "ms_per_cycle": ms, "model_evals_per_s": evals / (ms * 1e-3),
            "reference_published_ms_per_cycle": {"numpy": 13109.0 / 3000, "numba": 8322.0 / 3000,
                                                 "hardware": "not stated", "source": "docs/manual_demos.rst:337-352"},
            "speedup_vs_published_numpy": (13109.0 / 3000) / ms}


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def format_cpulist(cpus):
    cpus, parts = sorted(set(cpus)), []
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(parts)


def gpu_local_cpus(index, sysfs="/sys"):
    """(cpus, numa node) local to HIP device ``index`` of this node, read from sysfs ONLY — no HIP, no torch, no
    rocm-smi: this runs in the launcher, and in every rank before anything has touched the GPU.  The runtime
    enumerates GPUs in KFD topology order (the nodes with SIMDs under /sys/class/kfd/kfd/topology/nodes); each
    names its DRM render node, whose PCI device directory carries local_cpulist / numa_node.  ROCR_VISIBLE_DEVICES
    / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES lists of integers are applied in that order.  None if anything is
    missing (a container without /sys/class/kfd, UUID-style visibility lists): then nothing is pinned."""
    try:
        nodes = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
        gpus = []
        for name in sorted(os.listdir(nodes), key=int):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(nodes, name, "properties")) if len(line.split()) >= 2)
            except OSError:
                continue        # (a GPU of the node that this container was not given: EPERM — the runtime does not see it either)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(int(props["drm_render_minor"]))
        order = list(range(len(gpus)))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            val = os.environ.get(var)
            if val is None or (var == "CUDA_VISIBLE_DEVICES" and "HIP_VISIBLE_DEVICES" in os.environ):
                continue
            order = [order[int(k)] for k in val.split(",") if k.strip() != ""]
        dev = os.path.join(sysfs, f"class/drm/renderD{gpus[order[index]]}/device")
        cpus = parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())
        node = int(open(os.path.join(dev, "numa_node")).read())
        return (cpus, node) if cpus else None
    except (OSError, ValueError, KeyError, IndexError):
        return None


def pin_to_gpu_numa_node(local_rank):
    """Before any torch / HIP import of a rank of an N > 1 run: restrict this process (and every thread it will
    start) to the cores of its GPU's NUMA node.  A rank's cycle holds ~0.25 ms that does not shard — host round
    trips, spin-waits on page-locked result words that the GPU writes over PCIe, the Python between two library
    calls — and a rank scheduled on the far socket pays the inter-socket hop on every one of them.  OBE_BENCH_CPUS
    (a cpulist, set by this script's own launcher) wins; OBE_BENCH_PIN=0 switches it off.  Returns what was done,
    for the "rccl" block of the line."""
    if os.environ.get("OBE_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return {"pinned": False, "why": "switched off" if hasattr(os, "sched_setaffinity") else "no sched_setaffinity"}
    node = None
    given = os.environ.get("OBE_BENCH_CPUS")
    if given:
        cpus = parse_cpulist(given)
        node = int(os.environ.get("OBE_BENCH_NUMA_NODE", "-1"))
    else:
        found = gpu_local_cpus(local_rank)
        if found is None:
            return {"pinned": False, "why": "no KFD topology / local_cpulist in sysfs"}
        cpus, node = found
    allowed = os.sched_getaffinity(0)
    want = sorted(set(cpus) & allowed)
    if not want:
        return {"pinned": False, "why": "the GPU's local cores are outside this process's cpuset", "numa_node": node}
    try:
        os.sched_setaffinity(0, want)
    except OSError as exc:
        return {"pinned": False, "why": str(exc), "numa_node": node}
    return {"pinned": True, "numa_node": node, "cpus": format_cpulist(want), "n_cpus": len(want)}


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher around it: start one child process per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, as torch.distributed.run would set
    them) BEFORE this process has touched the GPU — it never does: no torch import, no HIP call —
    relay rank 0's single JSON line and exit non-zero if any rank does.  Children are fresh
    interpreters (subprocess, not fork / exec of an initialised process)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OBE_BENCH_CHILD="1")
        # This pool's host driver only supports dmabuf IPC: with the legacy mode RCCL's intra-node transport (and any
        # device-tensor sharing across processes) fails at communicator set-up with `hipIpcGetMemHandle: invalid
        # argument`.  The image exports the variable already; a rank started from a scrubbed environment needs it too
        # ("More than one GPU").  Harmless where the legacy mode works: it only selects the other path.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the cores of the rank's GPU, found here (the launcher never touches a GPU) and applied by the child before
        # it imports torch (pin_to_gpu_numa_node)
        local = None if os.environ.get("OBE_BENCH_ONE_DEVICE") else gpu_local_cpus(r)
        if local is None and os.environ.get("OBE_BENCH_ONE_DEVICE"):
            local = gpu_local_cpus(0)
        if local is not None and "OBE_BENCH_CPUS" not in os.environ:
            env["OBE_BENCH_CPUS"], env["OBE_BENCH_NUMA_NODE"] = format_cpulist(local[0]), str(local[1])
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = procs[0].stdout
    lines, rcs, failed_at = [], [None] * n, None
    import selectors
    sel = selectors.DefaultSelector()
    sel.register(out0, selectors.EVENT_READ)
    open_pipe = True
    while any(rc is None for rc in rcs) or open_pipe:
        if open_pipe and sel.select(timeout=0.2):
            chunk = out0.readline()
            if chunk:
                lines.append(chunk.decode(errors="replace"))
            else:
                open_pipe = False
                sel.unregister(out0)
        elif not open_pipe:
            time.sleep(0.2)
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad and failed_at is None:
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 10.0:
            # a rank died: the others wait in a collective for ever — end exactly the processes started here
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.kill()
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write(f"bench.py: ranks failed (rank, exit code): {bad}\n")
        return 1
    return 0


def projection_from_one_rank(cfg, world, measured_ms_per_step, combine_us):
    """What tools/shard_cycle.py predicted for this (config, world size) from ONE rank's measured cycle on ONE GPU
    (profiles/shard_projection.json, committed with the round's profiles), put next to what this run measured, so
    that a SCALE line can be read against the projection without another round: predicted speed-up = one-GPU
    cycle / (rank cycle + all-gather), for the all-gather latencies the projection brackets and for the combine
    THIS run measured on an idle stream."""
    path = os.path.join(ROOT, "profiles", "shard_projection.json")
    try:
        e = json.load(open(path))[cfg][str(world)]
    except (OSError, KeyError, ValueError):
        return {"available": False, "why": f"no entry for {cfg} x {world} ranks in profiles/shard_projection.json"}
    one, rank_ms = e["one_gpu_ms_per_cycle"], e["rank_ms_per_cycle"]
    out = {"available": True, "source": "profiles/shard_projection.json (tools/shard_cycle.py: one rank's slice through "
                                        "the same cycles on one GPU, the collective replaced by a device-to-host read)",
           "projection_cycles": {"steps": e["steps"], "warmup": e["warmup"], "resamples": e["resamples"]},
           "one_gpu_ms_per_cycle_then": one, "rank_ms_per_cycle_then": rank_ms,
           "predicted_speedup_from_rank_cycle": {f"{us}us_all_gather": one / (rank_ms + 1e-3 * us) for us in (0, 30, 60, 100)},
           "measured_ms_per_step_now": measured_ms_per_step,
           "measured_speedup_vs_one_gpu_cycle_then": one / measured_ms_per_step,
           "note": "the driver computes the scaling ratio itself from its own N = 1 run; this block only says what one "
                   "rank's measured cycle predicted, under the cycle mix (resamples) the projection was made with"}
    if combine_us is not None:
        out["predicted_speedup_with_this_runs_idle_combine"] = one / (rank_ms + 1e-3 * combine_us)
    return out


def collective_report(obe, shard, backend, world, rank, ns, steps, elapsed_local, k1_total_ms, k1_launches, k1_cycles,
                      affinity=None):
    """The "rccl" block of an N > 1 line: proof that the communicator saw N ranks, and where each rank's time
    went.  (1) an all-gather in which every rank contributes its rank id; (2) the arg-max combine of the timed
    cycles: device microseconds between this rank's record being ready and everybody's having arrived (events
    on the launch stream around the all-gather: the collective itself + the skew between the ranks' sweeps);
    (3) the same call on an idle stream, 200 times back to back: the bare latency of the 32-byte all-gather +
    copy to the host; (4) per rank: ms per cycle, K1 ms per sweep, everything else — min and max over the ranks."""
    import torch
    import torch.distributed as dist
    from optbayesexpt_amd import _lib
    dev = "cuda" if backend == "nccl" else "cpu"
    timing, shard.timing = shard.timing or [], None
    ids = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int64, device=dev))
    ids = ids.cpu().tolist()
    off = _lib.OBE_WS_RESULT_OFFSET
    rec = obe._ws[off:off + 4].clone()
    torch.cuda.synchronize()
    dist.barrier()
    idle = []
    for _ in range(200):
        t0 = time.perf_counter()
        shard.combine_records(rec, ns)
        idle.append(1e6 * (time.perf_counter() - t0))
    in_cycle_dev = [a for a, _ in timing]
    in_cycle_host = [b for _, b in timing]
    cycle_ms = 1e3 * elapsed_local / steps
    k1_ms = k1_total_ms / max(k1_launches, 1)
    k1_per_cycle = k1_total_ms / max(k1_cycles, 1)       # (K1 and the combine: events in k1_cycles further cycles)
    mine = torch.tensor([cycle_ms, k1_ms, cycle_ms - k1_per_cycle,
                         float(np.median(in_cycle_dev)) if in_cycle_dev else float("nan"),
                         float(np.max(in_cycle_dev)) if in_cycle_dev else float("nan"),
                         float(np.median(idle)), float(len(timing)), float(k1_launches)],
                        dtype=torch.float64, device=dev)
    table = torch.empty(world * mine.numel(), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(table, mine)
    table = table.cpu().numpy().reshape(world, -1)

    def span(col):
        vals = table[:, col]
        if np.all(np.isnan(vals)):            # (a world of one gathers nothing in its cycles: null, not NaN, in the JSON)
            return {"min": None, "max": None, "per_rank": [None] * len(vals)}
        return {"min": float(np.nanmin(vals)), "max": float(np.nanmax(vals)),
                "per_rank": [None if np.isnan(v) else float(v) for v in vals]}

    masks = [None] * world
    dist.all_gather_object(masks, affinity)
    return {"backend": dist.get_backend(), "world_size_reported_by_backend": dist.get_world_size(),
            "cpu_affinity_per_rank": masks,
            "cpu_affinity_note": "set by each rank before it imported torch: the cores of its GPU's NUMA node (sysfs: KFD "
                                 "topology -> DRM render node -> local_cpulist); OBE_BENCH_PIN=0 switches it off",
            "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None),
            "all_gather_rank_ids": ids, "all_gather_rank_ids_ok": ids == list(range(world)),
            "instrumented_cycles": int(k1_cycles),
            "instrumented_cycles_note": "the timed steps run without events; K1 (events around the sweep kernel) and the "
                                        "combine (events around the all-gather) are measured in this many further cycles "
                                        "of the same experiment right after them",
            "combines_in_instrumented_cycles": int(table[0, 6]), "sweep_launches_in_instrumented_cycles": int(table[0, 7]),
            "cycle_ms": span(0), "k1_ms_per_sweep": span(1), "non_k1_ms_per_cycle": span(2),
            "combine_us_in_cycle_median": span(3), "combine_us_in_cycle_max": span(4),
            "combine_us_in_cycle_note": "events on the launch stream around the 32-byte all-gather of the arg-max "
                                        "records: the collective + the wait for the slowest rank's sweep",
            "combine_us_idle_median": span(5),
            "combine_us_idle_note": "200 back-to-back combine_records() calls on an idle stream (host clock): "
                                    "all-gather + copy of the records to the host + stream synchronisation"}


def warm_clocks(obe, at_least_ms=150.0):
    """Untimed, before the warm-up steps: the sweep kernel of this object back to back for ~150 ms.  An MI355X that
    has been idle (building the object takes the host a second) needs ~35 ms of sustained load before its clocks
    reach their sustained level: K1 of one rank's c5 slice takes 1.24 ms in the first cycle and 1.04 ms from the
    22nd on (profiles/r05_k1_per_cycle_c5_cold.txt).  Three warm-up steps of 1.4 ms do not get there, and a short
    sharded run would be timed entirely inside the ramp while the 14 ms cycles of the one-GPU run are not — the
    scaling ratio would compare a cold chip with a warm one.  A real experiment runs thousands of cycles."""
    import torch
    from optbayesexpt_amd import _lib
    from optbayesexpt_amd.particlepdf import _ptr
    if obe.utility_method != "variance_full" or obe._s_end <= obe._s_begin:
        return 0.0
    mom = obe._moments_on_device()
    p, w = obe._pw_tensors()
    s_ptr = ctypes.c_void_p(obe._settings_dev.data_ptr() + 8 * obe._s_begin)
    ms = ctypes.c_float(0.0)

    def run(iters):
        obe._mlib.call("obe_sweep_kernel_time", obe._model_struct, s_ptr, obe._n_settings, obe._s_end - obe._s_begin,
                       _ptr(p), p.shape[1], obe.n_particles, _ptr(w), _ptr(mom), _lib.OBE_SWEEP_SHIFTED, _ptr(obe._ws),
                       obe._ws_bytes, iters, ctypes.byref(ms), obe._stream())
        return ms.value
    one = max(run(1), 1e-3)
    total = 2.0 * one
    left = at_least_ms - total
    if left > 0:
        iters = int(min(2000, max(1, round(left / one))))
        total += iters * run(iters)
    torch.cuda.synchronize()
    return total


def warm_resample_path(cfg, settings, prior, cons):
    """Untimed, before the warm-up steps: ONE forced resample of a scratch object of the workload's shape.  A process's
    first launch of a kernel loads its code object, and a resample is ~15 kernels that the warm-up steps may never reach
    (whether they resample depends on the simulated measurements): on a freshly started box the first resample INSIDE the
    timed steps then costs milliseconds — one c3 run of the round's collection measured 14.7 ms per step with a median
    step of 13.5 ms for exactly that reason.  Warm-up, like the warm-up steps themselves: nothing of the timed object
    is touched."""
    import torch
    scratch = build_obe(cfg, None, (settings[0][:: max(1, settings[0].size // 64)],), prior.copy(), cons)
    scratch.rng = np.random.default_rng(99)
    scratch.tuning_parameters["resample_threshold"] = 1.0        # this update resamples, whatever the data
    x = (float(settings[0][settings[0].size // 2]),)
    y = float(np.mean(scratch.model_function(x, prior[:, :64], cons)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        # (the cycle's own route: update -> resample test -> pipelined resample [-> constraint mask] -> moments)
        scratch.pdf_update((x, y, 500.0) if CONFIGS[cfg][2] == "lorentzian" else (x, y))
        scratch.mean()
    torch.cuda.synchronize()            # (a warm-up: nothing depends on whether the scratch update really resampled)
    del scratch


def update_at_boundary(obe, record):
    """pdf_update() for the last cycle before a timed region starts or ends.  From the third cycle on,
    pdf_update() enqueues the NEXT cycle's sweep behind its update (obe_base.py: speculative sweep); a sweep
    enqueued by the last warm-up cycle would be work of the first timed cycle done before the clock starts, and
    one enqueued by the last timed cycle work of a cycle that is not counted.  Those two updates run without
    it, so that a timed region of K cycles holds exactly K sweeps and K updates, start to finish."""
    had = "speculative_sweep" in obe.tuning_parameters
    prev = obe.tuning_parameters.get("speculative_sweep")
    obe.tuning_parameters["speculative_sweep"] = False
    try:
        obe.pdf_update(record)
    finally:
        if had:
            obe.tuning_parameters["speculative_sweep"] = prev
        else:
            del obe.tuning_parameters["speculative_sweep"]


def other_config(cfg, steps, warmup):
    """One of the other single-GPU BASELINE configs through the same cycle loop, after (outside) the main
    timed region: the driver's line then carries every config, not just the headline one.  The same
    measures as the headline: steps of opt_setting() + pdf_update() incl. the resamples the data trigger,
    K1 timed by HIP events inside those cycles."""
    import torch
    ns, n_p, model, _ = CONFIGS[cfg]
    settings, prior, cons, true, sigma = make_workload(cfg)
    obe = build_obe(cfg, None, settings, prior.copy(), cons)
    obe.rng = np.random.default_rng(1234)
    sim = np.random.default_rng(4321)
    fn = obe.model_function
    noise_rec = model == "lorentzian"
    warm_resample_path(cfg, settings, prior, cons)
    warm_clocks(obe, 50.0)             # (the chip is warm from the main run; building this object left it idle)
    step_ms, res = [], []
    full = obe.utility_method == "variance_full"
    # the cycles are timed WITHOUT the event pair around the sweep kernel (at these sizes the two event packets
    # cost 10-25 us of a 0.3 ms cycle); K1 is timed by those events in a few more cycles of the same experiment
    k1_steps = max(4, steps // 4) if full else 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        for c in range(warmup + steps + k1_steps):
            if c == warmup:
                torch.cuda.synchronize()
                t_all = time.perf_counter()
            if c == warmup + steps:
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t_all
                obe._mlib.call("obe_sweep_timing", 1, None, None)
            ts = time.perf_counter()
            x = obe.opt_setting()
            y = float(fn(x, true, cons)) + sigma * sim.standard_normal()
            rec = (x, y, sigma) if noise_rec else (x, y)
            if c in (warmup - 1, warmup + steps - 1):
                update_at_boundary(obe, rec)
            else:
                obe.pdf_update(rec)
            if warmup <= c < warmup + steps:
                step_ms.append(1e3 * (time.perf_counter() - ts))
                res.append(bool(obe.just_resampled))
    torch.cuda.synchronize()
    if not k1_steps:
        elapsed = time.perf_counter() - t_all
    k1_ms, k1_n = ctypes.c_double(0.0), ctypes.c_int64(0)
    obe._mlib.call("obe_sweep_timing", 0, ctypes.byref(k1_ms), ctypes.byref(k1_n))
    n_draws = n_p if full else obe.N_DRAWS
    step_ms, res = np.array(step_ms), np.array(res)
    out = {"workload": CONFIGS[cfg][3], "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "value": steps * (ns * n_draws + n_p) / elapsed, "unit": "model-evals/s", "resamples": int(res.sum()),
           "median_ms_plain_cycle": float(np.median(step_ms[~res])) if (~res).any() else None,
           "median_ms_resample_cycle": float(np.median(step_ms[res])) if res.any() else None}
    if full and k1_n.value:
        k1 = k1_ms.value / k1_n.value
        out["k1_ms"] = k1
        out["k1_timing"] = f"HIP events around the sweep kernel in {k1_steps} further cycles of the same experiment"
        out["roofline_frac"] = FLOP_PER_EVAL[model] * ns * n_p / (k1 * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS
        shifted, safe = bool(obe.last_sweep["shifted"]), bool(obe.last_sweep.get("safe"))
        slots = ISSUE_SLOTS_SAFE_FORM.get(model, ISSUE_SLOTS_PER_EVAL[model][1]) if safe \
            else ISSUE_SLOTS_PER_EVAL[model][1 if shifted else 0]
        out["valu_issue_frac"] = slots * ns * n_p / (k1 * 1e-3) / VALU_ISSUE_PEAK
        out["variant"] = ("safe " if safe else "") + ("shifted" if shifted else "unshifted")
    else:       # reference semantics: a one-workgroup sweep of N_DRAWS draws, latency-bound (no roofline to speak of)
        out["k1_ms"] = None
        out["roofline_frac"] = None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the c1 / c2 / c5 cycles that follow the timed region of the default (c3) run")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even with one rank (exercises the N > 1 code path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around this process: be the launcher (nothing here has touched the GPU yet)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (N = 1 too: the GPU part of the run — c1's cycle is mostly host time — runs on the GPU's own socket; the full mask is
    # restored before the CPU legs, which use every core)
    all_cpus = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    affinity = {"pinned": False, "why": "switched off"}
    if os.environ.get("OBE_BENCH_PIN", "1") != "0":
        # BEFORE torch / HIP are imported (threads started later inherit the mask; nothing here touches the GPU)
        try:
            affinity = pin_to_gpu_numa_node(0 if os.environ.get("OBE_BENCH_ONE_DEVICE") else local_rank)
        except Exception as exc:          # (placement is an optimisation: never a reason for a rank to die)
            affinity = {"pinned": False, "why": f"{type(exc).__name__}: {exc}"}
    import torch
    import torch.distributed as dist
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launched by a torch.distributed.run with "
                 f"a different --nproc-per-node?)")
    # test hooks (flow check on a 1-GPU box): OBE_BENCH_BACKEND=gloo OBE_BENCH_ONE_DEVICE=1 lets
    # several ranks share cuda:0; the numbers of such a run mean nothing
    backend = os.environ.get("OBE_BENCH_BACKEND", "nccl")
    if os.environ.get("OBE_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    shard = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # (dmabuf IPC only on this pool: see launch_ranks)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when the communicator comes up; keep
        # stdout for the one JSON line by pointing fd 1 at stderr until it is up
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            warm = torch.zeros(4, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)       # RCCL printf()s into libc's buffer: flush it to stderr too
            os.dup2(saved, 1)
            os.close(saved)
        from optbayesexpt_amd import SettingsShard
        shard = SettingsShard()
        shard.always_collective = True        # (--force-dist, a world of one: still through the backend's collectives)

    from optbayesexpt_amd import _lib
    from optbayesexpt_amd.particlepdf import _ptr

    cfg = args.config
    ns, n_p, model, desc = CONFIGS[cfg]
    settings, prior, cons, true, sigma = make_workload(cfg)
    obe = build_obe(cfg, shard, settings, prior.copy(), cons)
    obe.rng = np.random.default_rng(1234)               # identical on every rank: replicas stay in step
    sim = np.random.default_rng(4321)
    truth_fn = obe.model_function          # the DeviceModel's NumPy form, as the demos' simulators use it
    noise_rec = model == "lorentzian"

    def one_step(boundary=False):
        x = obe.opt_setting()
        y = float(truth_fn(x, true, cons)) + sigma * sim.standard_normal()
        rec = (x, y, sigma) if noise_rec else (x, y)
        if boundary:        # last cycle before the clock starts / stops: no sweep enqueued across the boundary
            update_at_boundary(obe, rec)
        else:
            obe.pdf_update(rec)
        return int(obe.just_resampled)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    if os.environ.get("OBE_BENCH_DIE_RANK") == str(rank) and world > 1:
        os._exit(17)       # test hook (tests/test_gpu_two_ranks.py): a rank that dies while its peers head for a collective
    try:
        warm_resample_path(cfg, settings, prior, cons)      # (untimed: the resample kernels' code objects)
    except Exception as exc:                                 # a warm-up must never take the benchmark down
        print(f"bench.py: resample warm-up skipped ({exc})", file=sys.stderr)
    warmed_ms = warm_clocks(obe)          # (untimed; see warm_clocks: the chip's clocks, not the code's caches)
    # state for the benchmark: a few real updates (non-uniform weights), SURVEY §8(d)
    for k in range(max(args.warmup, 0)):
        one_step(boundary=k == args.warmup - 1)
    barrier()
    # K1 inside the timed cycles: HIP events around every sweep-kernel launch on its own stream.  One GPU: in the
    # timed steps themselves (two barrier packets per sweep: 0.1 % of a 14 ms cycle).  Sharded (N > 1): a rank's
    # cycle is 1.3-2 ms and carries a second pair of events around the arg-max all-gather — ~25 us per cycle, 1-2 %
    # that the one-GPU line does not pay and that would come off the scaling ratio —, so the timed steps run
    # uninstrumented and K1 / the combine are measured in further cycles of the same experiment right after them.
    instrument_in_timed_steps = not use_dist
    if instrument_in_timed_steps:
        obe._mlib.call("obe_sweep_timing", 1, None, None)
    t0 = time.perf_counter()
    resamples = 0
    step_ms, step_resampled = [], []
    for k in range(args.steps):
        ts = time.perf_counter()                 # every step ends on pdf_update's device sync
        r = one_step(boundary=k == args.steps - 1)
        step_ms.append(1e3 * (time.perf_counter() - ts))
        step_resampled.append(r)
        resamples += r
    barrier()
    elapsed = time.perf_counter() - t0
    k1_cycles = args.steps
    if not instrument_in_timed_steps:
        k1_cycles = max(4, min(args.steps, 12))
        obe._mlib.call("obe_sweep_timing", 1, None, None)
        shard.timing = []                 # events around every arg-max all-gather (dist.SettingsShard.timing)
        for k in range(k1_cycles):
            one_step(boundary=k == k1_cycles - 1)
        barrier()
    k1_total_ms, k1_launches = ctypes.c_double(0.0), ctypes.c_int64(0)
    obe._mlib.call("obe_sweep_timing", 0, ctypes.byref(k1_total_ms), ctypes.byref(k1_launches))
    rccl = None
    if use_dist:
        elapsed_local = elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rccl = collective_report(obe, shard, backend, world, rank, ns, args.steps, elapsed_local,
                                 k1_total_ms.value, k1_launches.value, k1_cycles, affinity)

    n_draws = obe.N_DRAWS if obe.utility_method == "variance_approx" else n_p
    evals_per_step = ns * n_draws + n_p
    value = args.steps * evals_per_step / elapsed

    # ---- roofline of the dominant kernel (K1), HIP events on the launch stream ----
    lib = _lib.load()
    n_local = obe._s_end - obe._s_begin
    mom = obe._moments_on_device()
    p, w = obe._pw_tensors()
    ms = ctypes.c_float(0.0)
    stream = obe._stream()
    s_ptr = ctypes.c_void_p(obe._settings_dev.data_ptr() + 8 * obe._s_begin)
    shifted = bool(obe.last_sweep["shifted"])           # the variant the timed cycles ended on
    safe_form = bool(obe.last_sweep.get("safe"))        # ... and the form (fast, or the model's in-range twin)
    flags = (_lib.OBE_SWEEP_SHIFTED if shifted else 0) | (_lib.OBE_SWEEP_SAFE if safe_form else 0)
    obe._mlib.call("obe_sweep_kernel_time", obe._model_struct, s_ptr, ns, n_local, _ptr(p), p.shape[1], n_p,
             _ptr(w), _ptr(mom), flags, _ptr(obe._ws), obe._ws_bytes, 5, ctypes.byref(ms), stream)
    k1_back_to_back_ms = ms.value
    # the figure the roofline uses: the average over the launches of the timed cycles themselves
    # (full-sweep configs; c1's reference-semantics sweep is a one-workgroup kernel timed back to back)
    in_cycle = k1_launches.value > 0 and n_draws == n_p
    if in_cycle:
        ms.value = k1_total_ms.value / k1_launches.value
    k1_s = ms.value * 1e-3
    flop = FLOP_PER_EVAL[model] * n_local * n_p          # K1 timed in full-sweep form
    d = prior.shape[0]
    k1_bytes = 8 * (d + 1) * n_p + 8 * 2 * n_local          # compulsory: cloud + settings + utility
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(cfg, {}).get("sweep_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"kernel": "sweep_kernel (K1)", "bound": "fp64_valu",
                "bound_note": "FP64 vector ALU roof: the sweep has no contraction over a shared operand, so nothing "
                              "goes on MFMA, and it moves ~1e-4 bytes per flop, so HBM is not the limit either; "
                              "78.6 TFLOP/s is also gfx950's dense FP64 MFMA peak",
                "achieved": flop / k1_s / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": flop / k1_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                "flop_per_eval": FLOP_PER_EVAL[model], "evals_per_launch": n_local * n_p,
                "launch_ms": ms.value,
                "launch_timing": ((f"HIP events around each of the {k1_launches.value} sweep-kernel launches of the timed "
                                   "steps, on the launch stream" if instrument_in_timed_steps else
                                   f"HIP events around each of the {k1_launches.value} sweep-kernel launches of {k1_cycles} "
                                   "further cycles right after the timed steps, on the launch stream (N > 1: the timed "
                                   "steps run uninstrumented)") if in_cycle else
                                  "5 back-to-back launches between two HIP events on the launch stream"),
                "launch_ms_back_to_back": k1_back_to_back_ms,
                "variant": "shifted" if shifted else "unshifted", "form": "safe" if safe_form else "fast",
                "kappa": obe.last_sweep["kappa"], "traffic": traffic,
                "traffic_source": (None if traffic is None else
                                   "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an "
                                   "EARLIER run of this command (tools/profile_rocprof.sh), with the guide's gfx950 "
                                   "corrections; read from that committed file, NOT measured in this run"),
                "valu_issue": (lambda slots: {
                    "slots_per_eval": slots, "achieved": slots * n_local * n_p / k1_s, "peak": VALU_ISSUE_PEAK,
                    "unit": "FP64 lane-instructions/s", "frac": slots * n_local * n_p / k1_s / VALU_ISSUE_PEAK,
                    "note": "the flop roofline prices every slot as an FMA; the kernel's mix is ~half mul/add"})(
                        ISSUE_SLOTS_SAFE_FORM.get(model, ISSUE_SLOTS_PER_EVAL[model][1]) if safe_form
                        else ISSUE_SLOTS_PER_EVAL[model][1 if shifted else 0]),
                "hbm_algorithmic": {"bytes": k1_bytes, "achieved": k1_bytes / k1_s / 1e9, "peak": HBM_PEAK_GBS,
                                    "unit": "GB/s", "frac": k1_bytes / k1_s / 1e9 / HBM_PEAK_GBS,
                                    "note": "compute-bound kernel: ~1e4 flop per compulsory byte"}}

    # ---- the HBM-bound update as pdf_update() issues it (K2 + the first moments of the posterior:
    #      2 launches), events around the whole call ----
    timer = ctypes.c_void_p()
    lib.call("obe_timer_create", ctypes.byref(timer))
    wcopy = w.clone()
    st_arr, yy, ss = np.zeros(4), np.zeros(4), np.ones(4) * sigma
    st_arr[0], yy[0] = 3.0, 49500.0
    rows = None
    if not noise_rec:
        rows = np.zeros(16, dtype=np.int32)
        rows[0] = 9
    mom_scratch = torch.zeros_like(obe._moments_dev)

    def update_call(pp, ww):
        obe._mlib.call("obe_bayes_update_model_moments", obe._model_struct, _ptr(pp), pp.shape[1], pp.shape[1], _ptr(ww),
                       _lib.host_ptr(st_arr), _lib.host_ptr(yy), _lib.host_ptr(ss) if rows is None else None,
                       None if rows is None else _lib.host_ptr(rows), 1, float("nan"), _ptr(mom_scratch), _ptr(obe._ws),
                       obe._ws_bytes, None, stream)

    reps, rounds = 50, 9
    upd_ms = ctypes.c_float(0.0)
    round_us = []
    # the call is ~20 us: 50 back-to-back calls per round are enqueued faster than they run, but one
    # host hiccup inside a round (a few ms of preemption) would dominate its average, so the figure is
    # the MEDIAN over the rounds; the first two are warm-up (short kernels on an idle chip clock low)
    for rnd in range(2 + rounds):
        lib.call("obe_timer_start", timer, stream)
        for _ in range(reps):
            update_call(p, wcopy)
        lib.call("obe_timer_stop", timer, stream, ctypes.byref(upd_ms))
        if rnd >= 2:
            round_us.append(upd_ms.value * 1e3 / reps)
        wcopy.copy_(w)                            # every round starts from the same weights
    # the same call on a cloud 16 x larger (the rows tiled): at 1 M particles the three launches are bound by
    # launch latency and the A -> B dependency; this is what the kernels stream when the cloud is large
    big = None
    if world == 1:
        rep16 = 16
        p16 = p.repeat(1, rep16).contiguous()
        w16 = (w / rep16).repeat(rep16).contiguous()
        w16c = w16.clone()
        big_us = []
        for rnd in range(2 + 5):
            lib.call("obe_timer_start", timer, stream)
            for _ in range(10):
                update_call(p16, w16c)
            lib.call("obe_timer_stop", timer, stream, ctypes.byref(upd_ms))
            if rnd >= 2:
                big_us.append(upd_ms.value * 1e3 / 10)
            w16c.copy_(w16)
        big = (p16.shape[1], float(np.median(big_us)))
        del p16, w16, w16c
    lib.call("obe_timer_destroy", timer)
    n_read = obe._device_model.n_read + (0 if noise_rec else 1)
    # pass A: (n_read + 1) rows read, t written; pass B': t and all D rows read, w' written
    k2_bytes = 8 * (n_read + 1) * n_p + 8 * n_p + 8 * (d + 1) * n_p + 8 * n_p
    k2_s = float(np.median(round_us)) * 1e-6
    roofline_update = {"kernel": "update_model_kernel + normalize_moments_kernel (whose last workgroup folds) "
                                 "(K2 + K3 pass 1: the two launches of pdf_update())", "bound": "hbm",
                       "achieved": k2_bytes / k2_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": k2_bytes / k2_s / 1e9 / HBM_PEAK_GBS, "bytes": k2_bytes,
                       "bytes_note": "8(n_read+1)N + 8N (likelihood pass) + 8(D+1)N + 8N (normalisation + first moments)",
                       "call_us": k2_s * 1e6, "call_us_min_max": [min(round_us), max(round_us)],
                       "timing": f"median of {rounds} rounds of {reps} back-to-back calls, HIP events on the launch stream",
                       "traffic": None}
    if big is not None:
        b_bytes = k2_bytes // n_p * big[0]
        roofline_update["large_cloud"] = {"n_particles": big[0], "bytes": b_bytes, "call_us": big[1],
                                          "achieved": b_bytes / (big[1] * 1e-6) / 1e9, "unit": "GB/s",
                                          "frac": b_bytes / (big[1] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                          "note": "the same launches on the cloud tiled 16 x: bandwidth-bound "
                                                  "instead of launch-bound"}

    full_sweep_mode = obe.utility_method == "variance_full"
    out = {"metric": "model-evals/sec (settings x particles) per opt_setting+update cycle, fp64",
           "value": value, "unit": "model-evals/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{cfg}: {desc}", "n_settings": ns, "n_particles": n_p,
                      "utility": ("variance_full (every particle a draw, weighted variance)"
                                  if obe.utility_method == "variance_full" else
                                  f"variance_approx (N_DRAWS = {obe.N_DRAWS} weighted draws, reference semantics)"),
                      "settings_per_rank": n_local, "sharding": f"settings axis / {world}",
                      "resamples_in_timed_steps": resamples,
                      "clock_warm_up_ms_before_the_warmup_steps": warmed_ms,
                      "ms_per_step_min_median_max": [float(np.min(step_ms)), float(np.median(step_ms)), float(np.max(step_ms))],
                      "median_ms_plain_cycle": (float(np.median([m for m, r in zip(step_ms, step_resampled) if not r]))
                                                if resamples < args.steps else None),
                      "median_ms_resample_cycle": (float(np.median([m for m, r in zip(step_ms, step_resampled) if r]))
                                                   if resamples else None)},
           "roofline": roofline, "roofline_update": roofline_update}
    if rccl is not None:
        out["rccl"] = rccl
        if world > 1:
            try:
                out["projection"] = projection_from_one_rank(cfg, world, 1e3 * elapsed / args.steps,
                                                             rccl["combine_us_idle_median"]["max"])
            except Exception as exc:
                out["projection"] = {"available": False, "why": f"{type(exc).__name__}: {exc}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (before the CPU legs: the 128-thread OpenMP run leaves the host busy for a while, and this loop is
        # ~25 us of host time per cycle)
        try:
            out["published_workload"] = published_workload()
        except Exception as exc:
            out["published_workload"] = {"error": str(exc)[:200]}
    if rank == 0 and world == 1 and cfg == "c3" and not args.no_other_configs:
        # every other single-GPU config of BASELINE.json on the same line (outside the timed region above; before
        # the CPU legs, whose 128 OpenMP threads leave the host noisy for a while: c1 is ~25 us of host time per step)
        obe = None
        torch.cuda.empty_cache()
        others = {}
        for name, k, wu in (("c1", 400, 20), ("c2", 40, 5), ("c5", 12, 3)):
            try:
                others[name] = other_config(name, k, wu)
            except Exception as exc:
                others[name] = {"error": str(exc)[:200]}
        out["other_configs"] = others
    out["config"]["cpu_affinity_of_the_gpu_part"] = affinity
    if all_cpus is not None and affinity.get("pinned"):
        try:
            os.sched_setaffinity(0, all_cpus)          # the CPU legs below use every core of the host
        except OSError:
            pass
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, settings, prior, cons, true, sigma)
        # ... and a short leg of the same kind next to every other config's throughput (north_star: the NumPy path
        # "next to" each reported number): ~2 s each on one core; c1 through the oracle class itself
        for name, entry in out.get("other_configs", {}).items():
            if "error" in entry:
                continue
            try:
                entry["cpu_baseline"] = cpu_baseline(name, *make_workload(name), target_s=2.0)
                entry["vs_cpu_1core"] = entry["value"] / entry["cpu_baseline"]["value"]
            except Exception as exc:
                entry["cpu_baseline"] = {"error": str(exc)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and full_sweep_mode:
        try:
            out["cpu_baseline_allcores"] = cpu_baseline_allcores(cfg, settings, prior, cons, true, sigma)
        except Exception as exc:          # no gcc/OpenMP on the box: the 1-core figure stands alone
            out["cpu_baseline_allcores"] = {"error": str(exc)[:200]}
        for name, entry in out.get("other_configs", {}).items():
            if "error" in entry or CONFIGS[name][0] < 1024:          # (c1's reference-semantics cycle has no C restatement)
                continue
            try:
                entry["cpu_baseline_allcores"] = cpu_baseline_allcores(name, *make_workload(name), target_s=1.0)
            except Exception as exc:
                entry["cpu_baseline_allcores"] = {"error": str(exc)[:200]}
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
