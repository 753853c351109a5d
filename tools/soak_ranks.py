"""Randomised experiments through a settings-sharded object in REAL processes (developer aid, GPU):

    python tools/soak_ranks.py [minutes=2] [seed=0] [world=2] [max_recipes=0 (no limit)]

`world` processes share this box's one GPU, torch.distributed backend gloo (RCCL wants one GPU per rank; the
collective calls are the same).  Every rank draws the same recipes from the seed: grids whose split over the
ranks is uneven and straddles the settings-per-lane thresholds of the sweep kernel, peaks of ordinary width and
peaks 1e12 / 1e40 times narrower than the grid (the fast sweep forms leave their range), full sweeps,
reference-semantics sweeps and y-space utilities, cost hooks, the three speculation modes, opt_setting / good_setting / utility() in any order (SOAK_SWEEPER=0.08: a share of sweeper-composition experiments mixed in — see
DESIGN.md section 4 for what that mix found), resamples forced and triggered,
set_pdf, and READS of the cloud or its moments done by one rank only (a script that logs on rank 0).  What is
checked: nobody hangs (a mismatch in the number of collectives is an error after 60 s), every rank logs the same
settings, forms and resample decisions cycle by cycle, the replicas stay identical (check_replicas), and the
settings are the unsharded run's (computed by every rank for itself; a tie broken differently by the 1e-13 that
the chunk order can move a utility is counted, not failed)."""
import datetime
import os
import socket
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def recipe(g):
    k = int(g.choice([1, 1, 2, 3, 7]))
    return dict(
        k=k, n=int(g.choice([700, 3000, 9000, 30000])),
        ns=int(g.choice([45, 301, 511, 1023, 1025, 2047, 2049, 4097, 5003, int(g.integers(40, 6000))])),
        d=float(g.choice([0.05, 0.05, 0.05, 2.5e-12, 2.5e-40])),
        method=str(g.choice(["variance_full"] * 13 + ["variance_approx"] * 5 + ["max_min", "pseudo_utility"])),
        cost=int(g.integers(0, 8)), speculate=[True, False, "auto", "auto"][int(g.integers(0, 4))],
        sweeper=bool(g.random() < float(os.environ.get("SOAK_SWEEPER", "0"))),
        noise=bool(g.random() < 0.3), threshold=float(g.choice([0.1, 0.5, 0.9])),
        cycles=int(g.integers(5, 14)), seed=int(g.integers(1 << 30)),
        acts=g.integers(0, 10, 16).tolist(), reads=g.integers(0, 6, 16).tolist(), readers=g.integers(0, 8, 16).tolist())


def run_sweeper(obe, r, shard, rank):
    """The sweeper composition (demos/sweeper): start/stop pairs chosen from the point utilities that every rank
    gathers, a whole sweep of points per update; its module-level generator is seeded alike everywhere (rank 0's
    draws are the ones used)."""
    from optbayesexpt_amd import sweeper
    g = np.random.default_rng(r["seed"])
    n = min(r["n"], 9000)
    prior = np.array([g.uniform(2, 4, n), g.uniform(400, 2000, n), g.normal(500, 1000, n), g.exponential(500, n) + 1.0])
    x = np.linspace(1.5, 4.5, max(40, min(r["ns"], 600)))
    o = obe.OptBayesExptSweeper(obe.models.lorentzian(), (x,), prior, (0.1,), 3, scale=False,
                                utility_method="variance_full", settings_shard=shard)
    o.rng = np.random.default_rng(r["seed"] + 1)
    sweeper.rng = np.random.default_rng(r["seed"] + 3)
    sim = np.random.default_rng(r["seed"] + 2)
    log = []
    try:
        for c in range(min(r["cycles"], 6)):
            pair = o.good_setting() if r["acts"][c] in (2, 3) else o.opt_setting()
            log.append((int(pair[0]) * 100000 + int(pair[1]), False, True))
            xs = x[pair[0]:pair[1]]
            ys = 300.0 + 1200.0 / (((xs - 3.1) / 0.1) ** 2 + 1) + 800.0 * sim.standard_normal(len(xs))
            o.pdf_update(((xs,), ys))
            log.append(bool(o.just_resampled))
            if shard is None or r["readers"][c] % shard.world_size == rank:
                o.mean(), o.std()
        if shard is not None:
            assert o.check_replicas()
        return log, np.asarray(o.mean())
    except (ValueError, np.linalg.LinAlgError) as exc:
        exc.partial_log = log
        raise


def run(obe, r, shard, rank):
    try:
        return _run(obe, r, shard, rank)
    finally:
        if os.environ.get("SOAK_DRAIN"):            # (diagnostic: no device work outlives the object that enqueued it)
            import torch
            torch.cuda.synchronize()


def _run(obe, r, shard, rank):
    if r.get("sweeper"):
        return run_sweeper(obe, r, shard, rank)
    g = np.random.default_rng(r["seed"])
    k, n = r["k"], r["n"]
    rows = [g.uniform(2, 4, (k, n)), g.uniform(400, 2000, (1, n)), g.normal(500, 1000, (1, n))]
    if r["noise"]:
        rows.append(g.exponential(500, (1, n)) + 1.0)
    prior = np.vstack(rows)
    sv = (np.linspace(1.5, 4.5, r["ns"]),)
    sv[0][::37] = prior[0, :sv[0][::37].size]            # some settings ON a particle's peak
    cls = obe.OptBayesExptNoiseParameter if r["noise"] else obe.OptBayesExpt
    kw = dict(utility_method=r["method"]) if r["method"] != "variance_approx" else dict(n_draws=30)
    full = r["method"] == "variance_full"
    if r["noise"]:
        kw["noise_parameter_index"] = k + 2
    o = cls(obe.models.lorentzian(k), sv, prior.copy(), (r["d"],), scale=False, default_noise_std=500.0,
            settings_shard=shard, resample_threshold=r["threshold"], **kw)
    o.tuning_parameters["speculative_sweep"] = r["speculate"]
    if r["cost"] == 0:                                   # a cost hook: per-setting values / a scalar
        per_setting = 1.0 + np.linspace(0.0, 1.0, r["ns"]) ** 2
        o.cost_estimate = lambda: per_setting
    elif r["cost"] == 1:
        o.cost_estimate = lambda: 2.5
    o.rng = np.random.default_rng(r["seed"] + 1)
    sim = np.random.default_rng(r["seed"] + 2)
    true = prior[:, 0]
    log = []
    try:
        return _cycles(obe, o, r, g, sim, prior, true, full, shard, rank, log)
    except (ValueError, np.linalg.LinAlgError) as exc:
        exc.partial_log = log
        raise


def _cycles(obe, o, r, g, sim, prior, true, full, shard, rank, log):
    n = r["n"]
    for c in range(r["cycles"]):
        act, read, reader = r["acts"][c], r["reads"][c], r["readers"][c]
        if act == 0:
            o.resample()
        elif act == 1:
            w = g.exponential(1.0, n)
            o.set_pdf(prior[:, ::-1].copy(), w / w.sum())
        if act in (2, 3):
            x = o.good_setting(pickiness=int(g.integers(1, 12)))
        elif act == 4:
            u = np.asarray(o.utility())
            assert u.shape == (r["ns"],)
            x = o.opt_setting()
        else:
            x = o.opt_setting()
        sweep = dict(o.last_sweep) if full else {}
        log.append((int(o.last_setting_index), bool(sweep.get("safe", False)), bool(sweep.get("shifted", True))))
        y = float(np.atleast_1d(o.model_function(x, true, (max(r["d"], 0.05),)))[0]) + 500.0 * sim.standard_normal()
        o.pdf_update((x, y) if r["noise"] else (x, y, 500.0))
        log.append(bool(o.just_resampled))
        if shard is None or reader % shard.world_size == rank:       # (the unsharded run: the reads of rank 0's script)
            if read == 0:
                assert np.isfinite(np.asarray(o.particles)).all()
            elif read == 1:
                o.mean(), o.std()
            elif read == 2:
                o.covariance()
            elif read == 3:
                np.asarray(o.particle_weights).sum()
    if shard is not None:
        assert o.check_replicas()
    return log, np.asarray(o.mean())


def worker(rank, world, port, minutes, seed, max_recipes, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    warnings.simplefilter("ignore")
    import optbayesexpt_amd as obe
    if os.environ.get("SOAK_TRACE"):
        # every collective of this rank, in order, one line each (flushed): after a mismatch the two ranks' files
        # show which call had no partner
        fh = open(os.path.join(os.environ["SOAK_TRACE"], f"coll{rank}.log"), "w")

        def traced(name):
            inner = getattr(dist, name)

            def call(*a, **kw):
                what = [tuple(x.shape) if hasattr(x, "shape") else type(x).__name__ for x in a[:2]]
                fh.write(f"{name} {what}\n")
                fh.flush()
                return inner(*a, **kw)
            setattr(dist, name, call)
        for name in ("broadcast", "all_gather_into_tensor", "broadcast_object_list", "all_gather_object", "all_gather"):
            traced(name)
    g = np.random.default_rng(seed)
    t_end = time.time() + 60 * minutes
    done = cycles = ties = safe = refused = resampled = 0
    try:
        while True:
            # (rank 0's clock decides when to stop: every rank must leave the loop in the same iteration)
            go = [time.time() < t_end and not (max_recipes and done >= max_recipes)]
            dist.broadcast_object_list(go, src=0)
            if not go[0]:
                break
            r = recipe(g)
            if done < int(os.environ.get("SOAK_SKIP", "0")):      # (replay aid: fast-forward the recipe stream)
                done += 1
                continue
            trace = os.environ.get("SOAK_TRACE")       # a directory: one line per recipe and rank (to find a divergence)
            if trace:
                with open(os.path.join(trace, f"rank{rank}.log"), "a") as fh2:
                    fh2.write(f"{done} begin {r}\n")
                with open(os.path.join(trace, f"coll{rank}.log"), "a") as fh2:
                    fh2.write(f"--- recipe {done}\n")
            try:
                mine = run(obe, r, obe.SettingsShard(), rank)
                err = None
            except (ValueError, np.linalg.LinAlgError) as exc:      # (numpy's own refusals: every rank alike)
                mine, err = None, (f"{type(exc).__name__}: {exc}"[:80], exc.partial_log)
                if trace:
                    import traceback
                    with open(os.path.join(trace, f"rank{rank}.log"), "a") as fh2:
                        fh2.write(f"{done} refused {err}\n{traceback.format_exc()}\n")
            try:
                ref = run(obe, r, None, 0)
                ref_err = None
            except (ValueError, np.linalg.LinAlgError) as exc:
                ref, ref_err = None, (f"{type(exc).__name__}: {exc}"[:80], exc.partial_log)
            everyone = [None] * world
            dist.all_gather_object(everyone, (mine, err))
            for other in everyone[1:]:
                assert other[1] == everyone[0][1], (r, everyone[0][1], other[1])
                if mine is not None:
                    assert other[0][0] == everyone[0][0][0], (r, everyone[0][0][0], other[0][0])
                    np.testing.assert_array_equal(other[0][1], everyone[0][0][1])
            strip = lambda lg: [e[0] if isinstance(e, tuple) else e for e in lg]       # noqa: E731
            if (err is None) != (ref_err is None):
                # refused on one side only: legitimate only if the two runs had parted at a tie before (then they
                # are different experiments from there on)
                a = strip(mine[0] if mine is not None else err[1])
                b = strip(ref[0] if ref is not None else ref_err[1])
                m = min(len(a), len(b))
                assert a[:m] != b[:m], (r, err and err[0], ref_err and ref_err[0], a, b)
                first = next(i for i in range(m) if a[i] != b[i])
                assert first % 2 == 0, (r, first, a, b)
                ties += 1
                done += 1
                continue
            if mine is not None:
                # against the unsharded run: settings and resample decisions (the FORM may differ — a lane of the
                # whole grid owns more settings than a lane of a slice, so its fast form leaves its range earlier)
                a, b = strip(mine[0]), strip(ref[0])
                if a != b:
                    first = next(i for i, (u, v) in enumerate(zip(a, b)) if u != v)
                    # a different setting: only as a tie (or a draw at the edge of a CDF step) moved by the
                    # partial-sum order; anything systematic shows as many of these
                    assert first % 2 == 0, (r, first, a, b)
                    ties += 1
                else:
                    np.testing.assert_allclose(mine[1], ref[1], rtol=1e-9)
                cycles += r["cycles"]
                safe += any(e[1] for e in mine[0] if isinstance(e, tuple))
                resampled += any(e for e in mine[0] if e is True)
            else:
                refused += 1
            done += 1
    finally:
        dist.destroy_process_group()
    ret[rank] = (done, cycles, ties, safe, refused, resampled)


def main():
    import torch.multiprocessing as mp
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    max_recipes = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(world, port, minutes, seed, max_recipes, ret), nprocs=world, join=True)
    done, cycles, ties, safe, refused, resampled = ret[0]
    assert all(ret[r][0] == done for r in range(world))
    print(f"ranks soak: {world} ranks, {done} experiments, {cycles} cycles, every rank the same log "
          f"({safe} experiments with sweeps in the safe form, {resampled} with resamples, {refused} that numpy refuses on "
          f"every rank alike); {ties} left the unsharded run's settings at a tie")


if __name__ == "__main__":
    main()
