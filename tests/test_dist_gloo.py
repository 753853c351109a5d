"""The N > 1 path on CPU: two processes, torch.distributed backend "gloo".  Each rank owns
its contiguous slice of the settings (optbayesexpt_amd.dist), computes that slice's
utility with the oracle, and the collectives must reproduce the single-process answer:
global first-max (np.argmax tie/NaN rules) and the gathered utility vector."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import models as omodels
from optbayesexpt_amd.dist import SettingsShard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _utility_cases():
    g = np.random.default_rng(77)
    n = 1500
    prior = np.array([g.uniform(2, 4, n), g.uniform(-2000, -400, n), g.normal(50000, 1000, n)])
    w = g.exponential(1.0, n)
    w /= w.sum()
    cases = []
    for ns in (201, 64, 5):
        sv = (np.linspace(1.5, 4.5, ns),)
        yvar = oracle.yvar_full_sweep(omodels.lorentzian, oracle.flatten_settings(sv), prior, w, (0.1,))
        cases.append(oracle.utility_from_yvar(yvar, 250000.0, 1.0))
    ties = np.zeros(40)
    ties[[3, 17, 25, 39]] = 2.0              # equal maxima on both ranks: lowest index wins
    cases.append(ties)
    nan = np.arange(30, dtype=float)
    nan[22] = np.nan                         # NaN beats everything (np.argmax)
    cases.append(nan)
    cases.append(np.full(9, -np.inf))
    return cases


class _ShardedStub:
    """The generator logic of OptBayesExpt on a sharded settings axis, without a GPU: the
    methods themselves, bound to an object that has only the attributes they touch."""
    from optbayesexpt_amd.obe_base import OptBayesExpt as _cls
    _sync_rng = _cls._sync_rng
    _device = "cpu"

    def __init__(self, shard, rng):
        self._shard, self._rng = shard, rng


def _adopt_rank0_generator(shard, rng):
    stub = _ShardedStub(shard, rng)
    stub._sync_rng()
    return stub._rng.random(5), type(stub._rng.bit_generator).__name__


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = SettingsShard()
        assert (shard.rank, shard.world_size) == (rank, world)
        results = []
        for u in _utility_cases():
            b, e = shard.bounds(u.size)
            local = u[b:e]
            if e > b:
                k = int(np.argmax(local))
                rec = SettingsShard.make_record(float(local[k]), k, kappa=float(rank))
            else:
                rec = SettingsShard.make_record(-np.inf, np.iinfo(np.int64).max // 2)
            best_val, best_idx, kappa = shard.combine_records(rec, u.size)
            assert kappa == float(world - 1) or u.size < world      # worst kappa over all ranks, same everywhere
            rows = np.zeros((2, max(e - b, 1)))
            rows[0, :e - b] = local
            rows[1, :e - b] = 2 * local
            full = shard.gather_rows(torch.from_numpy(rows), u.size)
            results.append((best_idx, best_val, full))
        # draws that must be taken once for the whole job: rank 0's values everywhere
        mine = np.random.default_rng(100 + rank)
        f = shard.broadcast_from_rank0(mine.normal(0, 1, 7))
        i = shard.broadcast_from_rank0(np.atleast_1d(mine.choice(np.arange(1000))))
        ret[rank] = results
        ret[f"bcast{rank}"] = (f, i)
        # the generator of a sharded object: every rank adopts rank 0's (unseeded) state
        ret[f"rng{rank}"] = _adopt_rank0_generator(shard, np.random.default_rng())
        ret[f"rng_mt{rank}"] = _adopt_rank0_generator(
            shard, np.random.Generator(np.random.MT19937(5)) if rank == 0 else np.random.default_rng())
        table = shard.all_gather_int64(np.array([7, rank, -rank]))
        assert table.shape == (world, 3) and table[:, 1].tolist() == list(range(world))
        # bench.py's instrumentation of the arg-max combine (the "rccl" block of an N > 1 line): one entry per combine
        shard.timing = []
        for k in range(3):
            shard.combine_records(SettingsShard.make_record(float(rank + k), 5, kappa=1.0), 1000)
        assert len(shard.timing) == 3 and all(a > 0.0 and b > 0.0 for a, b in shard.timing)
        shard.timing = None
        shard.combine_records(SettingsShard.make_record(1.0, 0), 1000)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_argmax_and_gather_match_single_process(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    cases = _utility_cases()
    for rank in range(world):
        for (best_idx, best_val, full), u in zip(ret[rank], cases):
            assert best_idx == int(np.argmax(u)), (rank, u.size)
            np.testing.assert_array_equal(best_val, u[int(np.argmax(u))])
            np.testing.assert_array_equal(full[0], u)
            np.testing.assert_array_equal(full[1], 2 * u)
        rank0 = np.random.default_rng(100)
        np.testing.assert_array_equal(ret[f"bcast{rank}"][0], rank0.normal(0, 1, 7))
        got = ret[f"bcast{rank}"][1]
        assert got.dtype == np.int64 and got[0] == rank0.choice(np.arange(1000))
        # unseeded generators: all ranks continue rank 0's stream; a different bit generator is adopted too
        np.testing.assert_array_equal(ret[f"rng{rank}"][0], ret["rng0"][0])
        np.testing.assert_array_equal(ret[f"rng_mt{rank}"][0], np.random.Generator(np.random.MT19937(5)).random(5))
        assert ret[f"rng_mt{rank}"][1] == "MT19937"
