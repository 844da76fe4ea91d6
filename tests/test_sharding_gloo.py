"""N>1 path on CPU: two ``gloo`` ranks shard a global lane batch, build their synthetic inputs and reduce their timings
exactly as ``bench.py`` does on RCCL.  The data path needs no collective; what is checked is that the shards form an
exact disjoint cover of the global batch, that the per-lane inputs do not depend on the sharding, and that the
max / sum reductions used for the JSON line are right."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from grid2op_amd.sharding import lane_range, max_over_ranks, sum_over_ranks, synthetic_lane_inputs  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, T, n_load, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lane0, n = lane_range(total, world, rank)
    off, sc = synthetic_lane_inputs(n_load, T, lane0 + np.arange(n))
    # gather the shards (test only -- the product path never gathers)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lane0, n, off, sc))
    tmax = max_over_ranks(0.5 + rank, dist)
    tsum = sum_over_ranks(float(n), dist)
    dist.barrier()
    if rank == 0:
        q.put((gathered, tmax, tsum))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4096, 37])
def test_two_rank_sharding_covers_the_batch(total):
    world, T, n_load = 2, 576, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, T, n_load, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, tmax, tsum = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lanes = np.concatenate([g[0] + np.arange(g[1]) for g in gathered])
    assert np.array_equal(lanes, np.arange(total))                      # exact disjoint cover, contiguous blocks
    off = np.concatenate([g[2] for g in gathered])
    sc = np.concatenate([g[3] for g in gathered])
    ref_off, ref_sc = synthetic_lane_inputs(n_load, T, np.arange(total))  # sharding-independent inputs
    assert np.array_equal(off, ref_off) and np.array_equal(sc, ref_sc)
    assert tmax == 1.5 and tsum == float(total)


def test_lane_range_properties():
    for total in (1, 7, 64, 4096, 61440):
        for world in (1, 2, 3, 8):
            blocks = [lane_range(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and sum(n for _, n in blocks) == total
            for (a0, an), (b0, _) in zip(blocks, blocks[1:]):
                assert a0 + an == b0
            assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1
