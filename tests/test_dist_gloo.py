"""N>1 path on CPU: world_size-2 gloo processes exercise the frame sharding and the measurement
fence that bench.py uses on the GPUs (no data-path collective exists to test)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from sesrq.dist import shard


def test_shard_partitions_exactly():
    for n in (0, 1, 7, 8, 32, 33, 1000):
        for world in (1, 2, 3, 4, 8):
            parts = [shard(n, world, r) for r in range(world)]
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard(4, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    import time
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from sesrq.dist import Group, run_timed, shard as sh
    g = Group(backend="gloo")
    mine = sh(11, g.world, g.rank)
    # a "step" = this rank's frames through a stand-in per-frame function (the engine is the only thing stubbed:
    # bench.py passes Engine.forward here and torch.cuda.synchronize as `sync`)
    log = []

    def step():
        for f in mine:
            log.append(f)
        time.sleep(0.02 * (1 + rank))            # rank 1 is the slow one

    synced = []
    res = run_timed(g, step, steps=3, warmup=2, repeats=2, sync=lambda: synced.append(len(log)), units_per_step=len(mine))
    elapsed = g.max_over_ranks(0.5 + rank)
    q.put((rank, list(mine), elapsed, res, len(log), synced))
    g.close()


def test_two_rank_gloo_fence_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] + res[1][1] == list(range(11))          # uneven shards (6 + 5), every frame exactly once
    assert all(abs(r[2] - 1.5) < 1e-12 for r in res), "MAX over ranks"
    for rank, mine, _, timed, nlog, synced in res:
        assert timed["units_per_step_total"] == 11, "sum over ranks of the frames per step"
        assert len(timed["elapsed"]) == 2 and nlog == len(mine) * (2 + 2 * 3), "W warm-up + repeats x K steps, exactly"
        # both ranks report the SLOW rank's block time (3 steps x 40 ms), not their own
        assert all(e >= 3 * 0.04 * 0.9 for e in timed["elapsed"]), timed
        assert all(abs(a - b) < 1e-9 for a, b in zip(res[0][3]["elapsed"], res[1][3]["elapsed"]))
        assert timed["rates"] == [3 * 11 / e for e in timed["elapsed"]]
        # the local drain (sync) runs on both sides of every block: before the start barrier and after the last step
        assert synced == [len(mine) * 2, len(mine) * 5, len(mine) * 5, len(mine) * 8]
