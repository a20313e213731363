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
    from sesrq.dist import Group, pin_rank_cpus, run_timed, shard as sh
    allowed = sorted(os.sched_getaffinity(0))
    cpus = pin_rank_cpus(rank, world)                 # bench.py's first act in a rank
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
    q.put((rank, list(mine), elapsed, res, len(log), synced, cpus, sorted(os.sched_getaffinity(0)), allowed))
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
    # round 5: every rank lives on its own slice of the CPUs (live affinities, not just the arithmetic): disjoint and covering
    allowed = res[0][8]
    if len(allowed) >= 8:
        assert all(r[6] == r[7] for r in res), "the affinity the kernel reports is the slice"
        assert not set(res[0][7]) & set(res[1][7]) and sorted(res[0][7] + res[1][7]) == allowed, (res[0][7], res[1][7])
    for rank, mine, _, timed, nlog, synced, *_aff in res:
        assert timed["units_per_step_total"] == 11, "sum over ranks of the frames per step"
        hs = timed["host_enqueue_sample_steps"]
        assert hs >= 16 and timed["host_enqueue_s_per_step"] >= 0.02 * 0.9
        assert len(timed["elapsed"]) == 2 and nlog == len(mine) * (2 + hs + 2 * 3), "W warm-up + the untimed host-enqueue sample + repeats x K steps, exactly"
        # both ranks report the SLOW rank's block time (3 steps x 40 ms), not their own
        assert all(e >= 3 * 0.04 * 0.9 for e in timed["elapsed"]), timed
        assert all(abs(a - b) < 1e-9 for a, b in zip(res[0][3]["elapsed"], res[1][3]["elapsed"]))
        assert timed["rates"] == [3 * 11 / e for e in timed["elapsed"]]
        # the local drain (sync) runs on both sides of every block: before the start barrier and after the last step
        u = len(mine) * (2 + hs)
        assert synced == [len(mine) * 2, u, u + len(mine) * 3, u + len(mine) * 3, u + len(mine) * 6]


def _worker8(rank, world, port, q):
    """config 4's split on 8 ranks: 32 frames -> 4 per rank, the whole measurement loop, rank 0 builds the one JSON line"""
    import json
    import time
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from sesrq.dist import Group, pin_rank_cpus, run_timed, shard as sh
    allowed = sorted(os.sched_getaffinity(0))
    cpus = pin_rank_cpus(rank, world)                 # as bench.py does first thing in a rank (LOCAL_RANK / LOCAL_WORLD_SIZE)
    assert sorted(os.sched_getaffinity(0)) == cpus     # a slice of >= 4 CPUs, or (fewer CPUs per rank than a rank's busy threads) no pin at all
    g = Group(backend="gloo")
    mine = sh(32, g.world, g.rank)
    done = []

    def step():
        done.extend(mine)
        time.sleep(0.002 * (1 + (rank == 5)))       # rank 5 is the straggler

    res = run_timed(g, step, steps=4, warmup=1, repeats=2, sync=lambda: None, units_per_step=len(mine))
    line = None
    if g.rank == 0:
        el = sorted(res["elapsed"])[len(res["elapsed"]) // 2]
        line = json.dumps({"value": 4 * res["units_per_step_total"] / el, "n_gpus": g.world, "frames_per_step": res["units_per_step_total"],
                           "scaling": "strong"})
    q.put((rank, list(mine), res["elapsed"], res["units_per_step_total"], line, cpus, allowed))
    g.close()


def test_eight_rank_gloo_config4_split():
    """BASELINE config 4's sharding rehearsed at its real rank count on CPU (the 8-GPU node is the driver's): shard(32, 8, r) gives every
    rank 4 contiguous frames, the block time every rank reports is the straggler's (MAX), the job's frames per step is the SUM, and
    only rank 0 prints.  Hardware scaling stays unmeasured until the driver's 8-GPU run."""
    import json
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [list(range(4 * k, 4 * k + 4)) for k in range(8)]
    assert all(r[3] == 32 for r in res)
    for r in res[1:]:
        assert all(abs(a - b) < 1e-9 for a, b in zip(r[2], res[0][2])), "every rank reports the same (max-over-ranks) block time"
        assert r[4] is None
    assert all(e >= 4 * 0.004 * 0.9 for e in res[0][2]), "the straggler's time"
    line = json.loads(res[0][4])
    assert line["n_gpus"] == 8 and line["frames_per_step"] == 32 and line["value"] > 0
    # round 5: per-rank CPU affinity -- the ranks' slices are disjoint and cover the CPUs the job may use (one slice each where CPUs >= ranks)
    allowed = res[0][6]
    slices = [r[5] for r in res]
    if len(allowed) >= 32:
        assert sorted(c for sl in slices for c in sl) == allowed, slices
        assert all(len(sl) >= len(allowed) // 8 for sl in slices)
    else:      # fewer than four CPUs per rank (this container: 8 CPUs for 8 ranks): nobody is pinned, everybody yields
        assert all(sl == allowed for sl in slices)
    from sesrq.dist import rank_cpus
    big = list(range(64))      # the slices themselves, on a node-sized CPU set: disjoint, covering, 8 each
    sl8 = [rank_cpus(r, 8, big) for r in range(8)]
    assert sorted(c for s_ in sl8 for c in s_) == big and all(len(s_) == 8 for s_ in sl8)


def test_rendezvous_failure_is_a_clean_exit():
    """No rendezvous at all (rank 1 never starts): the rank prints what failed and exits non-zero -- no hang, no re-exec.  (Round 5: the group
    that can fail this way is the gloo one; a failing RCCL no longer ends the run, see the next test.)"""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', "
            "MASTER_PORT='%d'); from sesrq.dist import Group; Group(backend='nccl', device=None, timeout_s=10)") % (
                os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sesr-pytorch-quantize_amd"), _free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    assert "sesrq.dist: init_process_group('gloo'" in r.stderr


def _worker_nccl_fails(rank, world, port, q, force):
    import json
    import time
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if force and rank == 1:
        os.environ["SESRQ_FORCE_NCCL_FAILURE"] = "1"          # ONE rank fails: every rank must fall back
    from sesrq.dist import Group, run_timed
    g = Group(backend="nccl")                                  # as bench.py: gloo group first, no device yet
    assert g.fence == "gloo"
    fence = g.bind_device(torch.device("cuda:0"))              # RCCL cannot start (this test runs without a usable second device)
    res = run_timed(g, lambda: time.sleep(0.001), steps=3, warmup=1, repeats=2, units_per_step=1)
    line = json.dumps({"value": 3 * res["units_per_step_total"] / res["elapsed"][0], "n_gpus": g.world, "fence": g.fence}) if g.rank == 0 else None
    q.put((rank, fence, res["units_per_step_total"], line))
    g.close()


@pytest.mark.parametrize("force", [False, True], ids=["no-device", "one-rank-fails"])
def test_failing_nccl_keeps_the_gloo_fence_and_one_json_line(force):
    """Round 5: a scaling run must not be lost to RCCL.  Two ranks ask for backend "nccl"; the communicator cannot be
    created (no device in this container / one rank is made to fail): every rank prints the cause, keeps the gloo fence it already has,
    the measurement loop runs, and rank 0 prints ONE JSON line whose `fence` says what happened."""
    import json
    if torch.cuda.is_available() and not force:
        pytest.skip("a HIP device is present: the un-forced failure needs a box without one")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_nccl_fails, args=(r, world, port, q, force)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1].startswith("gloo (nccl: ") for r in res), res
    assert all(r[2] == 2 for r in res)
    lines = [r[3] for r in res if r[3]]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["fence"].startswith("gloo (nccl: ")


def test_rank_cpu_slices_and_oversubscription():
    from sesrq.dist import oversubscribed, rank_cpus
    cpus = list(range(10, 42))
    sl = [rank_cpus(r, 8, cpus) for r in range(8)]
    assert [c for s_ in sl for c in s_] == cpus and all(len(s_) == 4 for s_ in sl)
    assert [rank_cpus(r, 3, range(8)) for r in range(3)] == [[0, 1], [2, 3, 4], [5, 6, 7]]
    assert rank_cpus(2, 8, range(4)) == [0, 1, 2, 3]                  # fewer CPUs than ranks: no pin
    assert not oversubscribed(8, range(32)) and oversubscribed(8, range(31)) and oversubscribed(1, range(3)) and not oversubscribed(1, range(4))


def test_run_timed_hands_whole_phases_to_step_many():
    """--submit many: every phase of the measurement loop reaches the library in ONE piece -- W warm-up steps, the untimed host-enqueue
    sample, then exactly K steps per timed block (world size 1: no process group)."""
    from sesrq.dist import Group, run_timed, HOST_SAMPLE
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    calls, synced = [], []
    res = run_timed(Group(backend="gloo"), step=lambda: calls.append("step"), steps=7, warmup=3, repeats=4, sync=lambda: synced.append(len(calls)),
                    units_per_step=2, step_many=lambda n: calls.append(n))
    assert calls == [3, HOST_SAMPLE] + [7] * 4, calls
    assert len(res["elapsed"]) == 4 and res["units_per_step_total"] == 2 and res["host_enqueue_sample_steps"] == HOST_SAMPLE
    assert res["rates"] == [7 * 2 / e for e in res["elapsed"]]
    calls.clear()
    run_timed(Group(backend="gloo"), step=lambda: calls.append("step"), steps=2, warmup=0, repeats=1, step_many=lambda n: calls.append(n))
    assert calls == [HOST_SAMPLE, 2]                      # W = 0: no empty hand-over
