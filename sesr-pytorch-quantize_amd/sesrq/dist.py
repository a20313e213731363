"""Frame sharding across the GPUs of one node: one process per GPU, NO data-path collective.

Frames are independent units (the reference itself is batch-1, quan_func.py:349), so a frame batch
is cut into contiguous blocks, one per rank, each rank runs its own engine on its own GPU, and the
only communication is the measurement fence (barrier + MAX of the elapsed time).  backend "nccl" is
RCCL on ROCm; "gloo" is used by the CPU tests."""
from __future__ import annotations

import os
from typing import Optional, Tuple


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard(num_frames: int, world: int, rank: int) -> range:
    """Contiguous block of frame indices owned by `rank`; blocks differ by at most one frame."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(num_frames, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


HOST_SAMPLE = 16      # untimed steps behind the warm-up on which the host's enqueue time per step is measured


def run_timed(group: "Group", step, steps: int, warmup: int, repeats: int = 1, sync=None, units_per_step: float = 1.0, step_many=None):
    """The measurement loop of bench.py, shared with the CPU tests so that on a multi-GPU node RCCL between devices is
    the only thing that has not run before:  W untimed warm-up steps, then `repeats` blocks of EXACTLY `steps` steps,
    each block bracketed by  sync() -> barrier  on both sides (local queue drained first, then the rendezvous, so every
    rank starts its clock with an idle device and stops it when ITS work is done and all ranks have arrived); the
    block's time is the MAX over ranks.  `units_per_step` = units (frames) THIS rank processes per step; the job's
    rate is the sum over ranks of units / the max-over-ranks time.  No data-path collective.
    step_many(n), if given, enqueues n consecutive steps with ONE call (sesrq_forward_many): every phase then hands its steps over in one
    piece -- W, then HOST_SAMPLE, then exactly K per block; step() is not used.

    Returns {"elapsed": [s per block], "units_per_step_total": sum over ranks, "rates": [units/s per block],
    "host_enqueue_s_per_step": host time to enqueue one step, measured on HOST_SAMPLE untimed steps behind the warm-up, queue drained first}."""
    import time
    sync = sync or (lambda: None)

    def fence():
        sync()
        group.barrier()

    # W warm-up steps, untimed.  Then -- still untimed, whatever W is -- HOST_SAMPLE more steps whose ENQUEUE time on the host is read
    # off with the device queue drained first (so the queue's back-pressure does not enter: the loop does not wait); they run on a
    # warm process (round 3 sampled warm-up steps 3..5 under --warmup 5 and read the lazy initialisation: 52.7 us vs 18-19)
    def run(n):
        if n <= 0:
            return
        if step_many is not None:
            step_many(n)
        else:
            for _ in range(n):
                step()

    run(warmup)
    sync()
    tw = time.perf_counter()
    run(HOST_SAMPLE)
    host_s = (time.perf_counter() - tw) / HOST_SAMPLE
    elapsed = []
    for _ in range(max(1, repeats)):
        fence()
        t0 = time.perf_counter()
        run(steps)
        fence()
        elapsed.append(group.max_over_ranks(time.perf_counter() - t0))
    total_units = group.sum_over_ranks(units_per_step)
    return {"elapsed": elapsed, "units_per_step_total": total_units,
            "rates": [steps * total_units / e for e in elapsed], "host_enqueue_s_per_step": host_s, "host_enqueue_sample_steps": HOST_SAMPLE}


def rank_cpus(local_rank: int, local_world: int, cpus=None):
    """The contiguous slice of this process's allowed CPUs that belongs to `local_rank` of `local_world` ranks on the node (slices are
    disjoint and cover the set); fewer CPUs than ranks: everyone keeps the whole set."""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    n = len(cpus)
    if local_world <= 1 or n < local_world:
        return cpus
    return cpus[local_rank * n // local_world:(local_rank + 1) * n // local_world]


def oversubscribed(local_world: int, cpus=None) -> bool:
    """Fewer than four CPUs per rank: busy-waiting threads (the submission pool's workers, the fence's event spin) would take cores from
    each other's enqueue threads -- they must yield instead (SESRQ_SPIN_US=0, os.sched_yield())."""
    n = len(os.sched_getaffinity(0)) if cpus is None else len(cpus)
    return n < 4 * max(1, local_world)


def pin_rank_cpus(local_rank: int, local_world: int):
    """Pin this process (and every thread it starts afterwards: torch's, the library's submission threads) to its slice of the node's CPUs.
    One rank per GPU each spins on its own streams; without a pin, eight ranks' enqueue threads migrate over each other.  Returns the CPUs
    the process may run on afterwards (its slice; the unchanged set where a slice would hold fewer than four CPUs)."""
    mine = rank_cpus(local_rank, local_world)
    if len(mine) < 4:      # a rank runs up to four busy threads (three submission threads + the fence): a smaller slice would stack them; no pin
        return sorted(os.sched_getaffinity(0))
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        pass
    return mine


class Group:
    """Thin wrapper over torch.distributed used by bench.py; a no-op for world_size 1.

    Round 5: the measurement fence cannot be lost to RCCL.  The process group is ALWAYS gloo first (host-only: created before the process has
    touched the GPU); backend "nccl" (= RCCL) is then TRIED as a second group once the device is known (bind_device) -- communicator, one
    all-reduce, agreement of all ranks over the gloo group -- and used for the fence only if every rank succeeded.  Otherwise the cause is
    printed once, the gloo fence stays, and `fence` says so ("gloo (nccl: <error>)"): the data path never crosses devices, so a scaling run
    must not end with exit 3 because a collective library could not start (rounds 3-4 did)."""

    def __init__(self, backend: Optional[str] = None, device=None, timeout_s: Optional[float] = None, force: bool = False):
        """force: build the process group at world size 1 too (the one-GPU box's way to run the RCCL fence: communicator creation,
        barrier, MAX / SUM all-reduce on the device -- tests/test_00_bench_spawn.py)"""
        self.rank, self.local_rank, self.world = env_world()
        self.dist = None
        self.pg = None              # the group the fence runs on (None = the default gloo group)
        self.device = None          # device of the reduction tensors (None = host)
        self.fence = "none (one process)"
        self.want = backend or "nccl"
        self.timeout_s = timeout_s
        if self.world > 1 or force:
            import torch.distributed as dist
            kw = {}
            if timeout_s is not None:
                import datetime
                kw["timeout"] = datetime.timedelta(seconds=timeout_s)
            try:
                dist.init_process_group("gloo", **kw)
            except Exception as exc:      # no rendezvous at all (MASTER_ADDR / port): nothing to fall back to
                import sys
                print(f"sesrq.dist: init_process_group('gloo', rank {self.rank} of {self.world}, "
                      f"MASTER_ADDR={os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}) failed: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
                raise SystemExit(3)
            self.dist = dist
            self.fence = "gloo"
            if device is not None:
                self.bind_device(device)

    def bind_device(self, device, probe_timeout_s: float = 120.0):
        """Try the device backend (RCCL) for the fence now that the rank's device is set.  Never raises, never exits: on any failure on any
        rank every rank keeps the gloo fence."""
        if self.dist is None or self.want != "nccl":
            return self.fence
        import datetime
        import torch
        dist = self.dist
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")      # a failed probe must not take the process down
        err, pg = None, None
        try:
            if os.environ.get("SESRQ_FORCE_NCCL_FAILURE"):                   # test hook: tests/test_dist_gloo.py
                raise RuntimeError("SESRQ_FORCE_NCCL_FAILURE is set")
            pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=self.timeout_s or probe_timeout_s), device_id=device)
            t = torch.ones(1, dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=pg)
            if int(t.item()) != dist.get_world_size():
                raise RuntimeError(f"all-reduce over RCCL returned {t.item()} for {dist.get_world_size()} ranks")
        except Exception as exc:
            err = f"{type(exc).__name__}: {str(exc).splitlines()[0][:200] if str(exc) else ''}"
        # agreement over the group that works: RCCL is used only if EVERY rank got through
        ok = torch.tensor([0.0 if err else 1.0], dtype=torch.float64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 1.0:
            self.pg, self.device, self.fence = pg, device, "nccl"
        else:
            import sys
            why = err or "another rank failed"
            print(f"sesrq.dist: rank {self.rank}: the RCCL fence is not available ({why}); keeping the gloo fence -- the data path needs no "
                  "collective (frames shard, bundle replicated), only the timing fence does", file=sys.stderr, flush=True)
            self.fence = f"gloo (nccl: {why})"
        return self.fence

    def barrier(self):
        if self.dist is not None:
            if self.pg is not None:
                self.dist.barrier(group=self.pg)
            else:
                self.dist.barrier()

    def _reduce(self, value: float, op) -> float:
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        if self.pg is not None:
            self.dist.all_reduce(t, op=op, group=self.pg)
        else:
            self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max_over_ranks(self, value: float) -> float:
        return float(value) if self.dist is None else self._reduce(value, self.dist.ReduceOp.MAX)

    def sum_over_ranks(self, value: float) -> float:
        return float(value) if self.dist is None else self._reduce(value, self.dist.ReduceOp.SUM)

    def close(self):
        if self.dist is not None:
            self.barrier()
            self.dist.destroy_process_group()
            self.dist = None
