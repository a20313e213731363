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


class Group:
    """Thin wrapper over torch.distributed used by bench.py; a no-op for world_size 1."""

    def __init__(self, backend: Optional[str] = None, device=None, timeout_s: Optional[float] = None, force: bool = False):
        """force: build the process group at world size 1 too (the one-GPU box's way to run the RCCL fence: communicator creation,
        barrier, MAX / SUM all-reduce on the device -- tests/test_00_bench_spawn.py)"""
        self.rank, self.local_rank, self.world = env_world()
        self.dist = None
        self.device = device
        if self.world > 1 or force:
            import torch.distributed as dist
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            if timeout_s is not None:
                import datetime
                kw["timeout"] = datetime.timedelta(seconds=timeout_s)
            try:
                dist.init_process_group(backend or "nccl", **kw)
            except Exception as exc:      # RCCL not usable on this box (no device, IPC refused, rendezvous failed ...)
                # A clean non-zero exit with the cause: the launcher (torchrun / the driver) sees rank failure at once.  Never a re-exec:
                # this process may have initialised the GPU already.
                import sys
                print(f"sesrq.dist: init_process_group(backend={backend or 'nccl'!r}, rank {self.rank} of {self.world}, "
                      f"MASTER_ADDR={os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}) failed: {type(exc).__name__}: {exc}\n"
                      "sesrq.dist: the data path needs no collective (frames shard, bundle replicated); only the timing fence does -- "
                      "rehearse with --dist-backend gloo, or fix the RCCL setup (HSA_ENABLE_IPC_MODE_LEGACY=0, one rank per GPU)", file=sys.stderr, flush=True)
                raise SystemExit(3)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
