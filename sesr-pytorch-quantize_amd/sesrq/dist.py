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


class Group:
    """Thin wrapper over torch.distributed used by bench.py; a no-op for world_size 1."""

    def __init__(self, backend: Optional[str] = None, device=None):
        self.rank, self.local_rank, self.world = env_world()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend or "nccl", **kw)
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
