"""Data-parallel plumbing for the hot path: one process per GPU, RCCL (backend "nccl" on ROCm) over xGMI.

The reference's multi-GPU mode is DDP over rays (runner.py:203-207, :575: per-rank batch = batch_size // world_size,
SURVEY.md F4: expert parallelism is disabled as shipped).  Rays are independent and routing is rank-local, so ranks
never exchange activations; the only collective is one all-reduce of the flat fp32 gradient buffer per step, after which
every rank applies the same Adam update with grad_scale = 1 / world_size.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise torch.distributed from torchrun's environment (RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_rays(n_rays: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous per-rank slice [begin, end) of a global ray batch (runner.py:575: batch_size // world_size each;
    the remainder, if any, is dropped like DataLoader(drop_last) would)."""
    per = n_rays // world
    return rank * per, (rank + 1) * per


def make_grad_allreduce(group=None) -> Callable[[torch.Tensor], float]:
    """Returns f(flat_grad) -> grad_scale: sums the flat gradient buffer over ranks in place (one bucket) and returns
    the 1/world factor that swn_adam_step applies."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1

    def f(flat: torch.Tensor) -> float:
        if world > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world
    return f
