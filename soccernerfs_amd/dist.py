"""Multi-GPU plumbing of the hot path: rays shard across ranks, ONE collective per step.

The reference wraps the model in DistributedDataParallel (NS/pipelines/base_pipeline.py:244-246): every rank draws its
own `train_num_rays_per_batch` rays (seed + rank, NSR/scripts/train.py:84) and gradients are mean-reduced.  Here the
whole gradient is one flat fp32 buffer, so the exchange is a single all-reduce(SUM) with the 1/world mean folded into
the optimiser (`snerf_adam_step(grad_scale=1/world)`).  Backend "nccl" = RCCL over xGMI on MI355X; "gloo" in CPU tests.
"""
import os
from typing import Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise torch.distributed from torchrun's RANK / WORLD_SIZE / MASTER_* (127.0.0.1 default). Returns (rank, world, group)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    kw = {"device_id": device} if backend == "nccl" and device is not None else {}
    dist.init_process_group(backend=backend, **kw)
    return rank, world, dist.group.WORLD


def rank_seed(base_seed: int, rank: int) -> int:
    """_set_random_seed(config.machine.seed + global_rank) (NSR/scripts/train.py:84)."""
    return base_seed + rank


def allreduce_flat_(flat_grads: torch.Tensor, group=None) -> float:
    """In-place SUM all-reduce of the flat gradient buffer; returns the grad_scale (1/world) the optimiser must apply."""
    if group is None or not dist.is_initialized():
        return 1.0
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world


def max_over_ranks(value: float, device, group=None) -> float:
    if group is None or not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
