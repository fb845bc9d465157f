"""Multi-GPU plumbing of the hot path: rays shard across ranks, ONE collective per step.

The reference wraps the model in DistributedDataParallel (NS/pipelines/base_pipeline.py:244-246): every rank draws its
own `train_num_rays_per_batch` rays (seed + rank, NSR/scripts/train.py:84) and gradients are mean-reduced.  Here the
whole gradient is one flat fp32 buffer, so the exchange is a single all-reduce(SUM) with the 1/world mean folded into
the optimiser (`snerf_adam_step(grad_scale=1/world)`).  Backend "nccl" = RCCL over xGMI on MI355X; "gloo" in CPU tests.

Default for world > 1 (trainer.KPlanesTrainer.shard_optimizer): the field planes -- 98 % of the bytes -- take the two halves of
that all-reduce separately with the optimiser in between: reduce-scatter of the gradient, Adam + regularisers on this rank's
1/world shard, all-gather of the new parameters.  Same bytes on the wire, but the dense optimiser sweep (the dominant kernel at
one GPU) shrinks by 1/world, the reduce-scatter runs under the proposal-network backward and the all-gather under the next
step's proposal forward.  The small segments (proposal planes, MLPs) keep the plain all-reduce.
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
        all_reduce_sum_(flat_grads, group)
    return 1.0 / world


class _Done:
    """Stand-in for a completed collective (blocking backends)."""

    def wait(self):
        return True


def _is_nccl(group) -> bool:
    return dist.get_backend(group) == "nccl"


def all_reduce_sum_(t: torch.Tensor, group, async_op: bool = False):
    """SUM all-reduce in place.  RCCL: on the device, optionally asynchronous (the returned work's wait() makes the CURRENT stream
    wait, the host never blocks).  gloo (tests): staged through host memory when the tensor lives on a GPU."""
    if _is_nccl(group):
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return w if async_op else _Done()
    if t.is_cuda:
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return _Done()


def all_reduce_max_(t: torch.Tensor, group):
    """MAX all-reduce in place (tiny tensors: the optimiser's non-finite flags)."""
    if _is_nccl(group):
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    elif t.is_cuda:
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)


def reduce_scatter_sum(out_shard: torch.Tensor, full: torch.Tensor, group, async_op: bool = False):
    """out_shard (numel n) = this rank's slice [rank*n, (rank+1)*n) of the SUM over ranks of `full` (numel world*n)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert full.numel() == world * out_shard.numel(), "reduce_scatter_sum: full must hold world equal shards"
    if _is_nccl(group):
        w = dist.reduce_scatter_tensor(out_shard, full, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        return w if async_op else _Done()
    host = full.detach().to("cpu", copy=True)  # gloo has no reduce-scatter: all-reduce a host copy, keep this rank's slice
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    n = out_shard.numel()
    out_shard.copy_(host[rank * n:(rank + 1) * n])
    return _Done()


def all_gather_shards(full: torch.Tensor, shard: torch.Tensor, group, async_op: bool = False):
    """full (numel world*n) = concatenation over ranks of `shard` (numel n)."""
    world = dist.get_world_size(group)
    assert full.numel() == world * shard.numel(), "all_gather_shards: full must hold world equal shards"
    if _is_nccl(group):
        w = dist.all_gather_into_tensor(full, shard, group=group, async_op=async_op)
        return w if async_op else _Done()
    parts = [torch.empty(shard.numel(), dtype=shard.dtype) for _ in range(world)]
    dist.all_gather(parts, shard.detach().cpu().contiguous(), group=group)
    full.copy_(torch.cat(parts))
    return _Done()


def max_over_ranks(value: float, device, group=None) -> float:
    if group is None or not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if _is_nccl(group) else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class CAbiComm:
    """The per-step gradient all-reduce through libsnerf's own RCCL communicator (snerf_comm_* / snerf_allreduce_grads, include/snerf.h)
    instead of torch.distributed: rank 0 makes the unique id, torch.distributed (any backend) carries its 128 bytes to the other ranks,
    every rank joins.  `all_reduce_sum_(flat)` enqueues ONE ncclAllReduce(SUM) on the current stream and returns the optimiser's grad_scale."""

    def __init__(self, group, device):
        import ctypes as C

        from . import _lib

        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self._lib, self._C = _lib, C
        ident = (C.c_ubyte * 128)()
        if self.rank == 0:
            _lib.check(_lib.lib().snerf_comm_unique_id(ident), "comm_unique_id")
        t = torch.tensor(list(ident), dtype=torch.uint8, device=device if _is_nccl(group) else "cpu")
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0, group=group)
        ident = (C.c_ubyte * 128)(*t.cpu().tolist())
        self._comm = C.c_void_p()
        _lib.check(_lib.lib().snerf_comm_create(self.world, self.rank, ident, C.byref(self._comm)), "comm_create")

    def all_reduce_sum_(self, flat: torch.Tensor) -> float:
        C = self._C
        self._lib.check(self._lib.lib().snerf_allreduce_grads(self._comm, C.c_void_p(flat.data_ptr()), C.c_int64(flat.numel()),
                                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "allreduce_grads")
        return 1.0 / self.world

    def close(self):
        if self._comm:
            self._lib.check(self._lib.lib().snerf_comm_destroy(self._comm), "comm_destroy")
            self._comm = None
