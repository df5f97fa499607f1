"""Multi-GPU host logic (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm).

The reference is single-device (nn/utils.py:73-85 picks ONE GPU; no tf.distribute).  Two modes:

  * replicas (throughput, BASELINE config 5): independent content/style pairs, one per GPU, no
    collective on the data path -- `bench.py --gpus N`;
  * region sharding (BASELINE config 4): with masks the loss is a mean over R regions sharing one
    VGG pass (run_strotss.py:104-125).  Regions are dealt round-robin to the ranks; every rank runs
    the replicated fold + trunk forward, its regions' losses and the trunk backward of THEIR
    gradient, then ONE all-reduce(sum) of the pixel gradient (3*H*W floats, 12 MiB at 1024^2) before
    the fold adjoint, so all ranks apply the identical RMSprop update.  The six pyramid-variable
    gradients are not reduced separately: the fold adjoint is linear.

Only this file touches torch.distributed; it has no GPU dependency, so the sharding logic is
covered by world_size-2 gloo tests on CPU (tests/test_parallel_gloo.py).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch


def world_info(group=None):
    import torch.distributed as dist
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def regions_for_rank(n_regions: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: region r belongs to rank r % world."""
    return list(range(rank, n_regions, world))


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks (RCCL all-reduce over xGMI on GPU tensors; gloo on CPU)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def sharded_pixel_gradient(region_grad: Callable[[int], torch.Tensor], n_regions: int, like: torch.Tensor,
                           group=None) -> torch.Tensor:
    """Sum of `region_grad(r)` (already scaled by 1/R) over all regions, computing only the regions
    this rank owns and all-reducing the pixel gradient once."""
    rank, world = world_info(group)
    g = torch.zeros_like(like)
    for r in regions_for_rank(n_regions, rank, world):
        g += region_grad(r)
    return allreduce_sum_(g, group)


def aggregate_throughput(units_per_rank: float, elapsed_local: float, group=None, device=None):
    """bench.py's contract: time = MAX over ranks, value = units all ranks processed / time."""
    import torch.distributed as dist
    rank, world = world_info(group)
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    elapsed = float(t.item())
    return world * units_per_rank / elapsed, elapsed
