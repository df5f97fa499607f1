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

  * image strips (one image faster on several GPUs): the trunk is ~90 % of a step, so rank r runs
    it only on rows [own0 - margin, own1 + margin) of the image, `margin` >= the receptive-field radius of the
    deepest tap (halo RECOMPUTE, no per-layer exchange: the trunk kernels are used unchanged on the sub-image).
    A sample belongs to the rank owning its image row; samples are ordered by owner so each rank gathers a
    contiguous block of feature rows, ONE all-reduce(sum) over zero-filled rows assembles the (N, D) matrix, the
    losses run replicated (bitwise identical on every rank), each rank scatters and back-propagates its own rows
    through its window, and ONE all-reduce(sum) of the pixel gradient (zero outside the window) precedes the
    replicated fold adjoint + RMSprop.  Exact up to fp32 rounding of the Winograd tile alignment.

Only this file touches torch.distributed; it has no GPU dependency, so the sharding logic is
covered by world_size-2 gloo tests on CPU (tests/test_parallel_gloo.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

import torch


WORLD = "world"      # StepEngine(dist_group=WORLD): shard over torch.distributed's default process group


def resolve_group(group):
    """WORLD -> None (what torch.distributed takes for the default group); a ProcessGroup stays itself."""
    return None if (isinstance(group, str) and group == WORLD) else group


def world_info(group=None):
    import torch.distributed as dist
    group = resolve_group(group)
    if group is None and not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def force_collectives() -> bool:
    """STROTSS_DIST_FORCE=1: join the process group and run every collective of a sharded step even in a world of ONE rank
    (a one-rank all-reduce is legal RCCL).  This is how the `nccl` branch below, the graph | all-reduce | graph ordering of
    a sharded step and the side-stream capture run on the one-GPU boxes this build is tested on
    (tests/test_hip_nccl_world1.py); a world of one otherwise never touches torch.distributed."""
    import os
    return os.environ.get("STROTSS_DIST_FORCE", "0") == "1"


def init_from_env(device_index=None):
    """Join the process group torchrun describes (RANK / WORLD_SIZE / MASTER_*): backend "nccl" (= RCCL over xGMI)
    unless STROTSS_DIST_BACKEND says otherwise (gloo: rehearsal of N ranks on one GPU or on the CPU).  Returns
    (rank, world); (0, 1) without touching torch.distributed when WORLD_SIZE is absent or 1."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not force_collectives():
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("STROTSS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            if device_index is None:
                device_index = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
            torch.cuda.set_device(device_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)
        ranks_share_a_gpu(None)          # the identity all-gather happens HERE, where every rank is known to be present
    return dist.get_rank(), dist.get_world_size()


_share_cache = {}


def device_identity(index=None) -> str:
    """host name + the physical identity of this process's current device (uuid, else PCI bus id): equal strings on two
    ranks = one card.  Unaffected by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES renumbering, unlike device counts."""
    import socket
    if not torch.cuda.is_available():
        return f"{socket.gethostname()}:cpu"
    p = torch.cuda.get_device_properties(torch.cuda.current_device() if index is None else index)
    ident = getattr(p, "uuid", None) or getattr(p, "pci_bus_id", None)
    if ident is None:
        ident = f"{getattr(p, 'pci_domain_id', 0)}:{getattr(p, 'pci_bus_id', '?')}:{getattr(p, 'pci_device_id', '?')}"
    return f"{socket.gethostname()}:{ident}"


def ranks_share_a_gpu(group=None) -> bool:
    """True when two ranks of the process group run on the SAME physical card (a rehearsal of N ranks on one GPU): decided
    from the devices' identities, all-gathered once per group -- not from device counts, which a launcher that masks every
    rank to one device (HIP_VISIBLE_DEVICES) makes 1 on a correct 8-ranks-on-8-GPUs node.  The engine then defaults to the
    sorted tap adjoint (the float-atomic one lost contributions next to another process in one build; the
    kernel has since been rewritten so that no long-lived lane mask decides a tap, the precaution stays).  Without a
    process group there is nothing to compare: False (independent jobs sharing a card cannot be seen from here; set
    STROTSS_DETERMINISTIC=1 for those).  Collective on first use per group: every rank builds its engine."""
    import logging
    import torch.distributed as dist
    g = resolve_group(group)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(g) <= 1:
        return False
    key = id(g)
    if key not in _share_cache:
        mine = device_identity()
        everyone = [None] * dist.get_world_size(g)
        dist.all_gather_object(everyone, mine, group=g)
        shared = len(set(everyone)) < len(everyone) and not mine.endswith(":cpu")
        _share_cache[key] = shared
        if shared and dist.get_rank(g) == 0:
            logging.getLogger("strotss").warning("ranks share a GPU (%s): the tap adjoint defaults to the sorted, atomic-free "
                                                 "form (STROTSS_DETERMINISTIC=0 overrides)", sorted(everyone))
    return _share_cache[key]


def regions_for_rank(n_regions: int, rank: int, world: int) -> List[int]:
    """Round-robin ownership: region r belongs to rank r % world."""
    return list(range(rank, n_regions, world))


def allreduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over the ranks (RCCL all-reduce over xGMI on GPU tensors; gloo on CPU)."""
    import torch.distributed as dist
    group = resolve_group(group)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collectives()):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def aggregate_throughput(units_per_rank: float, elapsed_local: float, group=None, device=None):
    """bench.py's contract: time = MAX over ranks, value = units all ranks processed / time."""
    import torch.distributed as dist
    rank, world = world_info(group)
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=resolve_group(group))
    elapsed = float(t.item())
    return world * units_per_rank / elapsed, elapsed


# ------------------------------------------------------------------ image strips
STRIP_MARGIN = 128        # halo rows per side; receptive-field radius of block5_conv3 in image pixels is < 110
STRIP_ALIGN = 16          # four 2x2 pools: windows start on multiples of 16 so every pooled grid stays aligned
HALO_MARGIN = 16          # halo-EXCHANGE strips: one row of the deepest maps (1/16 resolution); 16 >> L rows at level L


@dataclass(frozen=True)
class StripPlan:
    """Rows of an `h`-row image for rank `rank` of `world`: it owns [own0, own1) (the samples whose image row lies
    there) and runs the trunk on the window [win0, win1) = own +- margin, clipped to the image."""
    h: int
    world: int
    rank: int
    bounds: Tuple[int, ...]          # world + 1 strip boundaries, bounds[0] = 0, bounds[-1] = h
    win0: int
    win1: int
    halo: bool = False               # True: margin = HALO_MARGIN and the window's outermost rows are refreshed from the
                                     # neighbours after every layer (HaloExchange) instead of being recomputed from a 128-row margin

    @property
    def own0(self) -> int:
        return self.bounds[self.rank]

    @property
    def own1(self) -> int:
        return self.bounds[self.rank + 1]


def strip_plan(h: int, world: int, rank: int, margin: int = STRIP_MARGIN, align: int = STRIP_ALIGN,
               halo: bool = False) -> Optional[StripPlan]:
    """None when sharding does not pay: one rank, an empty strip, or windows that cover (almost) the whole image.
    halo=True: halo-exchange strips (margin = HALO_MARGIN; every strip at least 2 * HALO_MARGIN rows)."""
    if world <= 1:
        return None
    if halo:
        margin = HALO_MARGIN
    assert margin % align == 0
    bounds = [0] + [min(h, (r * h // world + align - 1) // align * align) for r in range(1, world)] + [h]
    if any(b1 <= b0 for b0, b1 in zip(bounds, bounds[1:])):
        return None
    win0, win1 = max(0, bounds[rank] - margin), min(h, bounds[rank + 1] + margin)
    widest = max(min(h, bounds[r + 1] + margin) - max(0, bounds[r] - margin) for r in range(world))
    if widest * 4 > h * 3:            # less than 25 % of the trunk saved: replicate instead
        return None
    if halo and any(b1 - b0 < 2 * HALO_MARGIN for b0, b1 in zip(bounds, bounds[1:])):
        return None                   # a strip must hold the rows it sends to both neighbours
    return StripPlan(h, world, rank, tuple(bounds), win0, win1, halo)


def halo_rows(plan: "StripPlan", n_rows: int, level: int) -> Tuple[int, int, int, int]:
    """Row indices in a window tensor of `n_rows` rows at pooling level `level`:
    (sent to the rank above, sent to the rank below, received from above, received from below)."""
    m = HALO_MARGIN >> level
    top = (plan.own0 - plan.win0) >> level
    bottom = (plan.win1 - plan.own1) >> level
    own = n_rows - top - bottom
    assert m >= 1 and own >= m and top in (0, m) and bottom in (0, m), (level, n_rows, top, bottom)
    return top + m - 1, top + own - m, 0, n_rows - 1


class HaloExchange:
    """Per-layer halo exchange of a strip-sharded trunk.  The window of rank r is its own rows +- HALO_MARGIN
    image rows; at pooling level L that is m = HALO_MARGIN >> L rows of margin.  A 3x3 convolution (or its transpose) run
    on the window with zero padding gets every row right except the OUTERMOST one on each interior side, whose true
    neighbour lies outside the window; `refresh` replaces exactly that row with the neighbour's copy (for which it is an
    own row, hence right) -- one row up and one row down per layer instead of recomputing a 128-row margin.
    Point-to-point over RCCL (xGMI neighbours); under gloo, which has no GPU send/recv, the rows are staged through the host.
    `level` = number of 2x2 pools above the tensor; tensors are (1, rows, w, c), rows = the window at that level."""

    def __init__(self, plan: StripPlan, group=None):
        import torch.distributed as dist
        assert plan.halo
        self.plan, self.group = plan, resolve_group(group)
        self.rank, self.world = plan.rank, plan.world
        self.up = self.rank - 1 if self.rank > 0 else None
        self.down = self.rank + 1 if self.rank + 1 < self.world else None
        self._stage = dist.get_backend(self.group) == "gloo"
        self.messages = 0

    def refresh(self, t: torch.Tensor, level: int) -> None:
        import torch.distributed as dist
        if self.up is None and self.down is None:
            return
        su, sd, ru, rd = halo_rows(self.plan, int(t.shape[1]), level)
        ops, recvs = [], []
        for peer, s_row, r_row in ((self.up, su, ru), (self.down, sd, rd)):
            if peer is None:
                continue
            src, dst = t[0, s_row], t[0, r_row]
            if self._stage:
                src = src.cpu()
                buf = torch.empty_like(src)
                recvs.append((dst, buf))
                dst = buf
            ops.append(dist.P2POp(dist.isend, src, peer, self.group))
            ops.append(dist.P2POp(dist.irecv, dst, peer, self.group))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for dst, buf in recvs:
            dst.copy_(buf)
        self.messages += len(ops) // 2


def sort_indices_by_strip(idx: np.ndarray, plan: StripPlan) -> Tuple[np.ndarray, List[int]]:
    """idx: (n, 2) float32 (row, col) sample coordinates.  Returns them stably ordered by owning rank and the
    world + 1 row offsets of the ranks' blocks.  The losses are invariant to the sample order as long as the content
    and prediction features use the same one."""
    owner = np.searchsorted(np.asarray(plan.bounds[1:-1]), idx[:, 0], side="right")
    order = np.argsort(owner, kind="stable")
    counts = np.bincount(owner, minlength=plan.world)
    return np.ascontiguousarray(idx[order]), [0] + np.cumsum(counts).tolist()
