"""Multi-GPU plan of the hot path: segment sharding + ONE all-gather of visual tokens (RCCL over xGMI).

The reference's inference is single-GPU (inference.py:17); this is new functionality described by
SURVEY.md §8(e): segments are independent in both encoders (models/llava_next_video.py:503-505,
:530-532), so rank r encodes a contiguous block of segments with replicated vision weights and the
per-segment token blocks `[image | temporal | newline]` (:563) are exchanged once per clip.  The
payload is small (<= 3.5 MB / rank) and latency-bound, so it is a single un-chunked collective; with
xGMI's point-to-point links every rank pushes its block to all peers at once.

One process per GPU, `torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_units: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous balanced blocks: the first (n_units % world) ranks get one extra unit."""
    base, rem = divmod(n_units, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def my_shard(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    return shard_bounds(n_units, world)[rank]


def allgather_visual(local: torch.Tensor, n_units: int, rows_per_unit: int, group: Optional[dist.ProcessGroup] = None,
                     force_collective: bool = False, gather=None, gatherv=None) -> torch.Tensor:
    """local: [n_local * rows_per_unit, D] for this rank's block -> full [n_units * rows_per_unit, D] on every rank.

    Blocks are padded to the largest block so that a single fixed-size all_gather_into_tensor is used
    (12 segments over 8 ranks is uneven: 2,2,2,2,1,1,1,1).
    gather: None = torch.distributed.all_gather_into_tensor; or a callable send [rows, D] -> recv [world * rows, D] in rank order --
    Engine.allgather_visual, i.e. libgvl's own RCCL communicator behind the C ABI (gvl_comm_init / gvl_allgather_visual): the exchange a
    non-Python host of the library performs.
    gatherv: Engine.allgatherv_visual (gvl_allgatherv_visual): the uneven blocks go STRAIGHT into the segment-ordered result -- no padded send buffer,
    no re-assembly (at <= 3.5 MB per rank the pad / copy / cat kernels cost more than the wire); takes precedence over `gather`."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1 and not force_collective:       # force_collective: run the (degenerate) collective anyway -- RCCL smoke test on one GPU
        return local
    bounds = shard_bounds(n_units, world)
    if gatherv is not None:
        lo, hi = bounds[rank]
        assert local.shape[0] == (hi - lo) * rows_per_unit, (local.shape, lo, hi)
        return gatherv(local, [(h - l) * rows_per_unit for l, h in bounds])
    max_units = max(hi - lo for lo, hi in bounds)
    D = local.shape[1]
    lo, hi = bounds[rank]
    assert local.shape[0] == (hi - lo) * rows_per_unit, (local.shape, lo, hi)
    send = local.new_zeros((max_units * rows_per_unit, D))
    send[: local.shape[0]] = local
    if gather is not None:
        recv = gather(send)
        assert recv.shape == (world * max_units * rows_per_unit, D), recv.shape
    else:
        recv = local.new_empty((world * max_units * rows_per_unit, D))
        dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, max_units * rows_per_unit, D)
    parts = [recv[r, : (b[1] - b[0]) * rows_per_unit] for r, b in enumerate(bounds)]
    return torch.cat(parts, dim=0)


# ---- N clips in flight (bench.py --gpus N): every clip's segment batch is sharded over ALL ranks, with the block
# assignment rotated per clip so that the n_units % world remainder balances (each rank encodes exactly n_units units).
def rotated_encode_plan(n_units: int, rank: int, world: int) -> List[Tuple[int, int, int]]:
    """Units this rank encodes, in encode order: [(clip, lo, hi)].  Block b of clip c goes to rank (b + c) % world."""
    bounds = shard_bounds(n_units, world)
    plan = []
    for c in range(world):
        lo, hi = bounds[(rank - c) % world]
        if hi > lo:
            plan.append((c, lo, hi))
    return plan


def init_gvl_comm(engine, group: Optional[dist.ProcessGroup] = None) -> None:
    """libgvl's own communicator over the ranks of `group`: ncclGetUniqueId on the group's first rank (gvl_comm_unique_id), its 128 bytes
    travel through torch.distributed's object broadcast (any host-side channel would do), ncclCommInitRank on every rank (gvl_comm_init).
    Afterwards Engine.allgather_visual is the exchange of the visual tokens."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    uid = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=src, group=group)
    engine.comm_init(uid[0], rank, world)


def rotated_gather_index(n_units: int, rank: int, world: int) -> List[Tuple[int, int, int]]:
    """Where clip == rank's units sit after the all-gather: [(src_rank, unit_offset_in_src, n)] in unit order 0..n_units."""
    bounds = shard_bounds(n_units, world)
    out = []
    for b, (lo, hi) in enumerate(bounds):
        if hi == lo:
            continue
        src = (b + rank) % world
        off = 0
        for c, l2, h2 in rotated_encode_plan(n_units, src, world):
            if c == rank:
                break
            off += h2 - l2
        out.append((src, off, hi - lo))
    return out
