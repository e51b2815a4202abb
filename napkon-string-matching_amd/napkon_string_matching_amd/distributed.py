"""Multi-GPU execution of one pair grid: one process per GPU (``torch.distributed``, backend
``nccl`` = RCCL over xGMI on ROCm; ``gloo`` for CPU rehearsals).

The grid shards naturally -- every pair's score depends on item i and item j only -- so the LEFT
rows are split into contiguous blocks, the right side is replicated and there is no collective on
the data path.  The only exchange is at the end: an all-gatherv of the above-threshold
``(score, i, j)`` records.  RCCL has no native gatherv: the hit buffer carries its counter in a trailing
record, so ONE all-gather of the max-padded storage moves records and counts (``all_gather_storage``: the same
function, on device-resident buffers, for ``bench.py`` and for the product's ``gen_comparable``).  Volumes are tiny next
to xGMI bandwidth (16 B per hit).
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch


def world() -> Tuple[int, int]:
    """(rank, world_size) of the default process group, (0, 1) when not initialised."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of ``n`` left rows owned by ``rank``: blocks of ceil(n / world)."""
    per = -(-n // world_size) if world_size > 0 else n
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def barrier() -> None:
    """Barrier of the default process group; nothing without one."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def broadcast_object(obj, src: int = 0):
    """``obj`` of rank ``src`` on every rank (host objects: verdicts, small texts)."""
    import torch.distributed as dist

    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def all_gather_storage(storage: torch.Tensor, out: Optional[torch.Tensor] = None, group=None, async_op: bool = False):
    """THE exchange step of a sharded grid: one all-gather of a hit buffer's ``storage`` -- ``capacity`` 16-byte
    ``(score f64, i i32, j i32)`` records followed by one record whose first 8 bytes are the hit counter
    (``grid.HitBuffer``), viewed as float64 ``[capacity + 1][2]`` -- so records and counts travel together and
    no second collective is needed (RCCL has no gatherv).  Every rank passes the same shape.

    With an RCCL group the tensors stay on the device (``out``: ``[world][capacity + 1][2]`` on the same
    device, allocated when omitted); with gloo the buffer is staged through host memory.  Returns
    ``(out, work)``; ``work`` is the handle of an ``async_op`` RCCL gather, else None.

    Callers: ``bench.py`` gathers the DEVICE-resident ``HitBuffer.storage`` at a fixed capacity, asynchronously,
    overlapped with the next grid; ``ComparableData.gen_comparable`` gathers its device-resident hits the same way
    (``all_gather_pending``: capacity agreed with one MAX all-reduce) whenever nothing has to be filtered on the host,
    and falls back to ``all_gather_hits`` (host-filtered hits: pack, copy to the device, gather, copy back) when a
    blacklist or a host-side category predicate has to look at the hits first."""
    import torch.distributed as dist

    size = dist.get_world_size(group)
    backend = dist.get_backend(group)
    if out is None:
        out = torch.empty((size,) + tuple(storage.shape), dtype=storage.dtype, device=storage.device)
    if backend == "nccl":
        work = dist.all_gather_into_tensor(out.view(-1, storage.shape[-1]), storage, group=group, async_op=async_op)
        return out, (work if async_op else None)
    host = torch.empty((size * storage.shape[0], storage.shape[1]), dtype=storage.dtype)
    dist.all_gather_into_tensor(host, storage.detach().cpu().contiguous(), group=group)
    out.copy_(host.view_as(out))
    return out, None


def agree_all(flag: bool) -> bool:
    """True iff ``flag`` is true on EVERY rank (one MIN all-reduce): decisions that change which collectives follow must
    be taken by all ranks together."""
    import torch.distributed as dist

    rank, size = world()
    if size == 1:
        return bool(flag)
    on_device = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def all_gather_pending(pending, n_left: int, nlev_left: np.ndarray, nlev_right: np.ndarray, world_size: int):
    """The exchange of a sharded ``gen_comparable`` whose hits are still in their DEVICE buffers (``grid.PendingHits``;
    a rank whose shard had nothing to score passes None): one MAX all-reduce agrees on the capacity, every rank's
    ``capacity + 1`` records go through ONE all-gather (``all_gather_storage``: device to device under RCCL), and only the
    gathered result is copied to the host.  The records carry row ids relative to each rank's own sub-grid (its left
    shard's items with at least one level x the right items with at least one level); every rank knows every shard's
    bounds, so the mapping back to frame positions is done here.  Returns (score, i, j) in the canonical order."""
    import torch.distributed as dist

    on_device = dist.get_backend() == "nccl"
    dev = pending.buf.records.device if pending is not None else (
        torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    red = dev if on_device else torch.device("cpu")
    cap = torch.tensor([pending.n if pending is not None else 0], dtype=torch.int64, device=red)
    dist.all_reduce(cap, op=dist.ReduceOp.MAX)
    capacity = max(1, int(cap.item()))
    if pending is not None:
        storage = pending.storage(capacity)
    else:
        storage = torch.zeros((capacity + 1, 2), dtype=torch.float64, device=dev)
    gathered, _ = all_gather_storage(storage)
    g = gathered.detach().cpu().numpy()
    counts = g.view(np.int64).reshape(g.shape[0], capacity + 1, 2)[:, capacity, 0]
    ij = g.view(np.int32).reshape(g.shape[0], capacity + 1, 4)
    keep_r = np.flatnonzero(np.asarray(nlev_right) > 0)
    has_level = np.asarray(nlev_left) > 0
    parts_s, parts_i, parts_j = [], [], []
    for r in range(world_size):
        lo, hi = shard_bounds(n_left, r, world_size)
        keep_l = lo + np.flatnonzero(has_level[lo:hi])
        n = int(min(max(counts[r], 0), capacity))
        parts_s.append(g[r, :n, 0])
        parts_i.append(keep_l[ij[r, :n, 2]] if n else np.zeros(0, np.int64))
        parts_j.append(keep_r[ij[r, :n, 3]] if n else np.zeros(0, np.int64))
    s, gi, gj = np.concatenate(parts_s), np.concatenate(parts_i).astype(np.int64), np.concatenate(parts_j).astype(np.int64)
    order = np.lexsort((gj, gi, -s))
    return s[order], gi[order], gj[order]


def pack_hits(score: np.ndarray, i: np.ndarray, j: np.ndarray, capacity: int) -> torch.Tensor:
    """Host arrays -> the wire format of ``all_gather_storage`` (``capacity`` >= len(score))."""
    n = len(score)
    rec = np.zeros((capacity + 1, 2), dtype=np.float64)
    rec[:n, 0] = score
    ij = rec.view(np.int32).reshape(capacity + 1, 4)
    ij[:n, 2], ij[:n, 3] = i, j
    rec.view(np.int64).reshape(capacity + 1, 2)[capacity, 0] = n
    return torch.from_numpy(rec)


def unpack_storage(gathered: torch.Tensor) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """``[world][capacity + 1][2]`` float64 -> the valid records of every rank, concatenated in rank order."""
    g = gathered.detach().cpu().numpy()
    cap = g.shape[1] - 1
    counts = g.view(np.int64).reshape(g.shape[0], cap + 1, 2)[:, cap, 0]
    ij = g.view(np.int32).reshape(g.shape[0], cap + 1, 4)
    parts_s, parts_i, parts_j = [], [], []
    for r in range(g.shape[0]):
        n = int(min(max(counts[r], 0), cap))
        parts_s.append(g[r, :n, 0])
        parts_i.append(ij[r, :n, 2])
        parts_j.append(ij[r, :n, 3])
    return np.concatenate(parts_s), np.concatenate(parts_i).astype(np.int64), np.concatenate(parts_j).astype(np.int64)


def all_gather_hits(
    score: np.ndarray, i: np.ndarray, j: np.ndarray, device: Optional[torch.device] = None
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Every rank contributes its local hits and receives all of them, in the canonical order
    (score descending, i, j ascending).  ``i`` must already be GLOBAL left indices (< 2^31).

    The ranks first agree on the padded capacity (one 8-byte MAX all-reduce), then the records and their
    counts move in the single all-gather of ``all_gather_storage``."""
    import torch.distributed as dist

    rank, size = world()
    if size == 1:
        order = np.lexsort((j, i, -score))
        return score[order], i[order], j[order]
    on_device = dist.get_backend() == "nccl"
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")
    cap = torch.tensor([len(score)], dtype=torch.int64, device=device)
    dist.all_reduce(cap, op=dist.ReduceOp.MAX)
    capacity = max(1, int(cap.item()))
    gathered, _ = all_gather_storage(pack_hits(score, i, j, capacity).to(device))
    s, gi, gj = unpack_storage(gathered)
    order = np.lexsort((gj, gi, -s))
    return s[order], gi[order], gj[order]
