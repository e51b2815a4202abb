"""Multi-GPU execution of one pair grid: one process per GPU (``torch.distributed``, backend
``nccl`` = RCCL over xGMI on ROCm; ``gloo`` for CPU rehearsals).

The grid shards naturally -- every pair's score depends on item i and item j only -- so the LEFT
rows are split into contiguous blocks, the right side is replicated and there is no collective on
the data path.  The only exchange is at the end: an all-gatherv of the above-threshold
``(score, i, j)`` records (RCCL has no native gatherv: all-gather of the counts, then all-gather of
max-padded buffers).  Volumes are tiny next to xGMI bandwidth (16 B per hit).
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


def all_gather_hits(
    score: np.ndarray, i: np.ndarray, j: np.ndarray, device: Optional[torch.device] = None
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Every rank contributes its local hits and receives all of them, in the canonical order
    (score descending, i, j ascending).  ``i`` must already be GLOBAL left indices."""
    import torch.distributed as dist

    rank, size = world()
    if size == 1:
        order = np.lexsort((j, i, -score))
        return score[order], i[order], j[order]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = len(score)
    counts = torch.zeros(size, dtype=torch.int64, device=device)
    mine = torch.tensor([n], dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(counts, mine)
    counts_h = counts.cpu().numpy()
    cap = max(1, int(counts_h.max()))
    rec = np.zeros((cap, 3), dtype=np.float64)  # score, i, j (indices are exact in a double)
    rec[:n, 0], rec[:n, 1], rec[:n, 2] = score, i, j
    local = torch.from_numpy(rec).to(device)
    gathered = torch.empty((size * cap, 3), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(gathered, local)
    g = gathered.cpu().numpy().reshape(size, cap, 3)
    parts = [g[r, : int(counts_h[r])] for r in range(size)]
    allrec = np.concatenate(parts, axis=0) if parts else np.zeros((0, 3))
    s, gi, gj = allrec[:, 0], allrec[:, 1].astype(np.int64), allrec[:, 2].astype(np.int64)
    order = np.lexsort((gj, gi, -s))
    return s[order], gi[order], gj[order]
