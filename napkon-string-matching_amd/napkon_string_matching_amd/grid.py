"""Launch wrappers: one N x M pair grid -> the list of above-threshold hits.

This is the device half of the reference's per-pair loop
(``gen_comparable``: types/comparable_data.py:223-232,243 and ``compare``: :123-126).  The
per-item error surfaces of the reference (``ZeroDivisionError`` for empty-vs-empty Jaccard,
score_functions.py:13; ``IndexError`` for a zero-level item, comparable_data.py:262) are raised
here, before the launch, from per-item properties; the kernels never see such a pair as a hit.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib
from .tables import LevelItems, SetTable, StrTable

HIT_BYTES = 16
DEFAULT_CAPACITY = 1 << 20


@dataclass
class Hits:
    """Above-threshold pairs in canonical order (score descending, then i, then j)."""

    score: np.ndarray  # float64
    i: np.ndarray  # int32, caller's left item index
    j: np.ndarray  # int32, caller's right item index

    def __len__(self) -> int:
        return int(self.score.shape[0])

    def as_tuples(self):
        return list(zip(self.score.tolist(), self.i.tolist(), self.j.tolist()))


def _require_gpu(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise _lib.NsmLibraryError(
            "the match loop only runs on an MI355X (HIP device); there is no CPU fallback"
        )
    return dev


class HitBuffer:
    """Caller-owned hit storage: ``capacity`` 16-byte records followed by the device counter, in ONE
    allocation -- so that the multi-GPU exchange is a single all-gather of ``storage``."""

    def __init__(self, capacity: int, device) -> None:
        self.capacity = int(capacity)
        rows = max(1, self.capacity)
        self.storage = torch.zeros((rows + 1, 2), dtype=torch.float64, device=device)
        self.records = self.storage[:rows]
        self.count = self.storage[rows:].view(torch.int64).view(-1)[:1]  # the trailing record's first 8 bytes
        self.scratch: Optional[torch.Tensor] = None

    def reset(self) -> None:
        self.count.zero_()

    def views(self, n: int):
        score = self.records[:n, 0]
        ij = self.records.view(torch.int32).view(-1, 4)[:n, 2:4]
        return score, ij[:, 0], ij[:, 1]


SMALL_SORT_MAX = 8192  # records nsm_sort_hits orders in one workgroup's LDS (no scratch buffer needed)


def sort_hits_device(buf: HitBuffer, n: int, id_limit: int = 0) -> Hits:
    """Canonical order on the device, then one D2H copy of the n records.  The host has read the counter (``n``), so
    the sort's geometry follows the hits, not the buffer: one launch up to 8192 records at any capacity."""
    if n == 0:
        return Hits(np.zeros(0, np.float64), np.zeros(0, np.int32), np.zeros(0, np.int32))
    lib = _lib.load()
    scratch_ptr = 0
    if n > SMALL_SORT_MAX:
        if buf.scratch is None or buf.scratch.shape[0] < n:
            buf.scratch = torch.empty((n, 2), dtype=torch.float64, device=buf.records.device)
        scratch_ptr = buf.scratch.data_ptr()
    stream = torch.cuda.current_stream(buf.records.device).cuda_stream
    _lib.check(
        lib.nsm_sort_hits(buf.records.data_ptr(), scratch_ptr, buf.capacity, buf.count.data_ptr(), n,
                          max(0, min(int(id_limit), 0x7FFFFFFF)), stream),
        "nsm_sort_hits",
    )
    host = buf.records[:n].cpu().numpy()
    ij = host.view(np.int32).reshape(n, 4)
    return Hits(host[:, 0].copy(), ij[:, 2].copy(), ij[:, 3].copy())


class PendingHits:
    """The hits of a finished grid still in their device buffer (``run_grid(..., defer=True)``): ``finish()`` orders them
    and copies them to the host as always; ``storage(capacity)`` is the wire format of the multi-GPU exchange
    (``distributed.all_gather_storage``) at a capacity the ranks agreed on -- ``capacity`` records and the counter record,
    built on the device, so a sharded ``compare()`` moves its hits GPU -> RCCL -> GPU without a detour through the host."""

    def __init__(self, buf: HitBuffer, n: int, id_limit: int) -> None:
        self.buf, self.n, self.id_limit = buf, int(n), int(id_limit)

    def finish(self) -> Hits:
        return sort_hits_device(self.buf, self.n, self.id_limit)

    def storage(self, capacity: int) -> torch.Tensor:
        out = torch.zeros((int(capacity) + 1, 2), dtype=torch.float64, device=self.buf.records.device)
        out[: self.n] = self.buf.records[: self.n]
        out[int(capacity):].view(torch.int64).view(-1)[0] = self.n
        return out


def run_grid(launch: Callable[[HitBuffer, int], int], device, capacity: Optional[int], what: str,
             id_limit: int = 0, defer: bool = False):
    """Run ``launch`` with a hit buffer, growing it once if the counter overflowed.  ``id_limit``: an upper bound of
    the row ids the grid reports (the larger side's item count), 0 = unknown.  ``defer``: return the hits in their
    device buffer (``PendingHits``) instead of ordering and copying them."""
    dev = _require_gpu(device)
    buf = HitBuffer(capacity or DEFAULT_CAPACITY, dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for _attempt in range(2):
        buf.reset()
        _lib.check(launch(buf, stream), what)
        n = int(buf.count.item())  # synchronises the stream
        if n <= buf.capacity:
            return PendingHits(buf, n, id_limit) if defer else sort_hits_device(buf, n, id_limit)
        buf = HitBuffer(n, dev)
    raise _lib.NsmLibraryError(f"{what}: hit count changed between two identical launches")


# ------------------------------------------------------------------------------- RAW grids
def jaccard_raw_grid(
    left: SetTable, right: SetTable, threshold: float, prune: bool = True, capacity: Optional[int] = None,
    index: Optional[bool] = None,
) -> Hits:
    """``intersection_vs_union`` on one set per item, all N x M pairs, hits ``>= threshold``.
    ``index``: None = the library decides (candidates from the right table's global inverted index where its posting
    statistics say they are few, from a per-tile index at low thresholds, else the signature kernel over all pairs);
    True = force an index (the global one when the right table carries it), "tile" = force the per-tile index,
    False = never an index."""
    if left.side != "left" or right.side != "right":
        raise ValueError("tables must be encoded with side='left' and side='right' (distinct padding)")
    if left.has_empty and right.has_empty:
        # score_functions.py:13 -- len(set() | set()) == 0
        raise ZeroDivisionError("division by zero")
    lib = _lib.load()
    ls, rs = left.struct(), right.struct()
    flags = (_lib.FLAG_PRUNE if prune else 0) | (0 if index is None else (_lib.FLAG_INDEX if index else _lib.FLAG_NO_INDEX))
    if index == "tile":
        flags |= _lib.FLAG_TILE_INDEX

    def launch(buf: HitBuffer, stream: int) -> int:
        return lib.nsm_jaccard_raw_grid(
            ls, rs, float(threshold), flags, buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(), stream
        )

    # (0 = unknown on either side: the sort keeps all 32 bits of the ids)
    id_limit = max(left.id_limit, right.id_limit) if left.id_limit and right.id_limit else 0
    return run_grid(launch, left.ids.device, capacity, "nsm_jaccard_raw_grid", id_limit=id_limit)


def indel_raw_grid(
    left: StrTable, right: StrTable, threshold: float, prune: bool = True, capacity: Optional[int] = None,
    two_stage: bool = True,
) -> Hits:
    """``fuzzy_match`` (QRatio/100 = Indel ratio after default_process) on one string per item.
    ``two_stage=False``: the 32-bucket histogram test for every pair even when both tables carry the 16-bucket
    column (A/B runs, tests; same hits)."""
    lib = _lib.load()
    ls, rs = left.struct(), right.struct()
    flags = (_lib.FLAG_PRUNE if prune else 0) | (0 if two_stage else _lib.FLAG_ONE_STAGE)

    def launch(buf: HitBuffer, stream: int) -> int:
        return lib.nsm_indel_raw_grid(
            ls, rs, float(threshold), flags, buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(), stream
        )

    return run_grid(launch, left.codes.device, capacity, "nsm_indel_raw_grid")


# ------------------------------------------------------------------------------- levels grids
def jaccard_levels_grid(
    left: SetTable, right: SetTable, threshold: float, category_mode: int = _lib.CAT_NONE, prune: bool = True,
    capacity: Optional[int] = None, index: Optional[bool] = None, defer: bool = False,
) -> Hits:
    """``compare_terms`` with ``intersection_vs_union`` over suffix-nested levels.
    ``index``: None = the library decides (candidates from the right table's global inverted index where its posting
    statistics say they are few, from a per-tile index at low thresholds, else the filter kernel over all pairs);
    True = force an index (the global one when the right table carries it), "tile" = force the per-tile index,
    False = never an index."""
    if left.nlev is None or right.nlev is None:
        raise ValueError("levels grid needs tables built with SetTable.from_levels")
    lib = _lib.load()
    ls, rs = left.struct(), right.struct()
    flags = (_lib.FLAG_PRUNE if prune else 0) | (0 if index is None else (_lib.FLAG_INDEX if index else _lib.FLAG_NO_INDEX))
    if index == "tile":
        flags |= _lib.FLAG_TILE_INDEX
    if (left.seg is None) != (right.seg is None) or left.category_mode != right.category_mode:
        raise ValueError("both sides must be encoded alike: same category_mode and partition (tables.partition_allowed)")
    if left.category_mode is not None:  # the encoder may have rewritten the predicate (partition)
        category_mode = left.category_mode

    def launch(buf: HitBuffer, stream: int) -> int:
        return lib.nsm_jaccard_levels_grid(
            ls, rs, float(threshold), int(category_mode), flags, buf.records.data_ptr(), buf.capacity,
            buf.count.data_ptr(), stream,
        )

    return run_grid(launch, left.ids.device, capacity, "nsm_jaccard_levels_grid", defer=defer)


PROBE_MIN_PAIRS = 1 << 27      # grids of at least this many pairs are probed before they are routed (11 600 x 11 600)
PROBE_LEFT_ROWS = 8192          # left rows of the probe's sample
SPLIT_MAX_SURVIVAL = 0.10       # of the pairs a grid visits: up to here the split path, beyond it the shared-tile kernel


def _visited_pairs(left: LevelItems, right: LevelItems) -> float:
    """Pairs the one-word kernels score step 1 for: all of them, or -- partitioned tables -- the same-category ones."""
    if left.seg_start is None or right.seg_start is None:
        return float(left.n) * float(right.n)
    a = left.seg_start.cpu().numpy().astype(np.float64)
    b = right.seg_start.cpu().numpy().astype(np.float64)
    return float(((a[1:] - a[:-1]) * (b[1:] - b[:-1])).sum())


def probe_survival(left: LevelItems, left_strings: StrTable, right: LevelItems, right_strings: StrTable, threshold: float,
                   category_mode: int):
    """MEASURE, on a sample of the left rows, how many pairs outlive step 1 of ``compare_terms`` x ``fuzzy_match`` on this
    grid (one-word level strings): every k-th row of the left items table (the rows stay grouped by category and ordered by
    depth) against the whole right side through the scan kernel alone (``NSM_FLAG_PROBE``), the survivors read from the
    workspace's queue counters.  Returns (expected survivors of the full grid, pairs the full grid visits), or None when
    the grid is not one the split path could take.  Which path is fastest depends on that rate and nothing else that a
    threshold could tell: word-like text at 0.55 lets 2.8 % of the same-category pairs through, digit strings at 0.65
    about 60 %."""
    lib = _lib.load()
    dev = left.first.device
    k = max(1, left.n // PROBE_LEFT_ROWS)
    idx = torch.arange(0, left.n, k, device=dev)
    seg = seg_start = None
    if left.seg is not None:
        seg = left.seg[idx].contiguous()
        seg_start = torch.zeros(65, dtype=torch.int32, device=dev)
        seg_start[1:] = torch.cumsum(torch.bincount(seg, minlength=64)[:64], 0).to(torch.int32)
    sample = LevelItems(first=left.first[idx].contiguous(), nlev=left.nlev[idx].contiguous(), orig=left.orig[idx].contiguous(),
                        cat=None if left.cat is None else left.cat[idx].contiguous(), n=int(idx.numel()), seg=seg,
                        seg_start=seg_start, category_mode=left.category_mode)
    flags = _lib.FLAG_PRUNE | _lib.FLAG_SPLIT | _lib.FLAG_PROBE
    si, ls, ri, rs = sample.struct(), left_strings.struct(), right.struct(), right_strings.struct()
    if int(lib.nsm_indel_levels_workspace_bytes(si, ls, ri, rs, float(threshold), flags, 0.0)) == 0:
        return None
    ws = torch.zeros(64 + 2 * (1 << 16), dtype=torch.int64, device=dev)  # control words + two small queue halves
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.nsm_indel_levels_grid(si, ls, ri, rs, float(threshold), int(category_mode), flags, 0, 0, count.data_ptr(),
                                         ws.data_ptr(), ws.numel() * 8, 1.0, stream), "nsm_indel_levels_grid (probe)")
    # (expected survivors 1: ONE round over the sample whatever the queue holds -- only the counters matter)
    survivors = float(ws[2:64].sum().item())  # (synchronises; the counters keep counting past the queue's capacity)
    visited_sample = _visited_pairs(sample, right)
    visited = _visited_pairs(left, right)
    if visited_sample <= 0:
        return 0.0, visited
    return survivors / visited_sample * visited, visited


def route_one_word(left: LevelItems, left_strings: StrTable, right: LevelItems, right_strings: StrTable, threshold: float,
                   category_mode: int):
    """(extra flags, expected survivors, what was measured) for a large grid of one-word level strings: probe, then the split
    path when few pairs outlive step 1, the shared-tile kernel when many do.  (0, 0.0, None): leave it to the library."""
    got = probe_survival(left, left_strings, right, right_strings, threshold, category_mode)
    if got is None:
        return 0, 0.0, None
    expected, visited = got
    rate = expected / visited if visited > 0 else 0.0
    if rate <= SPLIT_MAX_SURVIVAL:
        # (sampling error: a quarter more queue than the estimate)
        return _lib.FLAG_SPLIT, max(1.0, 1.25 * expected), {"survival_rate": rate, "path": "split", "pairs_visited": visited}
    return _lib.FLAG_TILE, 0.0, {"survival_rate": rate, "path": "tile", "pairs_visited": visited}


def indel_levels_grid(
    left: LevelItems, left_strings: StrTable, right: LevelItems, right_strings: StrTable, threshold: float,
    category_mode: int = _lib.CAT_NONE, prune: bool = True, capacity: Optional[int] = None, wave_wide: bool = False,
    park: bool = False, workspace: Optional[int] = None, return_overflow: Optional[list] = None, defer: bool = False,
    probe: Optional[bool] = None, route: Optional[list] = None,
) -> Hits:
    """``compare_terms`` with ``fuzzy_match`` over per-level strings.  ``wave_wide`` selects the kernel
    without block-cooperative parking, ``park`` the round-2 kernel for multi-word strings (same hits; A/B runs
    and tests).  ``workspace``: bytes of split-path scratch to hand to the library (None = what it asks for, 0 = none:
    the single-kernel path); ``return_overflow``: a list that receives the workspace's overflow word (tests).
    ``probe``: measure the survival rate of step 1 on a sample first and route by it (None = for large grids of one-word
    strings; ``route`` receives what was decided)."""
    lib = _lib.load()
    li, ls, ri, rs = left.struct(), left_strings.struct(), right.struct(), right_strings.struct()
    flags = (_lib.FLAG_PRUNE if prune else 0) | (_lib.FLAG_WAVE_WIDE if wave_wide else 0) | (_lib.FLAG_PARK if park else 0)
    if (left.seg is None) != (right.seg is None) or left.category_mode != right.category_mode:
        raise ValueError("both sides must be encoded alike: same category_mode and partition (tables.partition_allowed)")
    if left.category_mode is not None:  # the encoder may have rewritten the predicate (partition)
        category_mode = left.category_mode

    dev = left.first.device
    expected = 0.0
    if probe is None:
        probe = (workspace is None and prune and not wave_wide and not park and left_strings.stride == 64 and threshold > 0 and
                 float(left.n) * float(right.n) >= PROBE_MIN_PAIRS)
    if probe:
        extra, expected, measured = route_one_word(left, left_strings, right, right_strings, threshold, category_mode)
        flags |= extra
        if route is not None and measured is not None:
            route.append(dict(measured, expected_survivors=expected))
    # the split path's survivor queue (scan kernel -> queue -> finish kernel; one-word strings at thresholds >= 0.7, or
    # wherever the probe found few survivors) is CALLER-owned scratch: a torch tensor, so torch's allocator owns it and it
    # goes back to the cache with this call
    want = int(lib.nsm_indel_levels_workspace_bytes(li, ls, ri, rs, float(threshold), flags, float(expected))) if workspace is None \
        else int(workspace)
    ws = split_workspace(want, dev) if want > 0 else None

    def launch(buf: HitBuffer, stream: int) -> int:
        return lib.nsm_indel_levels_grid(
            li, ls, ri, rs, float(threshold), int(category_mode), flags, buf.records.data_ptr(), buf.capacity,
            buf.count.data_ptr(), ws.data_ptr() if ws is not None else 0, ws.numel() * 8 if ws is not None else 0, float(expected),
            stream,
        )

    hits = run_grid(launch, dev, capacity, "nsm_indel_levels_grid", defer=defer)  # (returns after the stream has been synchronised)
    if ws is not None and return_overflow is not None:
        return_overflow.append(int(ws[1].item()) & 0xFFFFFFFF)
    return hits


def split_workspace(nbytes: int, device) -> Optional[torch.Tensor]:
    """Scratch for ``nsm_indel_levels_grid`` (include/nsm_hip.h): ``nbytes`` rounded down to 8-byte words, at most what
    the device can spare -- less than the library asks for only means more rounds.  None when memory is too tight for a
    useful queue: the grid then runs its single-kernel path."""
    words = int(nbytes) // 8
    try:
        free, _total = torch.cuda.mem_get_info(device)
        words = min(words, int(free * 0.5) // 8)
    except RuntimeError:
        pass
    if words < 128:
        return None
    try:
        return torch.empty(words, dtype=torch.int64, device=device)
    except torch.cuda.OutOfMemoryError:
        return None
