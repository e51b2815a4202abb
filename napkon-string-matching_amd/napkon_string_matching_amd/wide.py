"""Operands beyond the register-resident kernels: items with more than 64 distinct tokens, (joined level) strings of
more than 512 code units, grids whose strings use more than 255 distinct code units.

The reference has no such limits -- ``intersection_vs_union`` builds Python sets of any size
(compare/score_functions.py:10-13) and ``fuzzy_match`` hands strings of any length to rapidfuzz (:27) -- so a drop-in must
not refuse them.  They are rare (a very long option list, a questionnaire in another script), so only the items that
need it leave the fast path: a grid is split into

    regular x regular      the fast kernels (``grid.*_grid``), unchanged
    wide    x all          ``nsm_*_any_grid`` (csrc/any_grids.hip: CSR operands, no pruning, still on the GPU)
    regular x wide         ``nsm_*_any_grid``

and the three hit lists are merged in the canonical order.  The same route takes the items whose LEVELS the fast layouts cannot hold: more than 64 levels (``compare_column="Variable"``
makes one level per character, types/comparable_data.py:283-285,567-574), and -- for ``intersection_vs_union`` -- levels
that are not suffix-nested (the reference re-tokenises every suffix, :287-299: a context-dependent tokenizer can make
level l differ from "level l - 1 plus more").  The general Jaccard kernel then scores every step's two level sets on their
own (``nsm_any_sets.first``), the general fuzzy kernel always did.

Caps of the general kernels (``NotImplementedError`` beyond):
4096 code units per string, 1023 distinct code units per grid, 65535 distinct tokens per item.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import _lib, grid

FAST_TOKENS, FAST_LEN, FAST_ALPHABET = 64, 512, 255
ANY_LEN, ANY_ALPHABET, ANY_IDS = 4096, 1023, 65535
FAST_LEVELS = 64  # levels per item in the fast layouts (and in the general Jaccard kernel's nested layout)


def _dev(array: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(array)).to(device)


def _default_device(device):
    """``device=None`` = the current HIP device (run_grid refuses anything that is not one)."""
    if device is not None:
        return device
    if not torch.cuda.is_available():
        raise _lib.NsmLibraryError("the match loop only runs on an MI355X (HIP device); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _empty_hits() -> grid.Hits:
    return grid.Hits(np.zeros(0, np.float64), np.zeros(0, np.int32), np.zeros(0, np.int32))


def merge(parts) -> grid.Hits:
    """``parts``: (Hits over sub-lists, left index array, right index array).  Indices are mapped back to the full
    lists and the union is put in the canonical order (score descending, i, j ascending)."""
    score = np.concatenate([h.score for h, _, _ in parts]) if parts else np.zeros(0, np.float64)
    i = np.concatenate([np.asarray(li, dtype=np.int64)[h.i] for h, li, _ in parts]) if parts else np.zeros(0, np.int64)
    j = np.concatenate([np.asarray(rj, dtype=np.int64)[h.j] for h, _, rj in parts]) if parts else np.zeros(0, np.int64)
    order = np.lexsort((j, i, -score))
    return grid.Hits(score[order], i[order].astype(np.int32), j[order].astype(np.int32))


def split_grid(wide_l: np.ndarray, wide_r: np.ndarray, fast: Callable, general: Callable) -> grid.Hits:
    """``fast(left_idx, right_idx)`` / ``general(left_idx, right_idx)`` score the sub-grid of those items and return
    Hits relative to the index lists."""
    all_r = np.arange(len(wide_r))
    reg_l, reg_r = np.flatnonzero(~wide_l), np.flatnonzero(~wide_r)
    irr_l, irr_r = np.flatnonzero(wide_l), np.flatnonzero(wide_r)
    parts = []
    if len(reg_l) and len(reg_r):
        parts.append((fast(reg_l, reg_r), reg_l, reg_r))
    if len(irr_l) and len(all_r):
        parts.append((general(irr_l, all_r), irr_l, all_r))
    if len(reg_l) and len(irr_r):
        parts.append((general(reg_l, irr_r), reg_l, irr_r))
    return merge(parts)


# ------------------------------------------------------------------------------------------------ strings
def wide_string_items(items_l: Sequence[Sequence[str]], items_r: Sequence[Sequence[str]]):
    """Which items must leave the fast fuzzy path: a level string longer than 512 code units, or a code unit outside the
    255 most frequent ones of the grid.  Returns (wide_l, wide_r) boolean arrays, or None when nothing is wide."""
    too_long = lambda items: np.fromiter((len(it) > FAST_LEVELS or any(len(s) > FAST_LEN for s in it) for it in items), dtype=bool,
                                         count=len(items))
    wide_l, wide_r = too_long(items_l), too_long(items_r)
    text = "".join(s for items in (items_l, items_r) for it in items for s in it)
    if text:
        points = np.frombuffer(text.encode("utf-32-le"), dtype=np.uint32)
        counts = np.bincount(points)
        uniq = np.flatnonzero(counts)
        if len(uniq) > FAST_ALPHABET:
            keep = set(uniq[np.argsort(-counts[uniq], kind="stable")[:FAST_ALPHABET]].tolist())
            rare = lambda items: np.fromiter((any(ord(ch) not in keep for s in it for ch in s) for it in items), dtype=bool,
                                             count=len(items))
            wide_l, wide_r = wide_l | rare(items_l), wide_r | rare(items_r)
    if not wide_l.any() and not wide_r.any():
        return None
    return wide_l, wide_r


def _any_strings(items: Sequence[Sequence[str]], lut: dict, alphabet: int, device):
    flat = [s for it in items for s in it]
    lengths = np.fromiter((len(s) for s in flat), dtype=np.int64, count=len(flat))
    if len(flat) and int(lengths.max()) > ANY_LEN:
        raise NotImplementedError(f"a string has {int(lengths.max())} code units; the general fuzzy kernel supports {ANY_LEN}")
    offset = np.zeros(len(flat) + 1, dtype=np.int64)
    np.cumsum(lengths, out=offset[1:])
    codes = np.fromiter((lut[ch] for s in flat for ch in s), dtype=np.uint16, count=int(offset[-1])) if len(flat) else \
        np.zeros(0, np.uint16)
    nlev = np.fromiter((len(it) for it in items), dtype=np.int32, count=len(items))
    first = np.zeros(len(items), dtype=np.int32)
    np.cumsum(nlev[:-1], out=first[1:])
    keep = dict(codes=_dev(codes if len(codes) else np.zeros(1, np.uint16), device), offset=_dev(offset, device),
                first=_dev(first, device), nlev=_dev(nlev, device), orig=_dev(np.arange(len(items), dtype=np.int32), device))
    strings = _lib.NsmAnyStrings(keep["codes"].data_ptr(), keep["offset"].data_ptr(), len(flat), alphabet,
                                 int(lengths.max()) if len(flat) else 0)
    return strings, keep


def indel_any_grid(items_l: Sequence[Sequence[str]], items_r: Sequence[Sequence[str]], threshold: float,
                   cat_l: Optional[np.ndarray] = None, cat_r: Optional[np.ndarray] = None, cat_mode: int = _lib.CAT_NONE,
                   raw: bool = False, device=None, capacity: Optional[int] = None) -> grid.Hits:
    """``compare_terms`` x ``fuzzy_match`` (``raw``: the plugin's ratio of the single strings) for pre-processed level
    strings of any length up to 4096 and any alphabet up to 1023 symbols, through ``nsm_indel_any_grid``."""
    if not len(items_l) or not len(items_r):
        return _empty_hits()
    device = _default_device(device)
    symbols = sorted({ch for items in (items_l, items_r) for it in items for s in it for ch in s})
    if len(symbols) > ANY_ALPHABET:
        raise NotImplementedError(f"{len(symbols)} distinct code units in one grid; the general fuzzy kernel supports {ANY_ALPHABET}")
    lut = {ch: k for k, ch in enumerate(symbols)}
    alphabet = max(1, len(symbols))
    lib = _lib.load()
    use_cat = cat_mode != _lib.CAT_NONE and cat_l is not None and cat_r is not None
    sides = []
    for items, cat in ((items_l, cat_l), (items_r, cat_r)):
        strings, keep = _any_strings(items, lut, alphabet, device)
        if use_cat:
            keep["cat"] = _dev(np.asarray(cat, dtype=np.uint64).view(np.int64), device)
        it = _lib.NsmAnyItems(keep["first"].data_ptr(), keep["nlev"].data_ptr(), keep["orig"].data_ptr(),
                              keep["cat"].data_ptr() if use_cat else None, len(items))
        sides.append((it, strings, keep))
    (li, ls, _kl), (ri, rs, _kr) = sides
    flags = _lib.FLAG_RAW_SCORE if raw else 0
    mode = cat_mode if use_cat else _lib.CAT_NONE

    def launch(buf: grid.HitBuffer, stream: int) -> int:
        return lib.nsm_indel_any_grid(li, ls, ri, rs, float(threshold), int(mode), flags, buf.records.data_ptr(), buf.capacity,
                                      buf.count.data_ptr(), stream)

    return grid.run_grid(launch, device, capacity, "nsm_indel_any_grid")


# ------------------------------------------------------------------------------------------------ sets
def _nested(levels: Sequence[Sequence]) -> bool:
    prev: frozenset = frozenset()
    for lv in levels:
        cur = frozenset(lv)
        if not prev <= cur:
            return False
        prev = cur
    return True


def wide_set_items(levels_l: Sequence[Sequence[Sequence]], levels_r: Sequence[Sequence[Sequence]], nesting: bool = False):
    """Items that must leave the fast Jaccard path: a level of more than 64 distinct tokens, more than 64 levels, and (with
    ``nesting``: the caller asks after the fast encoder refused the grid) levels that are not suffix-nested.  None when
    there is none."""
    def irregular(it) -> bool:
        return len(it) > FAST_LEVELS or max((len(set(lv)) for lv in it), default=0) > FAST_TOKENS or (nesting and not _nested(it))

    wide_l = np.fromiter((irregular(it) for it in levels_l), dtype=bool, count=len(levels_l))
    wide_r = np.fromiter((irregular(it) for it in levels_r), dtype=bool, count=len(levels_r))
    if not wide_l.any() and not wide_r.any():
        return None
    return wide_l, wide_r


def _any_sets(items: Sequence[Sequence[Sequence]], vocab: dict, max_levels: int, device, cat, independent: bool):
    """CSR operand of ``nsm_jaccard_any_grid``.  Nested layout: per item its distinct ids sorted by id, each with the first
    level that contains it, and the number of ids per level.  ``independent``: every level is a row of its own (its ids
    sorted by id) -- levels that are not suffix-nested, or more than 64 of them."""
    n = len(items)
    nlev = np.fromiter((len(it) for it in items), dtype=np.int32, count=n)
    keep = dict(nlev=_dev(nlev, device), orig=_dev(np.arange(n, dtype=np.int32), device))
    if cat is not None:
        keep["cat"] = _dev(np.asarray(cat, dtype=np.uint64).view(np.int64), device)
    if independent:
        rows = [np.fromiter(sorted({vocab.setdefault(tok, len(vocab)) for tok in level}), dtype=np.int32) for it in items for level in it]
        lengths = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
        offset = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum(lengths, out=offset[1:])
        most = int(lengths.max(initial=0))
        if most > ANY_IDS:
            raise NotImplementedError(f"a level has {most} distinct tokens; the general Jaccard kernel supports {ANY_IDS}")
        first = np.zeros(n, dtype=np.int32)
        np.cumsum(nlev[:-1], out=first[1:])
        ids = np.concatenate(rows) if rows and int(offset[-1]) else np.zeros(1, np.int32)
        keep.update(ids=_dev(ids, device), offset=_dev(offset, device), first=_dev(first, device))
        st = _lib.NsmAnySets(keep["ids"].data_ptr(), None, keep["offset"].data_ptr(), keep["nlev"].data_ptr(), None,
                             keep["orig"].data_ptr(), keep["cat"].data_ptr() if cat is not None else None, n, 1, most,
                             keep["first"].data_ptr())
        return st, keep
    ids_all: List[np.ndarray] = []
    lv_all: List[np.ndarray] = []
    plen = np.zeros((n, max_levels), dtype=np.int32)
    offset = np.zeros(n + 1, dtype=np.int64)
    for k, levels in enumerate(items):
        first_level: dict = {}
        for lv, level in enumerate(levels):
            cur = {vocab.setdefault(tok, len(vocab)) for tok in level}
            for v in cur:
                first_level.setdefault(v, lv)
            plen[k, lv] = len(cur)
        order = sorted(first_level)
        ids_all.append(np.fromiter(order, dtype=np.int32, count=len(order)))
        lv_all.append(np.fromiter((first_level[v] for v in order), dtype=np.uint8, count=len(order)))
        offset[k + 1] = offset[k] + len(order)
    most = int(np.diff(offset).max(initial=0))
    if most > ANY_IDS:
        raise NotImplementedError(f"an item has {most} distinct tokens; the general Jaccard kernel supports {ANY_IDS}")
    ids = np.concatenate(ids_all) if ids_all else np.zeros(0, np.int32)
    lvs = np.concatenate(lv_all) if lv_all else np.zeros(0, np.uint8)
    keep.update(ids=_dev(ids if len(ids) else np.zeros(1, np.int32), device), lv=_dev(lvs if len(lvs) else np.zeros(1, np.uint8), device),
                offset=_dev(offset, device), plen=_dev(plen, device))
    st = _lib.NsmAnySets(keep["ids"].data_ptr(), keep["lv"].data_ptr(), keep["offset"].data_ptr(), keep["nlev"].data_ptr(),
                         keep["plen"].data_ptr(), keep["orig"].data_ptr(), keep["cat"].data_ptr() if cat is not None else None,
                         n, max_levels, most, None)
    return st, keep


def jaccard_any_grid(levels_l: Sequence[Sequence[Sequence]], levels_r: Sequence[Sequence[Sequence]], threshold: float,
                     cat_l: Optional[np.ndarray] = None, cat_r: Optional[np.ndarray] = None, cat_mode: int = _lib.CAT_NONE,
                     raw: bool = False, device=None, capacity: Optional[int] = None) -> grid.Hits:
    """``compare_terms`` x ``intersection_vs_union`` (``raw``: the plugin's quotient of the single sets) for items of any
    number of distinct tokens up to 65535, through ``nsm_jaccard_any_grid``."""
    if not len(levels_l) or not len(levels_r):
        return _empty_hits()
    device = _default_device(device)
    deepest = max(max((len(it) for it in levels_l), default=1), max((len(it) for it in levels_r), default=1), 1)
    # the nested layout (one merge per pair) when every item is suffix-nested and at most 64 levels deep; else every level a
    # row of its own (one merge per step), on both sides alike
    independent = deepest > FAST_LEVELS or not all(_nested(it) for items in (levels_l, levels_r) for it in items)
    use_cat = cat_mode != _lib.CAT_NONE and cat_l is not None and cat_r is not None
    vocab: dict = {}
    lt, _kl = _any_sets(levels_l, vocab, deepest, device, cat_l if use_cat else None, independent)
    rt, _kr = _any_sets(levels_r, vocab, deepest, device, cat_r if use_cat else None, independent)
    lib = _lib.load()
    flags = _lib.FLAG_RAW_SCORE if raw else 0
    mode = cat_mode if use_cat else _lib.CAT_NONE

    def launch(buf: grid.HitBuffer, stream: int) -> int:
        return lib.nsm_jaccard_any_grid(lt, rt, float(threshold), int(mode), flags, buf.records.data_ptr(), buf.capacity,
                                        buf.count.data_ptr(), stream)

    return grid.run_grid(launch, device, capacity, "nsm_jaccard_any_grid")
