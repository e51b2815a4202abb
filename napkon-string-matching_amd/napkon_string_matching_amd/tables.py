"""Host encoders: per-item operands -> HBM tables the HIP kernels read (include/nsm_hip.h).

Everything here is per-ITEM work (O(N + M)); the per-PAIR work is on the GPU.  The reference does
the equivalent per-item preparation in ``ComparableData.gen_comp_value`` / ``tokenize``
(types/comparable_data.py:283-299) and -- wastefully, once per pair -- in ``join_sorted`` and
``set(...)`` (compare/score_functions.py:10-11,16-17,24-25).

Layouts (one row per item, rows sorted by size descending so that a wavefront's 64 rows have the
same size class; ``orig`` maps a row back to the caller's item index):

* ``SetTable``  int32 ids [n][W], W in {16, 32, 64}; unused slots hold -1 (left) / -2 (right) so
  that padding never compares equal; ``cnt``; a 64-bit signature per row for the exact prune.
  In levels mode the ids are stored in "suffix-nested" order: level l of the item is the first
  ``plen[l]`` ids, so ONE equality matrix per pair serves every level.
* ``StrTable``  uint8 codes [n][64] over a dense per-corpus alphabet (<= 255 symbols), padded with
  the code ``alphabet``; ``len``.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, Hashable, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

WIDTHS = (16, 32, 64)
STRIDES = (64, 128, 256, 512)  # code units per string row: 1, 2, 4 or 8 words of the bit-parallel LCS
MAX_LEVELS = 64
MAX_INDEX_VOCAB = 1 << 23  # most index keys (ids; x 64 with a category partition) for which a right table gets its global inverted index by default (20 bytes of offsets per key)
LEFT_PAD, RIGHT_PAD = -1, -2
EMPTY_CATEGORY_BIT = 63  # stands for "no category at all" when empty-vs-empty counts as a match
_GOLDEN = np.uint32(0x9E3779B1)
_GOLDEN2 = np.uint32(0xC2B2AE35)


class IrregularLevels(NotImplementedError):
    """An item the suffix-nested fast layout cannot hold (levels that are not nested, or more than 64 of them): the
    host routes such items -- and only them -- through the general kernels (wide.py)."""


def _id_limit(orig, n: int) -> int:
    """An exclusive upper bound of the ids a table reports: ``n`` for the default ``arange(n)``, else max(orig) + 1;
    0 (= unknown: the sort keeps all 32 bits) when an id is negative."""
    if orig is None:
        return int(n)
    o = np.asarray(orig)
    if o.size == 0:
        return int(n)
    return 0 if int(o.min()) < 0 else int(o.max()) + 1


COMPACT_POSTINGS = True  # (tests switch it off to cover the 64-bit entries at small sizes)
RAW_POST_FORMAT = 2      # compact posting entries of RAW tables: 2 = with the signature fold, 1 = 32 bits (A/B runs)


def post_row_bits(rows: int, width: int) -> int:
    """Row bits of the 32-bit posting entry (row | position << bits | (cnt - 1) << (bits + log2 width)), 0 when the
    table's rows do not fit beside the two width-sized fields: 64-bit entries then (include/nsm_hip.h: post)."""
    bits = 32 - 2 * (int(width).bit_length() - 1)
    return bits if (COMPACT_POSTINGS and rows <= (1 << bits)) else 0


def sig_fold27(sig: np.ndarray) -> np.ndarray:
    """27-bit fold of signature words (bit j = OR of the hash bits j, j + 27, j + 54): what posting entries of format 2 carry."""
    s = np.asarray(sig, dtype=np.uint64)
    m = np.uint64((1 << 27) - 1)
    return ((s & m) | ((s >> np.uint64(27)) & m) | ((s >> np.uint64(54)) & np.uint64(0xF))).astype(np.uint32)


def inverted_index(ids: np.ndarray, cnt: np.ndarray, vocab: int, seg: Optional[np.ndarray] = None, row_bits: int = 0,
                   sig: Optional[np.ndarray] = None):
    """The global inverted index of a set table (include/nsm_hip.h: post / post_start / post_sq), the numpy way --
    what ``nsm_build_set_table`` builds on the GPU, byte for byte.  ``ids`` [n][W] (RAW: ascending per row), ``cnt`` [n];
    ``seg`` [n]: a partitioned levels table keys its postings by (category segment, id).  Entry format (``post_format``):
    0 = 64 bits (``row_bits`` 0); 1 = 32 bits (``row_bits`` > 0); 2 = 64 bits, the 32-bit entry with the fold of the row's
    signature word above it (``row_bits`` > 0 and ``sig`` given)."""
    n, width = ids.shape
    valid = np.arange(width, dtype=np.int64)[None, :] < np.asarray(cnt, dtype=np.int64)[:, None]
    r_idx, k_idx = np.nonzero(valid)  # row-major: the stable sort below keeps rows ascending inside one (id, position)
    tok = ids[valid].astype(np.int64)
    if len(tok) and int(tok.max()) >= vocab:
        raise ValueError("an id is >= vocab")
    if seg is not None:
        tok = np.asarray(seg, dtype=np.int64)[r_idx] * vocab + tok
        vocab = 64 * vocab
    order = np.lexsort((k_idx, tok))
    if row_bits > 0:
        wl = int(width).bit_length() - 1
        entry = (r_idx.astype(np.uint32) | (k_idx.astype(np.uint32) << np.uint32(row_bits)) |
                 ((np.asarray(cnt, dtype=np.uint32)[r_idx] - np.uint32(1)) << np.uint32(row_bits + wl)))
        post = np.zeros(n * width, dtype=np.uint32)
        if sig is not None:
            entry = entry.astype(np.uint64) | (sig_fold27(sig)[r_idx].astype(np.uint64) << np.uint64(32))
            post = np.zeros(n * width, dtype=np.uint64)
    else:
        entry = (r_idx.astype(np.uint64) | (k_idx.astype(np.uint64) << np.uint64(32)) |
                 (np.asarray(cnt, dtype=np.uint64)[r_idx] << np.uint64(40)))
        post = np.zeros(n * width, dtype=np.uint64)
    post[: len(order)] = entry[order]
    k_s = k_idx[order]
    cls = np.where(k_s < 1, 0, np.where(k_s < 2, 1, np.where(k_s < 4, 2, np.where(k_s < 8, 3, 4))))
    ckey = tok[order] * 5 + cls
    post_start = np.searchsorted(ckey, np.arange(5 * vocab + 1, dtype=np.int64), side="left").astype(np.int32)
    base = post_start[:-1:5].astype(np.int64)
    post_sq = tuple(int(((post_start[c + 1:: 5].astype(np.int64) - base) ** 2).sum()) for c in range(5))
    return post, post_start, post_sq


def levels_index_vocab(ids: np.ndarray, side: str, categories, category_mode: int, partition: bool, index: Optional[bool]) -> int:
    """``vocab`` of a levels table's global inverted index, 0 = none: built for right tables by default, while the offsets
    (5 per key; 64 key spaces with a category partition) stay below ``MAX_INDEX_VOCAB`` keys."""
    if index is False or (index is None and side != "right"):
        return 0
    vocab = int(np.asarray(ids).max(initial=-1)) + 1
    parts = 64 if (partition and categories is not None and category_mode != _lib.CAT_NONE) else 1
    if vocab < 1 or (index is None and (vocab * parts > MAX_INDEX_VOCAB or np.asarray(ids).size >= (1 << 31))):
        return 0
    return vocab


def pick_width(*max_counts: int) -> int:
    need = max([1, *max_counts])
    for w in WIDTHS:
        if need <= w:
            return w
    raise NotImplementedError(
        f"an item has {need} distinct tokens; the HIP kernels hold one row in at most {WIDTHS[-1]} registers"
    )


def pick_stride(max_len: int) -> int:
    for stride in STRIDES:
        if max_len <= stride:
            return stride
    raise NotImplementedError(
        f"a string has {max_len} code units; the Indel kernels support up to {STRIDES[-1]} (8 words of 64 bits)"
    )


def partition_allowed(category_mode: int, *category_masks: Optional[np.ndarray]) -> bool:
    """Whether BOTH sides of a levels grid can be partitioned by category.  With the "both empty also
    matches" predicate the empty items become one more category (bit 63), which needs that bit to be
    free on both sides -- a joint property, so the caller decides once and encodes both sides alike."""
    if category_mode == _lib.CAT_NONE or any(m is None for m in category_masks):
        return False
    if category_mode == _lib.CAT_INTERSECT_OR_BOTH_EMPTY:
        return not any(len(m) and int(np.asarray(m, dtype=np.uint64).max()) >> EMPTY_CATEGORY_BIT for m in category_masks)
    return True


def signatures(ids: np.ndarray, cnt: np.ndarray, multiplier: np.uint32 = _GOLDEN) -> np.ndarray:
    """Signature word per row (include/nsm_hip.h): 58 hash bits; the top 6 bits hold, in UNARY, the
    number c of ids that collided with an earlier id of the same row, so that with the other side's
    top bits forced to one popcount(a & b) = popcount(common hash bits) + c >= |A n B|.  A row with
    c > 6 is all ones (always passes; both sides encode it that way)."""
    n, w = ids.shape
    if not n:
        return np.zeros(0, np.uint64)
    valid = np.arange(w, dtype=np.int32)[None, :] < cnt[:, None]
    h16 = ((ids.astype(np.uint32) * multiplier) >> np.uint32(16)) & np.uint32(0xFFFF)
    pos = ((h16.astype(np.uint64) * np.uint64(58)) >> np.uint64(16)).astype(np.uint64)
    bits = np.where(valid, np.uint64(1) << pos, np.uint64(0))
    word = np.bitwise_or.reduce(bits, axis=1)
    extra = cnt.astype(np.int64) - np.bitwise_count(word).astype(np.int64)
    unary = ((np.uint64(1) << np.minimum(extra, 6).astype(np.uint64)) - np.uint64(1)) << np.uint64(58)
    return np.where(extra > 6, np.uint64(0xFFFFFFFFFFFFFFFF), word | unary)


def _dev(array: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(array)).to(device)


def _on_gpu(device) -> bool:
    """Tables for a HIP device are built by the library's device-side builders (nsm_build_*: derived columns,
    sorts, category partition); the numpy encoders below remain for host-side tables (CPU tests, and as the
    parity reference of the builders: tests/test_gpu_builders.py)."""
    return torch.device(device).type == "cuda"


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream(device) -> int:
    return torch.cuda.current_stream(torch.device(device)).cuda_stream


class Vocabulary:
    """Equality-preserving token -> dense int32 id map shared by both sides of a grid."""

    def __init__(self) -> None:
        self._ids: Dict[Hashable, int] = {}

    def id(self, token: Hashable) -> int:
        got = self._ids.get(token)
        if got is None:
            got = len(self._ids)
            self._ids[token] = got
        return got

    def __len__(self) -> int:
        return len(self._ids)


class LevelPool:
    """Encoded levels of items that take part in SEVERAL grids (k cohorts: every item is in k - 1 of them):
    one shared vocabulary, every item encoded once -- by the identity of its level list -- into pooled
    arrays of the widest layout (64 ids, 64 levels); a grid gathers its rows from the pool."""

    def __init__(self) -> None:
        self.vocab = Vocabulary()
        self._slot: Dict[int, int] = {}
        self._keep: List[object] = []  # keeps the level lists alive, so their ids cannot be reused
        self._ids = np.zeros((0, WIDTHS[-1]), dtype=np.int32)
        self._plen = np.zeros((0, MAX_LEVELS), dtype=np.uint8)
        self._nlev = np.zeros(0, dtype=np.int32)

    def rows(self, items: Sequence[Sequence[Iterable[Hashable]]]):
        """(ids [n][64], plen [n][64], nlev [n]) of ``items``; unseen items are encoded and pooled."""
        return self.rows_keyed(items, None)

    def rows_keyed(self, keys: Sequence[object], convert=None):
        """As ``rows``, with the pool keyed by the identity of ``keys[k]``; the levels to encode for a NEW key
        are ``convert(key)`` (the caller's token-list view of the item), only evaluated for new keys."""
        slot = self._slot
        fresh = list({id(k): k for k in keys if id(k) not in slot}.values())
        if fresh:
            todo = fresh if convert is None else [convert(k) for k in fresh]
            ids, plen, nlev, _, _ = SetTable.encode_levels(todo, self.vocab, width=WIDTHS[-1], max_levels=MAX_LEVELS)
            base = len(self._keep)
            for k, it in enumerate(fresh):
                slot[id(it)] = base + k
            self._keep.extend(fresh)
            self._ids = np.concatenate([self._ids, ids])
            self._plen = np.concatenate([self._plen, plen])
            self._nlev = np.concatenate([self._nlev, nlev])
        idx = np.fromiter((slot[id(k)] for k in keys), dtype=np.int64, count=len(keys))
        return self._ids[idx], self._plen[idx], self._nlev[idx]


@dataclass
class SetTable:
    ids: torch.Tensor
    cnt: torch.Tensor
    sig: torch.Tensor
    orig: torch.Tensor
    side: str
    width: int
    n: int
    has_empty: bool
    sig2: Optional[torch.Tensor] = None
    size_start: Optional[torch.Tensor] = None
    nlev: Optional[torch.Tensor] = None
    plen: Optional[torch.Tensor] = None
    cat: Optional[torch.Tensor] = None
    filt: Optional[torch.Tensor] = None
    max_levels: int = 0
    seg: Optional[torch.Tensor] = None
    seg_start: Optional[torch.Tensor] = None
    category_mode: Optional[int] = None  # the mode the levels kernel must be called with (set by the encoder)
    # global inverted index (RAW right tables; include/nsm_hip.h): postings sorted by (id, position), 5 offsets per id
    post: Optional[torch.Tensor] = None
    post_start: Optional[torch.Tensor] = None
    vocab: int = 0
    post_sq: Tuple[int, ...] = (0, 0, 0, 0, 0)
    post_row_bits: int = 0  # > 0: compact posting entries with this many row bits (``post_row_bits(rows, width)``)
    post_format: int = 0  # 0: 64-bit entries; 1: 32-bit (levels tables); 2: 64-bit with the signature fold (RAW tables)
    id_limit: int = 0  # every ``orig`` entry is below it (the sort's key width, nsm_sort_hits); 0 = unknown

    # ------------------------------------------------------------------ builders
    @classmethod
    def from_padded(
        cls,
        ids: np.ndarray,
        side: str,
        device,
        width: Optional[int] = None,
        orig: Optional[np.ndarray] = None,
        validate: bool = True,
        index: Optional[bool] = None,
    ) -> "SetTable":
        """RAW table from an int array [n][w]; negative entries are padding, the rest must be
        unique per row (``validate`` checks).  ``index``: build the global inverted index (None = for right tables
        whose ids stay below ``MAX_INDEX_VOCAB``): ``nsm_jaccard_raw_grid`` then generates candidate pairs from it
        wherever that is cheaper than visiting all N x M."""
        ids = np.asarray(ids)
        if ids.ndim != 2:
            raise ValueError("ids must be [n][w]")
        n, w_in = ids.shape
        valid = ids >= 0
        cnt = valid.sum(axis=1).astype(np.int32)
        width = width or pick_width(int(cnt.max()) if n else 1)
        if n and int(cnt.max()) > width:
            raise ValueError(f"row with {int(cnt.max())} ids does not fit width {width}")
        if w_in < 2 or bool((valid[:, :-1] >= valid[:, 1:]).all()):  # already packed: valid ids first
            packed = ids[:, : min(w_in, width)].astype(np.int32)
        else:
            order = np.argsort(~valid, axis=1, kind="stable")  # valid ids first, input order kept
            packed = np.take_along_axis(ids, order, axis=1)[:, : min(w_in, width)].astype(np.int32)
        if packed.shape[1] < width:
            packed = np.pad(packed, ((0, 0), (0, width - packed.shape[1])), constant_values=-1)
        if validate and n:
            srt = np.sort(np.where(packed >= 0, packed, -np.arange(1, width + 1, dtype=np.int64)[None, :]), axis=1)
            if (srt[:, 1:] == srt[:, :-1]).any():
                raise ValueError("duplicate id inside a row: sets must be de-duplicated before encoding")
        vocab = int(packed.max(initial=-1)) + 1
        if index is None:
            index = side == "right" and 0 < vocab <= MAX_INDEX_VOCAB and n * width < (1 << 31)
        return cls._finish(packed, cnt, side, device, width, orig, index_vocab=vocab if index else 0)

    @classmethod
    def from_rows(
        cls, rows: Sequence[Iterable[Hashable]], side: str, device, vocab: Vocabulary, width: Optional[int] = None
    ) -> "SetTable":
        """RAW table from Python collections of hashable tokens (duplicates collapse like ``set``)."""
        uniq: List[List[int]] = []
        for row in rows:
            seen: Dict[int, None] = {}
            for tok in row:
                seen.setdefault(vocab.id(tok), None)
            uniq.append(list(seen))
        width = width or pick_width(max((len(u) for u in uniq), default=1))
        ids = np.full((len(uniq), width), -1, dtype=np.int32)
        for k, u in enumerate(uniq):
            if len(u) > width:
                raise ValueError(f"row {k} has {len(u)} ids > width {width}")
            ids[k, : len(u)] = u
        return cls.from_padded(ids, side, device, width=width, validate=False)

    @classmethod
    def from_levels(
        cls,
        items: Sequence[Sequence[Iterable[Hashable]]],
        side: str,
        device,
        vocab: Vocabulary,
        width: Optional[int] = None,
        categories: Optional[np.ndarray] = None,
        category_mode: int = _lib.CAT_NONE,
        partition: bool = True,
        index: Optional[bool] = None,
    ) -> "SetTable":
        """Levels table.  ``items[k]`` is the level list of item k (``gen_comp_value`` output,
        types/comparable_data.py:283-285): level l must contain level l-1 (suffix nesting).  ``index``: build the global
        inverted index (None = for right tables, when its offsets stay small -- ``levels_index_vocab``)."""
        ids, plen, nlev, max_levels, width = cls.encode_levels(items, vocab, width)
        cnt = (ids >= 0).sum(axis=1).astype(np.int32)
        return cls._finish(ids, cnt, side, device, width, None, nlev=nlev, plen=plen, cat=categories,
                           max_levels=max_levels, category_mode=category_mode, partition=partition,
                           index_vocab=levels_index_vocab(ids, side, categories, category_mode, partition, index))

    @staticmethod
    def encode_levels(items: Sequence[Sequence[Iterable[Hashable]]], vocab: Vocabulary, width: Optional[int] = None,
                      max_levels: Optional[int] = None):
        """The per-item part of ``from_levels``: (ids [n][width], plen [n][max_levels] uint8, nlev [n] int32,
        max_levels, width) in suffix-nested layout; raises for items that are not nested."""
        n = len(items)
        nlev = np.fromiter((len(levels) for levels in items), dtype=np.int64, count=n)
        if n and int(nlev.max()) > MAX_LEVELS:
            k = int(np.argmax(nlev > MAX_LEVELS))
            raise IrregularLevels(f"item {k} has {int(nlev[k])} levels > {MAX_LEVELS}")
        n_levels = int(nlev.sum())
        level_len = np.fromiter((len(level) for levels in items for level in levels), dtype=np.int64, count=n_levels)
        ids_of = vocab._ids  # token -> id, new tokens numbered in order of first appearance
        flat = np.fromiter(
            (ids_of.setdefault(tok, len(ids_of)) for levels in items for level in levels for tok in level),
            dtype=np.int64, count=int(level_len.sum()))
        # per token occurrence: its item and its level inside the item
        level_item = np.repeat(np.arange(n, dtype=np.int64), nlev)
        level_start = np.zeros(n, dtype=np.int64)
        np.cumsum(nlev[:-1], out=level_start[1:])
        level_in_item = np.arange(n_levels, dtype=np.int64) - np.repeat(level_start, nlev)
        tok_item = np.repeat(level_item, level_len)
        tok_level = np.repeat(level_in_item, level_len)
        span = max(1, len(ids_of))
        # unique ids of an item in first-appearance order (= level order): one row, level l = a prefix
        _, first = np.unique(tok_item * span + flat, return_index=True)
        first.sort()
        u_item, u_id, u_level = tok_item[first], flat[first], tok_level[first]
        per_item = np.bincount(u_item, minlength=n)[:n]
        item_start = np.zeros(n, dtype=np.int64)
        np.cumsum(per_item[:-1], out=item_start[1:])
        rank = np.arange(len(first), dtype=np.int64) - item_start[u_item]
        if n and int(per_item.max()) > WIDTHS[-1]:
            pick_width(int(per_item.max()))  # raises: more distinct tokens than the kernels hold
        width = width or pick_width(int(per_item.max()) if n else 1)
        if n and int(per_item.max()) > width:
            k = int(np.argmax(per_item > width))
            raise ValueError(f"item {k} has {int(per_item[k])} ids > width {width}")
        ids = np.full((n, width), -1, dtype=np.int32)
        ids[u_item, rank] = u_id
        max_levels = max_levels or max(4, -(-(int(nlev.max()) if n else 1) // 4) * 4)
        # plen[k][l] = ids first seen at a level <= l; levels past the item's last repeat its last value
        new_at = np.zeros((n, max_levels), dtype=np.int64)
        np.add.at(new_at, (u_item, u_level), 1)
        plen_full = np.cumsum(new_at, axis=1)
        # suffix nesting: level l must hold every id seen before it, i.e. its own distinct ids == plen[l]
        distinct = np.zeros((n, max_levels), dtype=np.int64)
        if len(flat):
            lev_key = np.unique((tok_item * max_levels + tok_level) * span + flat) // span
            np.add.at(distinct, (lev_key // max_levels, lev_key % max_levels), 1)
        live = np.arange(max_levels)[None, :] < nlev[:, None]
        bad = live & (distinct != plen_full)
        if bad.any():
            k, lv = (int(v[0]) for v in np.nonzero(bad))
            raise IrregularLevels(
                f"item {k}: level {lv} does not contain level {lv - 1}; the suffix-nested fast layout (what "
                "gen_comp_value produces with a whitespace tokenizer) cannot hold it")
        return ids, plen_full.astype(np.uint8), nlev.astype(np.int32), max_levels, width

    @classmethod
    def from_nested_arrays(
        cls, ids: np.ndarray, plen: np.ndarray, nlev: np.ndarray, side: str, device,
        categories: Optional[np.ndarray] = None, width: Optional[int] = None,
        category_mode: int = _lib.CAT_NONE, partition: bool = True, orig: Optional[np.ndarray] = None,
        index: Optional[bool] = None,
    ) -> "SetTable":
        """Levels table from arrays that already are in suffix-nested layout: ``ids`` [n][w] unique
        per row (negative = padding, valid ids first), ``plen`` [n][L] non-decreasing prefix lengths,
        ``nlev`` [n] (vectorised path for large synthetic cohorts).  ``orig`` = the item ids reported in
        hits (default 0..n-1; a rank's block of a sharded left side passes its global row numbers)."""
        ids = np.asarray(ids, dtype=np.int32)
        n, w_in = ids.shape
        cnt = (ids >= 0).sum(axis=1).astype(np.int32)
        width = width or pick_width(int(cnt.max()) if n else 1)
        if w_in < width:
            ids = np.pad(ids, ((0, 0), (0, width - w_in)), constant_values=-1)
        elif w_in > width:
            if n and int(cnt.max()) > width:
                raise ValueError("row does not fit the requested width")
            ids = ids[:, :width]
        plen = np.asarray(plen, dtype=np.uint8)
        max_levels = -(-plen.shape[1] // 4) * 4
        if plen.shape[1] < max_levels:  # pad with the last value (clamped level index)
            plen = np.concatenate([plen, np.repeat(plen[:, -1:], max_levels - plen.shape[1], axis=1)], axis=1)
        return cls._finish(ids, cnt, side, device, width, orig, nlev=np.asarray(nlev, dtype=np.int32), plen=plen,
                           cat=categories, max_levels=max_levels, category_mode=category_mode, partition=partition,
                           index_vocab=levels_index_vocab(ids, side, categories, category_mode, partition, index))

    @classmethod
    def _finish(cls, ids, cnt, side, device, width, orig, nlev=None, plen=None, cat=None, max_levels=0,
                category_mode=_lib.CAT_NONE, partition=False, index_vocab=0):
        if side not in ("left", "right"):
            raise ValueError("side must be 'left' or 'right'")
        if nlev is not None:  # entries of plen past an item's last level repeat it (the clamped level index)
            plen = np.asarray(plen, dtype=np.uint8)
            clamp = np.minimum(np.arange(plen.shape[1])[None, :], np.maximum(np.asarray(nlev), 1)[:, None] - 1)
            plen = np.take_along_axis(plen, clamp, axis=1)
        if _on_gpu(device):
            return cls._finish_device(ids, cnt, side, device, width, orig, nlev, plen, cat, max_levels, category_mode, partition,
                                      index_vocab)
        n = ids.shape[0]
        ids = ids.copy()
        pad = LEFT_PAD if side == "left" else RIGHT_PAD
        if nlev is None and n:  # RAW rows: ids ascending (the global token order of the inverted index)
            big = np.iinfo(np.int32).max
            ids[np.arange(width, dtype=np.int32)[None, :] >= cnt[:, None]] = big
            ids.sort(axis=1)
        ids[np.arange(width, dtype=np.int32)[None, :] >= cnt[:, None]] = pad
        base = np.arange(n, dtype=np.int32) if orig is None else np.asarray(orig, dtype=np.int32)
        id_limit = _id_limit(orig, n)  # (n = the caller's items here; a partition changes it below)
        seg = seg_start = None
        mode = category_mode if (nlev is not None and cat is not None) else _lib.CAT_NONE
        if cat is not None:
            cat = np.asarray(cat, dtype=np.uint64).copy()
        if mode != _lib.CAT_NONE and partition:
            # Category partition (see encode_level_strings): an item with k categories becomes k rows,
            # rows are grouped per category; "both empty" becomes one more category of the empty items.
            if mode == _lib.CAT_INTERSECT_OR_BOTH_EMPTY:
                if not partition_allowed(mode, cat):
                    raise ValueError(
                        "category bit 63 is in use, so the empty items cannot become a category of their own: "
                        "encode BOTH sides with partition=False (tables.partition_allowed decides for a pair)")
                cat[cat == 0] = np.uint64(1) << np.uint64(EMPTY_CATEGORY_BIT)
                mode = _lib.CAT_INTERSECT
        if mode != _lib.CAT_NONE and partition:
            rows, segs = [], []
            for c in range(64):
                has = np.flatnonzero((cat >> np.uint64(c)) & np.uint64(1))
                rows.append(has)
                segs.append(np.full(len(has), c, dtype=np.int32))
            rows = np.concatenate(rows).astype(np.int64)
            seg = np.concatenate(segs)
            order = np.lexsort((-cnt[rows], seg))  # per category: larger sets first
            perm, seg = rows[order], seg[order]
            seg_start = np.zeros(65, dtype=np.int32)
            seg_start[1:] = np.cumsum(np.bincount(seg, minlength=64)[:64])
            n = len(perm)
        else:
            perm = np.argsort(-cnt, kind="stable")  # size descending, ties by input order
        ids, cnt_s = ids[perm], cnt[perm]
        # rows of size (width - c) occupy [size_start[c], size_start[c + 1]) in the sorted table
        per_size = np.bincount(cnt_s, minlength=width + 1)[: width + 1]
        size_start = np.zeros(width + 2, dtype=np.int32)
        size_start[1:] = np.cumsum(per_size[::-1])
        sig1 = signatures(ids, cnt_s)
        filt = None
        if nlev is not None:  # levels table: one 32-byte filter record per row
            filt = np.zeros((n, 8), dtype=np.uint32)
            filt[:, 0] = (sig1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            filt[:, 1] = (sig1 >> np.uint64(32)).astype(np.uint32)
            if cat is not None:
                c = cat[perm]
                filt[:, 2] = (c & np.uint64(0xFFFFFFFF)).astype(np.uint32)
                filt[:, 3] = (c >> np.uint64(32)).astype(np.uint32)
            plen1 = plen[perm][:, 1].astype(np.int32)  # size of the step-1 set (plen is padded with its last value)
            filt[:, 4] = plen1.astype(np.uint32) | (cnt_s.astype(np.uint32) << 8) | (nlev[perm].astype(np.uint32) << 16)
            sig_l1 = signatures(ids, np.minimum(plen1, cnt_s))
            filt[:, 5] = (sig_l1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            filt[:, 6] = (sig_l1 >> np.uint64(32)).astype(np.uint32)
        post = post_start = None
        post_sq = (0, 0, 0, 0, 0)
        row_bits = post_row_bits(n, width) if index_vocab else 0
        fmt = 0 if not row_bits else (1 if nlev is not None else RAW_POST_FORMAT)
        if index_vocab:
            post, post_start, post_sq = inverted_index(ids, cnt_s, index_vocab, seg=seg, row_bits=row_bits,
                                                       sig=sig1 if fmt == 2 else None)
        return cls(
            post=None if post is None else _dev(post.view(np.int32 if fmt == 1 else np.int64), device),
            post_row_bits=row_bits, post_format=fmt,
            post_start=None if post_start is None else _dev(post_start, device), vocab=int(index_vocab), post_sq=post_sq,
            filt=None if filt is None else _dev(filt, device),
            ids=_dev(ids, device),
            cnt=_dev(cnt_s, device),
            sig=_dev(sig1, device),
            sig2=_dev(signatures(ids, cnt_s, _GOLDEN2), device),
            orig=_dev(base[perm], device),
            side=side,
            width=width,
            n=n,
            has_empty=bool(n and cnt.min() == 0),
            size_start=_dev(size_start, device),
            nlev=None if nlev is None else _dev(nlev[perm], device),
            plen=None if plen is None else _dev(plen[perm], device),
            cat=None if cat is None else _dev(cat[perm], device),
            max_levels=max_levels,
            seg=None if seg is None else _dev(seg, device),
            seg_start=None if seg_start is None else _dev(seg_start, device),
            category_mode=mode if nlev is not None else None,
            id_limit=id_limit,
        )

    @classmethod
    def _finish_device(cls, ids, cnt, side, device, width, orig, nlev, plen, cat, max_levels, category_mode, partition,
                       index_vocab=0):
        """The same table as ``_finish``, built on the GPU by ``nsm_build_set_table`` from the packed ids."""
        lib = _lib.load()
        n = ids.shape[0]
        levels = nlev is not None
        mode = category_mode if (levels and cat is not None) else _lib.CAT_NONE
        do_part = mode != _lib.CAT_NONE and partition
        rows, out_mode = n, mode
        if cat is not None:
            cat = np.asarray(cat, dtype=np.uint64)
        if do_part:
            c = cat.copy()
            if mode == _lib.CAT_INTERSECT_OR_BOTH_EMPTY:
                if not partition_allowed(mode, cat):
                    raise ValueError(
                        "category bit 63 is in use, so the empty items cannot become a category of their own: "
                        "encode BOTH sides with partition=False (tables.partition_allowed decides for a pair)")
                c[c == 0] = np.uint64(1) << np.uint64(EMPTY_CATEGORY_BIT)
                out_mode = _lib.CAT_INTERSECT
            rows = int(np.bitwise_count(c).sum())
        dev = torch.device(device)
        cap = max(rows, 1)  # (an empty tensor has no data pointer: columns are allocated with one row at least)
        new = lambda shape, dtype: torch.empty(shape, dtype=dtype, device=dev)
        t = cls(
            ids=new((cap, width), torch.int32), cnt=new(cap, torch.int32),
            sig=new(cap, torch.int64), orig=new(cap, torch.int32), side=side, width=width,
            n=rows, has_empty=bool(n and int(np.min(cnt)) == 0), sig2=new(cap, torch.int64),
            size_start=new(width + 2, torch.int32),
            nlev=new(cap, torch.int32) if levels else None,
            plen=new((cap, max_levels), torch.uint8) if levels else None,
            cat=new(cap, torch.int64) if (levels and cat is not None) else None,
            filt=new((cap, 8), torch.int32) if levels else None,
            max_levels=max_levels, seg=new(cap, torch.int32) if do_part else None,
            seg_start=new(65, torch.int32) if do_part else None, category_mode=out_mode if levels else None,
            post=new(cap * width, torch.int32 if (post_row_bits(rows, width) and (levels or RAW_POST_FORMAT == 1)) else torch.int64) if index_vocab else None,
            post_row_bits=post_row_bits(rows, width) if index_vocab else 0,
            post_format=(0 if not post_row_bits(rows, width) else 1 if levels else RAW_POST_FORMAT) if index_vocab else 0,
            post_start=new(5 * index_vocab * (64 if do_part else 1) + 1, torch.int32) if index_vocab else None,
            vocab=int(index_vocab),
        )
        d_ids = _dev(np.asarray(ids, dtype=np.int32), dev)
        d_nlev = _dev(np.asarray(nlev, dtype=np.int32), dev) if levels else None
        d_plen = _dev(np.asarray(plen, dtype=np.uint8), dev) if levels else None
        d_cat = _dev(cat.view(np.int64), dev) if (levels and cat is not None) else None
        d_orig = None if orig is None else _dev(np.asarray(orig, dtype=np.int32), dev)
        st = t.struct()
        flags = _lib.BUILD_PARTITION if do_part else 0
        _lib.check(lib.nsm_build_set_table(_ptr(d_ids), n, ids.shape[1], 0 if side == "left" else 1, _ptr(d_nlev), _ptr(d_plen),
                                           _ptr(d_cat), _ptr(d_orig), mode, flags, st, _stream(dev)), "nsm_build_set_table")
        if st.n != rows:
            raise _lib.NsmLibraryError(f"nsm_build_set_table built {st.n} rows, expected {rows}")
        t.post_sq = tuple(int(v) for v in st.post_sq)
        t.id_limit = _id_limit(orig, n)
        for col in ("ids", "cnt", "sig", "sig2", "orig", "nlev", "plen", "cat", "filt", "seg"):
            if getattr(t, col) is not None:
                setattr(t, col, getattr(t, col)[:rows])
        return t

    # ------------------------------------------------------------------ C view
    def struct(self) -> _lib.NsmSetTable:
        def ptr(t):
            return None if t is None else t.data_ptr()

        return _lib.NsmSetTable(
            ptr(self.ids), ptr(self.cnt), ptr(self.sig), ptr(self.sig2), ptr(self.orig), ptr(self.size_start),
            ptr(self.nlev),
            ptr(self.plen),
            ptr(self.cat), ptr(self.filt), ptr(self.seg), ptr(self.seg_start), self.n, self.width, self.max_levels,
            self.vocab, ptr(self.post), ptr(self.post_start), (ctypes.c_uint64 * 5)(*self.post_sq), self.post_row_bits,
            self.post_format,
        )

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.ids, self.cnt, self.sig, self.orig))


class Alphabet:
    """Dense code units shared by both sides (<= 255 distinct symbols; the pad code is ``size``).

    Built once from every string of a grid.  Codes are assigned by DESCENDING frequency: the match
    masks of symbols c and c' only collide in LDS when c == c' (mod 32), so the 32 most frequent
    symbols never conflict with each other.
    """

    def __init__(self, strings: Iterable[str] = ()) -> None:
        text = "".join(strings)
        if text:
            points = np.frombuffer(text.encode("utf-32-le"), dtype=np.uint32)
            counts = np.bincount(points)  # at most 0x110000 bins
            uniq = np.flatnonzero(counts)
            order = np.argsort(-counts[uniq], kind="stable")
            self._points = uniq[order].astype(np.uint32)
        else:
            self._points = np.zeros(0, dtype=np.uint32)
        if len(self._points) > 255:
            raise NotImplementedError("more than 255 distinct code units in one grid")
        # code point -> code (255 = not in the alphabet; codes are < 255)
        self._lut = np.full(int(self._points.max(initial=0)) + 2, 255, dtype=np.uint8)
        self._lut[self._points] = np.arange(len(self._points), dtype=np.uint8)

    @property
    def size(self) -> int:
        return max(1, len(self._points))

    def encode(self, strings: Sequence[str], stride: int):
        """codes uint8 [n][stride] (unused slots 0) and lengths int32 [n] of ``strings``."""
        n = len(strings)
        lengths = np.fromiter((len(s) for s in strings), dtype=np.int32, count=n)
        if n and int(lengths.max()) > stride:
            k = int(np.argmax(lengths))
            raise NotImplementedError(f"string {k} has {int(lengths[k])} code units > row stride {stride}")
        codes = np.zeros((n, stride), dtype=np.uint8)
        total = int(lengths.sum())
        if total:
            points = np.frombuffer("".join(strings).encode("utf-32-le"), dtype=np.uint32)
            flat = self._lut[np.minimum(points, len(self._lut) - 1)]
            if len(self._points) == 0 or (flat == 255).any():
                raise ValueError("string contains a code unit the alphabet was not built with")
            # row-major order of the live slots == concatenation order of the strings
            codes[np.arange(stride, dtype=np.int32)[None, :] < lengths[:, None]] = flat
        return codes, lengths


@dataclass
class StrTable:
    codes: torch.Tensor
    len: torch.Tensor
    orig: torch.Tensor
    n: int
    stride: int
    alphabet: int
    has_empty: bool
    hist: Optional[torch.Tensor] = None
    len_start: Optional[torch.Tensor] = None
    # 16-bucket histogram (bucket b + bucket b + 16 of ``hist``): the first stage of the RAW grid's filter on 64-unit
    # strings (include/nsm_hip.h, ABI 5); only sorted 64-unit tables carry it
    hist16: Optional[torch.Tensor] = None

    @classmethod
    def from_codes(
        cls, codes: np.ndarray, lengths: np.ndarray, alphabet: int, device, orig: Optional[np.ndarray] = None,
        sort: bool = True,
    ) -> "StrTable":
        """From uint8 codes [n][stride], stride in {64, 128, 256, 512} (entries at positions >= len are
        ignored) and lengths."""
        codes = np.asarray(codes, dtype=np.uint8)
        lengths = np.asarray(lengths, dtype=np.int32)
        n, stride = codes.shape
        if stride not in STRIDES:
            raise NotImplementedError(f"string table stride {stride} not in {STRIDES}")
        if n and (lengths.max() > stride or lengths.min() < 0):
            raise NotImplementedError(f"string longer than its row ({stride} code units)")
        if not 1 <= alphabet <= 255:
            raise ValueError("alphabet must be in [1, 255]")
        if _on_gpu(device):
            return cls._from_codes_device(codes, lengths, alphabet, device, orig, sort)
        codes = codes.copy()
        codes[np.arange(stride, dtype=np.int32)[None, :] >= lengths[:, None]] = alphabet
        if n and int(codes.max()) > alphabet:
            raise ValueError("code unit outside the alphabet")
        perm = np.argsort(-lengths, kind="stable") if sort else np.arange(n)
        base = np.arange(n, dtype=np.int32) if orig is None else np.asarray(orig, dtype=np.int32)
        # 32-bucket symbol histogram per row (bucket = code & 31) for the exact LCS upper bound;
        # counts saturate at 255, which only weakens the bound
        hist = np.zeros((n, 32), dtype=np.int64)
        if n:
            live = np.arange(stride, dtype=np.int32)[None, :] < lengths[:, None]
            slot = (np.arange(n, dtype=np.int64)[:, None] * 32 + (codes & 31))[live]
            hist = np.bincount(slot, minlength=n * 32).reshape(n, 32)
        hist16 = np.minimum(hist[:, :16] + hist[:, 16:], 255).astype(np.uint8) if (sort and stride == 64) else None
        hist = np.minimum(hist, 255).astype(np.uint8)
        # rows of length (stride - c) occupy [len_start[c], len_start[c + 1]) in the length-sorted table
        len_start = np.zeros(stride + 2, dtype=np.int32)
        len_start[1:] = np.cumsum(np.bincount(lengths, minlength=stride + 1)[: stride + 1][::-1])
        return cls(
            hist=_dev(hist[perm], device),
            hist16=None if hist16 is None else _dev(hist16[perm], device),
            len_start=_dev(len_start, device) if sort else None,
            codes=_dev(codes[perm], device),
            len=_dev(lengths[perm], device),
            orig=_dev(base[perm], device),
            n=n,
            stride=stride,
            alphabet=alphabet,
            has_empty=bool(n and lengths.min() == 0),
        )

    @classmethod
    def _from_codes_device(cls, codes, lengths, alphabet, device, orig, sort):
        """The same table as the numpy path of ``from_codes``, built on the GPU by ``nsm_build_str_table``."""
        lib = _lib.load()
        n, stride = codes.shape
        dev = torch.device(device)
        new = lambda shape, dtype: torch.empty(shape, dtype=dtype, device=dev)
        # one spare row: an item without levels that closes the items table points one row past the strings, and the
        # one-word levels kernel stages the heads of every row of a batch before it knows the row's depth
        cap = n + 1
        t = cls(codes=new((cap, stride), torch.uint8), len=new(cap, torch.int32),
                orig=new(cap, torch.int32), n=n, stride=stride, alphabet=alphabet,
                has_empty=bool(n and int(lengths.min()) == 0), hist=new((cap, 32), torch.uint8),
                hist16=new((cap, 16), torch.uint8) if (sort and stride == 64) else None,
                len_start=new(stride + 2, torch.int32) if sort else None)
        d_codes, d_len = _dev(codes, dev), _dev(lengths, dev)
        d_orig = None if orig is None else _dev(np.asarray(orig, dtype=np.int32), dev)
        st = t.struct()
        try:
            _lib.check(lib.nsm_build_str_table(_ptr(d_codes), _ptr(d_len), _ptr(d_orig), n, _lib.BUILD_SORT if sort else 0, st,
                                               _stream(dev)), "nsm_build_str_table")
        except _lib.NsmLibraryError as exc:
            if "code unit outside the alphabet" in str(exc):
                raise ValueError("code unit outside the alphabet") from exc
            raise
        # the spare row is defined memory: an empty string (length 0, pad codes, empty histogram)
        t.codes[n:].fill_(alphabet)
        t.len[n:].zero_()
        t.orig[n:].zero_()
        t.hist[n:].zero_()
        t.codes, t.len, t.orig, t.hist = t.codes[:n], t.len[:n], t.orig[:n], t.hist[:n]
        if t.hist16 is not None:
            t.hist16[n:].zero_()
            t.hist16 = t.hist16[:n]
        return t

    @classmethod
    def from_strings(cls, strings: Sequence[str], alphabet: Alphabet, device, sort: bool = True,
                     stride: Optional[int] = None) -> "StrTable":
        """From already pre-processed Python strings (see ``score_functions.default_process``).
        ``alphabet`` must have been built from BOTH sides' strings (``encode_strings`` does that and
        picks one stride for both sides)."""
        stride = stride or pick_stride(max((len(s) for s in strings), default=0))
        codes, lengths = alphabet.encode(strings, stride)
        return cls.from_codes(codes, lengths, alphabet.size, device, sort=sort)

    def struct(self) -> _lib.NsmStrTable:
        return _lib.NsmStrTable(
            self.codes.data_ptr(), self.len.data_ptr(), self.orig.data_ptr(),
            None if self.len_start is None else self.len_start.data_ptr(),
            None if self.hist is None else self.hist.data_ptr(), self.n, self.stride, self.alphabet,
            None if self.hist16 is None else self.hist16.data_ptr(),
        )

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.codes, self.len, self.orig))


def encode_strings(left: Sequence[str], right: Sequence[str], device):
    """Both sides of a RAW fuzzy grid over one shared alphabet."""
    alpha = Alphabet(list(left) + list(right))
    stride = pick_stride(max((len(s) for s in list(left) + list(right)), default=0))
    lt = StrTable.from_strings(left, alpha, device, stride=stride)
    rt = StrTable.from_strings(right, alpha, device, stride=stride)
    # both tables must agree on the alphabet size (it is also the pad code)
    assert lt.alphabet == rt.alphabet == alpha.size
    return lt, rt


@dataclass
class LevelItems:
    """Items whose levels are consecutive rows of a (not length-sorted) ``StrTable``."""

    first: torch.Tensor
    nlev: torch.Tensor
    orig: torch.Tensor
    cat: Optional[torch.Tensor]
    n: int
    seg: Optional[torch.Tensor] = None
    seg_start: Optional[torch.Tensor] = None
    category_mode: Optional[int] = None  # the mode the kernel must be called with (set by the encoder)

    def struct(self) -> _lib.NsmLevelItems:
        def ptr(t):
            return None if t is None else t.data_ptr()

        return _lib.NsmLevelItems(
            self.first.data_ptr(), self.nlev.data_ptr(), self.orig.data_ptr(), ptr(self.cat), ptr(self.seg),
            ptr(self.seg_start), self.n,
        )


def _level_items_device(first, nlev, cat, orig, mode, do_partition, device) -> "LevelItems":
    """``LevelItems`` built on the GPU by ``nsm_build_level_items`` (``cat`` already carries bit 63 for the
    empty items when the partition turned "both empty" into a category)."""
    lib = _lib.load()
    n = len(first)
    dev = torch.device(device)
    rows = int(np.bitwise_count(cat).sum()) if do_partition else n
    new = lambda shape, dtype: torch.empty(shape, dtype=dtype, device=dev)
    cap = max(rows, 1)  # (an empty tensor has no data pointer)
    li = LevelItems(first=new(cap, torch.int32), nlev=new(cap, torch.int32),
                    orig=new(cap, torch.int32), cat=None if cat is None else new(cap, torch.int64),
                    n=rows, seg=new(cap, torch.int32) if do_partition else None,
                    seg_start=new(65, torch.int32) if do_partition else None, category_mode=mode)
    d_first, d_nlev = _dev(first, dev), _dev(nlev, dev)
    d_cat = None if cat is None else _dev(np.asarray(cat, dtype=np.uint64).view(np.int64), dev)
    d_orig = _dev(np.asarray(orig, dtype=np.int32), dev)  # the caller's item ids (the rows arrive pre-ordered)
    st = li.struct()
    # the mode the builder sees is the one AFTER the rewrite: the caller already folded "both empty" into bit 63
    _lib.check(lib.nsm_build_level_items(_ptr(d_first), _ptr(d_nlev), _ptr(d_cat), _ptr(d_orig), n,
                                         _lib.CAT_INTERSECT if cat is not None else _lib.CAT_NONE,
                                         _lib.BUILD_PARTITION if do_partition else 0, st, _stream(dev)), "nsm_build_level_items")
    if st.n != rows:
        raise _lib.NsmLibraryError(f"nsm_build_level_items built {st.n} rows, expected {rows}")
    for col in ("first", "nlev", "orig", "cat", "seg"):
        if getattr(li, col) is not None:
            setattr(li, col, getattr(li, col)[:rows])
    return li


def encode_level_strings(
    left_items: Sequence[Sequence[str]], right_items: Sequence[Sequence[str]], device,
    left_cat: Optional[np.ndarray] = None, right_cat: Optional[np.ndarray] = None,
    category_mode: int = _lib.CAT_NONE, partition: bool = True, left_offset: int = 0,
):
    """Levels-mode fuzzy operands: every level of every item is one (pre-processed) string.
    ``left_offset`` is added to the left item ids reported in hits (a rank's block of a sharded left side).

    With a category predicate (and ``partition``) both sides are PARTITIONED by category: an item
    with k categories becomes k rows, rows are grouped per category, and the kernel only visits
    same-category pairs (each matching pair once, in its lowest common category).  That turns the
    predicate from a per-lane mask test -- which saves nothing, a wavefront of 64 unrelated items
    almost always contains one that matches -- into not visiting the other ~(1 - k/C) of the grid.
    """
    alpha = Alphabet(s for items in (left_items, right_items) for levels in items for s in levels)
    stride = pick_stride(max((len(s) for items in (left_items, right_items) for lv in items for s in lv), default=0))

    def flatten(items):
        flat: List[str] = []
        first = np.zeros(len(items), dtype=np.int32)
        nlev = np.zeros(len(items), dtype=np.int32)
        for k, levels in enumerate(items):
            if len(levels) > MAX_LEVELS:
                raise NotImplementedError(f"item {k} has {len(levels)} levels > {MAX_LEVELS}")
            first[k] = len(flat)
            nlev[k] = len(levels)
            flat.extend(levels)
        codes, lengths = alpha.encode(flat, stride)
        return codes, lengths, first, nlev

    return encode_level_codes(flatten(left_items), flatten(right_items), alpha.size, device, left_cat, right_cat,
                              category_mode, partition, left_offset)


def encode_level_codes(
    left, right, alphabet: int, device, left_cat: Optional[np.ndarray] = None, right_cat: Optional[np.ndarray] = None,
    category_mode: int = _lib.CAT_NONE, partition: bool = True, left_offset: int = 0,
):
    """``encode_level_strings`` for operands that already are dense code units: ``left`` / ``right`` =
    ``(codes uint8 [rows][stride], lengths int32 [rows], first int32 [n], nlev int32 [n])`` over ONE alphabet of
    ``alphabet`` symbols and one stride; level l of item k is row ``first[k] + l``."""
    if left[0].shape[1] != right[0].shape[1]:
        raise ValueError("both sides must use the same row stride")
    use_cat = category_mode != _lib.CAT_NONE and left_cat is not None and right_cat is not None
    mode = category_mode if use_cat else _lib.CAT_NONE
    cats = {}
    if use_cat:
        for name, cat in (("l", left_cat), ("r", right_cat)):
            cats[name] = np.asarray(cat, dtype=np.uint64).copy()
        if partition:
            if mode == _lib.CAT_INTERSECT_OR_BOTH_EMPTY:
                if any(int(c.max(initial=0)) >> EMPTY_CATEGORY_BIT for c in cats.values()):
                    partition = False  # all 64 bits are real categories: keep the per-lane predicate
                else:  # "both empty" becomes one more category shared by the empty items
                    for c in cats.values():
                        c[c == 0] = np.uint64(1) << np.uint64(EMPTY_CATEGORY_BIT)
                    mode = _lib.CAT_INTERSECT
    do_partition = use_cat and partition

    def side(operand, cat, offset=0):
        codes, lengths, first, nlev = operand
        first, nlev = np.asarray(first, dtype=np.int32), np.asarray(nlev, dtype=np.int32)
        if len(nlev) and int(nlev.max()) > MAX_LEVELS:
            raise NotImplementedError(f"an item has {int(nlev.max())} levels > {MAX_LEVELS}")
        table = StrTable.from_codes(codes, lengths, alphabet, device, sort=False)
        # Items are pre-ordered by (depth, length of the step-1 level string), both descending: the builder's sort
        # (category, depth) is stable, so inside a depth class a wavefront's 64 items have step-1 strings of similar
        # length -- the LCS loops run to the wave's LONGEST text (Term-like items: mean 59, wave maximum ~110
        # code units when unordered).
        lengths = np.asarray(lengths, dtype=np.int32)

        def step_len(t):  # length of the level string step t compares: level min(t, depth - 1)
            if not len(lengths):
                return np.zeros(len(first), dtype=np.int32)
            row = np.minimum(first + np.minimum(t, np.maximum(nlev, 1) - 1), len(lengths) - 1)
            return np.where(nlev > 0, lengths[row], 0)

        len1 = step_len(1)
        if codes.shape[1] > 64:
            # Multi-word strings (shared-tile kernel): a block's 64 items stay together for EVERY step, and steps 2 and 3
            # cost more per code unit than step 1 (more words), so their lengths come first, in buckets of 16 code
            # units, and step 1 orders the items inside a bucket.  Term-like items (20k, mean 59 / 87 / 104 code units
            # at steps 1 / 2 / 3), mean longest text of a tile: (depth, len1) 60 / 113 / 138, this key 65 / 96 / 113.
            pre = np.lexsort((-len1, -(step_len(2) // 16), -(step_len(3) // 16), -nlev)).astype(np.int32)
        else:
            pre = np.lexsort((-len1, -nlev)).astype(np.int32)
        if _on_gpu(device):
            return _level_items_device(first[pre], nlev[pre], None if cat is None else cat[pre], pre + offset, mode,
                                       do_partition, device), table
        first_p, nlev_p, cat_p = first[pre], nlev[pre], (None if cat is None else cat[pre])
        item = np.arange(len(first), dtype=np.int32)
        seg = seg_start = None
        if do_partition:
            rows, segs = [], []
            for c in range(64):
                has = np.flatnonzero((cat_p >> np.uint64(c)) & np.uint64(1))
                rows.append(has.astype(np.int32))
                segs.append(np.full(len(has), c, dtype=np.int32))
            item = np.concatenate(rows) if rows else item[:0]
            seg = np.concatenate(segs) if segs else np.zeros(0, np.int32)
            # inside a category: deeper items first, so a wavefront's items have similar depth (stable: the
            # pre-order survives inside a depth class)
            order = np.lexsort((-nlev_p[item], seg))
            item, seg = item[order], seg[order]
            seg_start = np.zeros(65, dtype=np.int32)
            seg_start[1:] = np.cumsum(np.bincount(seg, minlength=64)[:64])
        else:
            item = item[np.argsort(-nlev_p, kind="stable")]
        li = LevelItems(
            first=_dev(first_p[item], device), nlev=_dev(nlev_p[item], device),
            orig=_dev((pre[item] + offset).astype(np.int32), device),
            cat=None if cat_p is None else _dev(cat_p[item], device), n=len(item),
            seg=None if seg is None else _dev(seg, device),
            seg_start=None if seg_start is None else _dev(seg_start, device), category_mode=mode,
        )
        return li, table

    l_items, l_table = side(left, cats.get("l"), left_offset)
    r_items, r_table = side(right, cats.get("r"))
    return l_items, l_table, r_items, r_table
