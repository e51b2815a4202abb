"""``ComparableData``: the cross-cohort N x M match loop with the reference's surface.

Mirrors ``napkon_string_matching/types/comparable_data.py`` (:69-299, :452-574): same method names,
keyword arguments, output columns, thresholds (``>=`` on doubles), exceptions and pair labels
(``i*M + j`` after dropna / whitelist removal).  What differs is WHERE the work runs:

* per ITEM (host, O(N + M)): dropna, whitelist removal, ``gen_comp_value`` levels, token -> id /
  string encoding, category bit masks;
* per PAIR (MI355X, one launch): the ``compare_terms`` sum over levels with the chosen score
  function, the category predicate and the threshold (csrc/jaccard_levels_impl.hpp,
  csrc/indel_levels.hip).  The reference's N*M-row ``merge(how="cross")`` frame (:191) is never built;
* per HIT (host): the blacklist -- dropping a blacklisted pair before scoring (:195-204) equals
  dropping it from the hit list -- and the output frame.

The reference's per-pair exceptions only depend on per-item properties (a zero-level item ->
``IndexError`` at :262, two empty level sets -> ``ZeroDivisionError`` at score_functions.py:13), so
they are resolved on the host, in pair order, against the same blacklist / category filters.
"""
from __future__ import annotations

import json
import logging
from contextlib import contextmanager
from hashlib import md5
from pathlib import Path
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from .. import _lib, distributed, grid, tables
from ..compare import score_functions
from .comparable import COLUMN_NAMES, IDENTIFIER, MATCH_SCORE, QUESTION_OUTPUT, Comparable
from .mapping import Mapping

logger = logging.getLogger(__name__)

PREPARE_REMOVE_SYMBOLS = "!?,.()[]:;*"  # comparable_data.py:24
CACHE_FILE_PATTERN = "compared__score_{}.json"  # comparable_data.py:25
COMP_COLUMN = "Compare"
TERM = "Term"
_INF = 1 << 30


# =============================================================================== per item
def flatten_list(list_) -> List[str]:
    """One level of flattening (:567-574); a ``str`` is walked character by character."""
    out: List[str] = []
    for part in list_:
        if isinstance(part, list):
            out += part
        else:
            out.append(part)
    return out


class Tokenizer:
    """The word tokenizer + stop-word list of ``tokenize`` (:287-299).

    Upstream uses nltk punkt and nltk's German stop words; neither ships with this package (no
    network at build time), so both are injectable.  The default (``str.split``, no stop words)
    equals punkt on text made of plain alphanumeric words that are not stop words.
    """

    def __init__(self, word_tokenize: Callable[[str], Iterable[str]] = str.split, stop_words: Iterable[str] = ()):
        self.word_tokenize = word_tokenize
        self.stop_words = frozenset(stop_words)

    @classmethod
    def from_nltk(cls, language: str = "german") -> "Tokenizer":
        from nltk.corpus import stopwords  # optional dependency
        from nltk.tokenize import word_tokenize

        return cls(word_tokenize, stopwords.words(language))

    def __call__(self, parts) -> List[str]:
        words = self.word_tokenize(" ".join(flatten_list(parts)))
        kept = {w for w in words if w.casefold() not in self.stop_words and w not in PREPARE_REMOVE_SYMBOLS}
        return sorted(kept, key=str.casefold)

    @property
    def compositional(self) -> bool:
        """True when tokenising a blank-joined sequence equals concatenating the parts' tokens
        (whitespace splitting): the suffix levels can then be built incrementally."""
        return self.word_tokenize is str.split

    def levels(self, items) -> List[List[str]]:
        """``[tokenize(items[-k:]) for k in 1..len(items)]`` (gen_comp_value, :283-285).  For a
        compositional tokenizer each entry is tokenised once and the suffix sets grow by union --
        the same lists, without the O(L^2) re-tokenisation."""
        if not self.compositional or isinstance(items, str):
            return [self(items[-k:]) for k in range(1, len(items) + 1)]
        out: List[List[str]] = []
        acc: set = set()
        stop = self.stop_words
        for entry in reversed(items):
            parts = entry if isinstance(entry, list) else [entry]
            for part in parts:
                for w in part.split():
                    if w not in PREPARE_REMOVE_SYMBOLS and (not stop or w.casefold() not in stop):
                        acc.add(w)
            out.append(sorted(acc, key=str.casefold))
        return out


# =============================================================================== mappings
def get_identifiers_from_mapping(mappings: Mapping, group: str) -> List[str]:
    out: List[str] = []
    for entry in mappings.values():
        out += entry[group]
    return out


def flatten_mapping(left_group: str, right_group: str, mapping: Mapping) -> List[Tuple[str, str]]:
    """Cartesian identifier pairs of every entry that lists both cohorts (:555-564)."""
    return [(a, b) for lefts, rights in mapping.get_all_mapping_for_groups(left_group, right_group)
            for a in lefts for b in rights]


def _as_mapping(value) -> Mapping:
    if value is None:
        return Mapping()
    if isinstance(value, Mapping):
        return value
    if isinstance(value, dict):
        return Mapping(value)
    if hasattr(value, "dict"):
        return Mapping(value.dict())
    raise TypeError("mappings must be a Mapping or a {uuid: {cohort: [identifiers]}} dict")


# =============================================================================== categories
class _Categories:
    """Category columns of both sides as <= 64-bit masks + the device predicate mode."""

    def __init__(self, left: Sequence, right: Sequence, first_left, first_right) -> None:
        l_list, r_list = isinstance(first_left, list), isinstance(first_right, list)
        if l_list and not r_list:
            # :471 evaluates `x in set(y)` with x a list: hashing a list raises
            raise TypeError("unhashable type: 'list'")
        self.mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY if (l_list and r_list) else _lib.CAT_INTERSECT
        self.scalar_scalar = not l_list and not r_list
        # inside item_memo() the label -> bit assignment is shared by the run's grids, which makes the
        # per-cell result reusable (a cohort's cells are looked at once per grid it takes part in)
        memo = ComparableData._item_memo
        bits: Dict[object, int] = {} if memo is None else memo.setdefault(("category bits",), {})

        def mask_of(value, as_list: bool) -> Tuple[int, frozenset]:
            if memo is None:
                return mask(value, as_list)
            key = ("category", as_list, id(value))
            got = memo.get(key)
            if got is None or got[0] is not value:
                got = (value, mask(value, as_list))
                memo[key] = got
            return got[1]

        def mask(value, as_list: bool) -> Tuple[int, frozenset]:
            labels = value if as_list else [value]
            if as_list and not isinstance(value, list):
                raise NotImplementedError("Category column mixes lists and scalars")
            if not as_list and isinstance(value, list):
                raise NotImplementedError("Category column mixes lists and scalars")
            m = 0
            keys = []
            for lab in labels:
                if isinstance(lab, float) and lab != lab:
                    continue  # NaN never equals anything
                keys.append(lab)
                m |= 1 << bits.setdefault(lab, len(bits))
            return m, frozenset(keys)

        lm = [mask_of(v, l_list) for v in left]
        rm = [mask_of(v, r_list) for v in right]
        self.left_sets = [s for _, s in lm]
        self.right_sets = [s for _, s in rm]
        self.on_device = len(bits) <= 64
        if self.on_device:
            self.left_mask = np.array([m for m, _ in lm], dtype=np.uint64)
            self.right_mask = np.array([m for m, _ in rm], dtype=np.uint64)

    def match(self, i: int, j: int) -> bool:
        a, b = self.left_sets[i], self.right_sets[j]
        if a & b:
            return True
        return self.mode == _lib.CAT_INTERSECT_OR_BOTH_EMPTY and not a and not b


# =============================================================================== the class
def _agree_on_rank0(flag: bool) -> bool:
    """Rank 0's verdict on every rank (one tiny broadcast; a no-op without a process group)."""
    rank, size = distributed.world()
    if size == 1:
        return bool(flag)
    return bool(distributed.broadcast_object(bool(flag) if rank == 0 else None))


def _read_cache_collectively(cache_file: Path) -> Comparable:
    """The cached ``Comparable``: read by rank 0 and handed to the other ranks (their ``cache_dir`` need not
    be the same file system)."""
    rank, size = distributed.world()
    if size == 1:
        return Comparable.read_json(cache_file)
    # rank 0 may fail (unreadable or corrupt file): it must still take part in the broadcast, or the other ranks wait
    # for it forever -- the failure travels as the payload and every rank raises it (round-3 advice)
    box = None
    if rank == 0:
        try:
            text = cache_file.read_text(encoding="utf-8")
            json.loads(text)
            box = (True, text)
        except Exception as exc:  # noqa: BLE001 -- re-raised on every rank below
            box = (False, f"{type(exc).__name__}: {exc}")
    ok, payload = distributed.broadcast_object(box)
    if not ok:
        raise RuntimeError(f"compare cache {cache_file} could not be read on rank 0: {payload}")
    return Comparable(data=json.loads(payload))


class ComparableData:
    """A cohort's items as a pandas frame plus the ``compare`` machinery.

    Needs the columns ``Identifier``, ``Term`` and the compare column; ``Variable``, ``Sheet`` are
    copied to the result when present; ``Category`` when ``filter_categories`` is on.
    """

    __column_mapping__: Dict[str, str] = {}
    __category_column__ = "Category"
    tokenizer: Tokenizer = Tokenizer()
    _item_memo: Optional[Dict[int, tuple]] = None  # see item_memo()

    @classmethod
    @contextmanager
    def item_memo(cls):
        """Per-ITEM results (levels, fuzzy operands) are remembered by the identity of the cell object
        while the context is open: with k cohorts every item takes part in k - 1 grids
        (Matcher.match_questionnaires), and dropna / rename hand the same cell objects on.  The cells
        must not be modified in place inside the context."""
        outer = ComparableData._item_memo
        ComparableData._item_memo = {} if outer is None else outer
        try:
            yield
        finally:
            ComparableData._item_memo = outer

    @staticmethod
    def _memoised(tag, values, func) -> list:
        memo = ComparableData._item_memo
        if memo is None:
            return [func(v) for v in values]
        out = []
        for v in values:
            key = (tag, id(v))
            got = memo.get(key)
            if got is None or got[0] is not v:  # the memo keeps `v` alive, so its id cannot be reused
                got = (v, func(v))
                memo[key] = got
            out.append(got[1])
        return out

    def __init__(self, data=None) -> None:
        if isinstance(data, ComparableData):
            data = data._data
        self.__dict__["_data"] = data if isinstance(data, pd.DataFrame) else pd.DataFrame(data)

    # ---- thin frame wrapper (reference: types/data.py)
    def __getattr__(self, name: str):
        return getattr(self._data, name)

    def __getitem__(self, item):
        got = self._data[item]
        return self.__class__(got) if isinstance(got, pd.DataFrame) else got

    def __setitem__(self, item, value) -> None:
        self._data[item] = value

    def __len__(self) -> int:
        return len(self._data)

    def __repr__(self) -> str:
        return repr(self._data)

    def dataframe(self) -> pd.DataFrame:
        return self._data

    def dropna(self, *args, **kwargs) -> "ComparableData":
        return self.__class__(self._data.dropna(*args, **kwargs))

    def to_csv(self) -> str:
        return self._data.to_csv(index=False)

    def map_for_comparable(self) -> pd.DataFrame:
        return self._data.rename(columns=self.__column_mapping__)

    # ---- per item
    @classmethod
    def tokenize(cls, parts, language: str = "german") -> List[str]:
        return cls.tokenizer(parts)

    @classmethod
    def gen_comp_value(cls, items) -> List[List[str]]:
        """Level l = tokens of the last l+1 entries (:283-285)."""
        return cls.tokenizer.levels(items)

    @staticmethod
    def gen_term(*items: str) -> List[str]:
        return [item for item in items if item]

    def get_existing_mapping_ids(self, group_name: str, mappings: Mapping) -> List[str]:
        per_group = mappings.filter_by_group(group_name)
        identifiers = list(self._data[IDENTIFIER])
        return list({key for key, members in per_group.items() for ident in identifiers if ident in members})

    def remove_existing_mappings(self, existing_mappings) -> None:
        drop = set(existing_mappings)
        self.__dict__["_data"] = self._data[[v not in drop for v in self._data[IDENTIFIER]]]

    # ---- per pair (single): compare_terms as a 1 x 1 levels grid on the GPU
    @classmethod
    def compare_terms(cls, left: Sequence, right: Sequence, score_func) -> float:
        """:248-265 for ONE pair.  ``score_func`` is one of this package's plugins (or its name)."""
        plugin = _resolve_plugin(score_func)
        if len(left) == 0 and len(right) == 0:
            return 0
        if len(left) == 0 or len(right) == 0:
            raise IndexError("list index out of range")
        if plugin.kind == "sets":
            sets = lambda levels: [list(score_functions.set_operand(lv)) for lv in levels]
            sl, sr = sets(left), sets(right)
            if _divides_by_zero(sl, _empty_levels(sl), sr, _empty_levels(sr)):
                raise ZeroDivisionError("division by zero")  # an empty-vs-empty level is visited
        hits = _levels_grid(plugin, [list(left)], [list(right)], float("-inf"), None, None)
        return float(hits.score[0])

    # ---- the grid
    def compare(
        self,
        other,
        existing_mappings_whitelist=None,
        existing_mappings_blacklist=None,
        compare_column: str = None,
        score_threshold: float = 0.1,
        cached: bool = True,
        cache_threshold: Optional[float] = None,
        cache_dir=None,
        identifier_column_left: Optional[str] = None,
        identifier_column_right: Optional[str] = None,
        *args,
        **kwargs,
    ) -> Comparable:
        """:69-128.  Scores at ``cache_threshold or score_threshold``, keeps ``>= score_threshold``,
        orders by score descending.

        Compare cache: the reference keys ``compared__score_{md5}.json`` by a hash
        that embeds object addresses (``str(kwargs.items())`` of ``Mapping`` objects, :61-67), so it
        never hits, and it leaves ``score_func`` out of the key.  Here the key is a stable content
        hash (both frames, both mappings, compare column, cache threshold, score function, category
        filter, cohort names) and the cache is only used when ``cache_dir`` is given.  The file format
        is the reference's: ``{"left_name", "right_name", "data": [records]}`` (indent 4); like
        upstream, a frame read back from the cache carries a fresh RangeIndex instead of pair labels.
        """
        first = cache_threshold if cache_threshold else score_threshold
        cache_file = None
        if cached and cache_dir is not None:
            cache_file = Path(cache_dir) / CACHE_FILE_PATTERN.format(
                self._hash_compare_args(other, existing_mappings_whitelist, existing_mappings_blacklist,
                                        compare_column, first, kwargs))
        # hit or miss is decided ONCE for all ranks of a sharded run (rank 0 looks, everybody follows): with a
        # rank-local exists() the ranks can disagree -- a cache_dir that is not shared between nodes, or a repeated
        # compare() where one rank looks before rank 0's rename has landed -- and the ranks that miss would then
        # wait in gen_comparable's collectives for a rank that never comes
        use_cache = cache_file is not None and _agree_on_rank0(cache_file.exists())
        if use_cache:
            logger.info("using cached result")
            result = _read_cache_collectively(cache_file)
        else:
            if cache_file is None:
                # the rows between `first` and `score_threshold` only ever fed the cache file: without
                # it, scoring at `first` and filtering at `score_threshold` equals scoring at the max
                first = max(first, score_threshold)
            result = self.gen_comparable(
                other,
                existing_mappings_whitelist,
                existing_mappings_blacklist,
                *args,
                score_threshold=first,
                compare_column=compare_column,
                identifier_column_left=identifier_column_left,
                identifier_column_right=identifier_column_right,
                **kwargs,
            )
            if cache_file is not None:
                # every rank of a sharded run holds the full result: rank 0 writes it (atomically, see write_json); its
                # verdict is broadcast, which also is the barrier: nobody returns (and calls compare again) before the
                # file is in place, and a failed write (disk full) raises on every rank instead of leaving them waiting
                error = None
                if distributed.world()[0] == 0:
                    try:
                        cache_file.parent.mkdir(parents=True, exist_ok=True)
                        logger.info("write cache to file")
                        result.write_json(cache_file)
                    except Exception as exc:  # noqa: BLE001 -- re-raised on every rank below
                        error = f"{type(exc).__name__}: {exc}"
                if distributed.world()[1] > 1:
                    error = distributed.broadcast_object(error)
                if error is not None:
                    raise RuntimeError(f"compare cache {cache_file} could not be written on rank 0: {error}")
        result = result[result.match_score >= score_threshold]
        logger.info("got %i filtered entries", len(result))
        result.sort_by_score()
        return result

    def _hash_compare_args(self, other, whitelist, blacklist, compare_column, cache_threshold, kwargs) -> str:
        other_csv = other.to_csv() if hasattr(other, "to_csv") else pd.DataFrame(other).to_csv(index=False)
        parts = [
            self.to_csv(), other_csv,
            json.dumps(_as_mapping(whitelist).dict(), sort_keys=True),
            json.dumps(_as_mapping(blacklist).dict(), sort_keys=True),
            str(compare_column), repr(cache_threshold),
            json.dumps({k: kwargs.get(k) for k in ("score_func", "filter_categories", "category_column", "left_name",
                                                    "right_name")}, sort_keys=True, default=str),
        ]
        return md5("\x1f".join(parts).encode("utf-8"), usedforsecurity=False).hexdigest()

    def gen_comparable(
        self,
        right,
        existing_mappings_whitelist=None,
        existing_mappings_blacklist=None,
        score_func: str = None,
        compare_column: str = None,
        category_column: str = "Category",
        score_threshold: float = 0.1,
        left_name: str = None,
        right_name: str = None,
        filter_categories: bool = False,
        identifier_column_left: Optional[str] = None,
        identifier_column_right: Optional[str] = None,
        *args,
        **kwargs,
    ) -> Comparable:
        """:133-246 (steps 1-12 of SURVEY.md 3.2), the per-pair part on the GPU."""
        plugin = getattr(score_functions, score_func)  # AttributeError for an unknown name (:150)
        whitelist = _as_mapping(existing_mappings_whitelist)
        blacklist = _as_mapping(existing_mappings_blacklist)
        if not isinstance(right, ComparableData):
            right = ComparableData(right)

        left = self.dropna(subset=[compare_column])
        right = right.dropna(subset=[compare_column])
        logger.info(
            "comparing number of items %i left, %i right, potential %s comparisons",
            len(left), len(right), "{:,}".format(len(left) * len(right)),
        )
        remove_existing_mappings(left, right, left_name, right_name, whitelist)
        logger.info("after removing existing whitelisted mappings: %i left, %i right", len(left), len(right))

        lf, rf = left.map_for_comparable(), right.map_for_comparable()
        lp, rp = left_name.title(), right_name.title()
        n_l, n_r = len(lf), len(rf)

        levels_l = self._memoised(("levels", id(self.tokenizer)), lf[compare_column], self.gen_comp_value)
        levels_r = self._memoised(("levels", id(self.tokenizer)), rf[compare_column], self.gen_comp_value)
        argument_l = [":".join(flatten_list(item)) for item in lf[TERM]]
        argument_r = [":".join(flatten_list(item)) for item in rf[TERM]]

        # ---- blacklist as position pairs
        id_l = list(lf[identifier_column_left or IDENTIFIER])
        id_r = list(rf[identifier_column_right or IDENTIFIER])
        banned = _banned_positions(flatten_mapping(left_name, right_name, blacklist), id_l, id_r)
        logger.info("remaining %s entries after removing blacklisted ones", "{:,}".format(n_l * n_r - len(banned)))

        # ---- categories
        cats: Optional[_Categories] = None
        if filter_categories:
            first = _first_surviving_pair(n_l, n_r, banned)
            if first is None:
                raise IndexError("single positional indexer is out-of-bounds")  # df.iloc[0] at :465
            cl, cr = list(lf[category_column]), list(rf[category_column])
            cats = _Categories(cl, cr, cl[first[0]], cr[first[1]])
        elif n_l == 0 or n_r == 0:
            # the reference's blacklist step indexes the cross join with a list of booleans (:549-552); for
            # an EMPTY cross join that list is empty, pandas reads it as "no columns", and :223-232 then
            # fails on the compare column
            raise KeyError(lp + COMP_COLUMN)

        def survives(i: int, j: int) -> bool:
            return (i, j) not in banned and (cats is None or cats.match(i, j))

        # ---- the reference's per-pair exceptions, in pair order
        nlev_l = np.array([len(v) for v in levels_l], dtype=np.int64)
        nlev_r = np.array([len(v) for v in levels_r], dtype=np.int64)
        extra_hits: List[Tuple[int, int]] = []  # zero-level x zero-level pairs score 0
        _raise_first_pair_error(
            plugin.kind == "sets", levels_l, levels_r, nlev_l, nlev_r, n_r, survives, extra_hits
        )

        # ---- device grid over the items that have at least one level
        keep_l = np.flatnonzero(nlev_l > 0)
        keep_r = np.flatnonzero(nlev_r > 0)
        rank, world_size = distributed.world()
        if world_size > 1:  # left rows block-sharded over the ranks, right side replicated
            row_lo, row_hi = distributed.shard_bounds(n_l, rank, world_size)
            keep_l = keep_l[(keep_l >= row_lo) & (keep_l < row_hi)]
            extra_hits = [p for p in extra_hits if row_lo <= p[0] < row_hi]
        logger.info("calculate score")
        host_filter = bool(banned) or (cats is not None and not cats.on_device)
        pending = None
        if keep_l.size and keep_r.size:
            on_device = cats is not None and cats.on_device
            hits = _levels_grid(
                plugin,
                [levels_l[k] for k in keep_l],
                [levels_r[k] for k in keep_r],
                score_threshold,
                cats.left_mask[keep_l] if on_device else None,
                cats.right_mask[keep_r] if on_device else None,
                cats.mode if on_device else _lib.CAT_NONE,
                defer=world_size > 1 and not host_filter,
            )
            if isinstance(hits, grid.PendingHits):
                pending = hits
        extra = bool(extra_hits) and 0.0 >= score_threshold
        if world_size > 1:
            # The one exchange step: all-gatherv of the hits.  When every rank's hits sit in ONE device buffer and nothing
            # has to be filtered on the host (no blacklist, the category predicate ran in the kernel, no zero-level pairs)
            # the buffers go GPU -> all-gather -> GPU at a capacity agreed with one MAX all-reduce; otherwise the ranks
            # exchange what they have filtered on the host.  The choice is collective (one MIN all-reduce).
            nothing_to_score = not (keep_l.size and keep_r.size)  # (this rank's shard: it then contributes an empty buffer)
            direct = distributed.agree_all((pending is not None or nothing_to_score) and not extra and not host_filter)
        else:
            direct = False
        if direct:
            hs, hi, hj = distributed.all_gather_pending(pending, n_l, nlev_l, nlev_r, world_size)
        else:
            if pending is not None:
                hits = pending.finish()
            if keep_l.size and keep_r.size:
                hi, hj, hs = keep_l[hits.i], keep_r[hits.j], hits.score
            else:
                hi, hj, hs = np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float64)
            if extra:
                ei = np.array([p[0] for p in extra_hits], dtype=np.int64)
                ej = np.array([p[1] for p in extra_hits], dtype=np.int64)
                hi, hj, hs = np.concatenate([hi, ei]), np.concatenate([hj, ej]), np.concatenate([hs, np.zeros(len(ei))])

            # ---- per hit: blacklist (and categories when they could not go to the device)
            if len(hs) and host_filter:
                ok = np.fromiter(
                    (
                        (int(a), int(b)) not in banned and (cats is None or cats.on_device or cats.match(int(a), int(b)))
                        for a, b in zip(hi, hj)
                    ),
                    dtype=bool, count=len(hs),
                )
                hi, hj, hs = hi[ok], hj[ok], hs[ok]

            if world_size > 1:  # (host-filtered hits: packed, copied to the device, gathered, copied back)
                hs, hi, hj = distributed.all_gather_hits(hs, hi, hj)

        # ---- output frame in the reference's row order (pair label ascending)
        label = hi.astype(np.int64) * n_r + hj.astype(np.int64)
        order = np.argsort(label, kind="stable")
        hi, hj, hs, label = hi[order], hj[order], hs[order], label[order]
        lf = lf.assign(**{COMP_COLUMN: levels_l, QUESTION_OUTPUT: argument_l})
        rf = rf.assign(**{COMP_COLUMN: levels_r, QUESTION_OUTPUT: argument_r})
        out = {}
        for frame, prefix, rows in ((lf, lp, hi), (rf, rp, hj)):
            for col in frame.columns:  # survivors keep the frame's column order (:236-240)
                if col in COLUMN_NAMES:
                    out[prefix + col] = frame[col].to_numpy()[rows] if len(rows) else frame[col].to_numpy()[:0]
        out[MATCH_SCORE] = hs
        table = pd.DataFrame(out, index=pd.Index(label))
        logger.info("got %s entries", "{:,}".format(len(table)))
        return Comparable(data=table, left_name=lp, right_name=rp)


# =============================================================================== helpers
def remove_existing_mappings(
    left: ComparableData, right: ComparableData, left_name: str, right_name: str, existing_mappings: Mapping
) -> None:
    """:493-513 -- drop items already linked by a whitelist entry; skipped as a whole on KeyError."""
    try:
        left_ids = left.get_existing_mapping_ids(left_name, existing_mappings)
        right_ids = right.get_existing_mapping_ids(right_name, existing_mappings)
    except KeyError:
        return
    used = set(left_ids) & set(right_ids)
    filtered = existing_mappings.get_filtered(used)
    left.remove_existing_mappings(get_identifiers_from_mapping(filtered, left_name))
    right.remove_existing_mappings(get_identifiers_from_mapping(filtered, right_name))


def _resolve_plugin(score_func):
    if isinstance(score_func, str):
        return getattr(score_functions, score_func)
    if score_func in (score_functions.intersection_vs_union, score_functions.fuzzy_match):
        return score_func
    raise NotImplementedError(
        "only this package's score functions (intersection_vs_union, fuzzy_match) have a device implementation"
    )


def _banned_positions(pairs, id_l: Sequence, id_r: Sequence) -> set:
    if not pairs:
        return set()
    pos_l: Dict[object, List[int]] = {}
    pos_r: Dict[object, List[int]] = {}
    for k, v in enumerate(id_l):
        pos_l.setdefault(v, []).append(k)
    for k, v in enumerate(id_r):
        pos_r.setdefault(v, []).append(k)
    out = set()
    for a, b in pairs:
        for i in pos_l.get(a, ()):
            for j in pos_r.get(b, ()):
                out.add((i, j))
    return out


def _first_surviving_pair(n_l: int, n_r: int, banned: set) -> Optional[Tuple[int, int]]:
    for i in range(n_l):
        for j in range(n_r):
            if (i, j) not in banned:
                return i, j
    return None


def _empty_levels(levels: Sequence[Sequence]) -> frozenset:
    """Indices of the item's EMPTY levels that a step can reach: step s >= 1 uses level min(s, L - 1), so level 0 only
    counts for a one-level item.  (Nested levels: the empty ones are a prefix; the reference's tokenizer may yield any.)"""
    n = len(levels)
    return frozenset(k for k, lv in enumerate(levels) if len(lv) == 0 and (k >= 1 or n == 1))


def _divides_by_zero(levels_l, empty_l: frozenset, levels_r, empty_r: frozenset) -> bool:
    """Does ``compare_terms`` reach a step at which BOTH level sets are empty (score_functions.py:13)?"""
    n_l, n_r = len(levels_l), len(levels_r)
    return any(min(s, n_l - 1) in empty_l and min(s, n_r - 1) in empty_r for s in range(1, max(n_l, n_r) + 1))


def _raise_first_pair_error(is_sets, levels_l, levels_r, nlev_l, nlev_r, n_r, survives, extra_hits) -> None:
    zero_l, zero_r = np.flatnonzero(nlev_l == 0), np.flatnonzero(nlev_r == 0)
    candidates: List[Tuple[int, int, type]] = []
    for i in zero_l:
        for j in range(len(nlev_r)):
            if nlev_r[j] == 0:
                extra_hits.append((int(i), j))
            else:
                candidates.append((int(i), j, IndexError))
    for j in zero_r:
        for i in range(len(nlev_l)):
            if nlev_l[i] != 0:
                candidates.append((i, int(j), IndexError))
    if is_sets:
        emp_l = [_empty_levels(v) for v in levels_l]
        emp_r = [_empty_levels(v) for v in levels_r]
        sus_l = [i for i, v in enumerate(emp_l) if v]
        sus_r = [j for j, v in enumerate(emp_r) if v]
        for i in sus_l:
            for j in sus_r:
                if _divides_by_zero(levels_l[i], emp_l[i], levels_r[j], emp_r[j]):
                    candidates.append((i, j, ZeroDivisionError))
    best = None
    for i, j, exc in candidates:
        if survives(i, j):
            lab = i * n_r + j
            if best is None or lab < best[0]:
                best = (lab, exc)
    extra_hits[:] = [p for p in extra_hits if survives(*p)]
    if best is not None:
        if best[1] is IndexError:
            raise IndexError("list index out of range")
        raise ZeroDivisionError("division by zero")


def _may_be_wide_sets(*sides) -> bool:
    """Cheap pre-check of the 64-token limit: the largest level's token count WITH repeats (strings: words)."""
    size = lambda lv: len(lv) if isinstance(lv, list) else len(lv.split())
    return any(size(it[-1]) > 64 for items in sides for it in items if len(it))


def _fast_jaccard_levels(levels_l, levels_r, as_set_levels, threshold, cat_l, cat_r, cat_mode, dev, defer=False):
    """The suffix-nested fast layout of both sides + ``nsm_jaccard_levels_grid``; raises ``tables.IrregularLevels`` when an
    item does not fit it."""
    part = tables.partition_allowed(cat_mode, cat_l, cat_r)
    memo = ComparableData._item_memo
    if memo is not None:
        # inside item_memo(): one vocabulary for the whole run, every item encoded once (tables.LevelPool,
        # keyed by the identity of the memoised level lists)
        pool = memo.setdefault(("level pool",), tables.LevelPool())
        vocab = pool.vocab
        rows = [pool.rows_keyed(levels, as_set_levels) for levels in (levels_l, levels_r)]
        width = tables.pick_width(*(int((ids >= 0).sum(axis=1).max(initial=1)) for ids, _, _ in rows))
        sides = []
        for (ids, plen, nlev), side, cat in zip(rows, ("left", "right"), (cat_l, cat_r)):
            deepest = max(4, -(-(int(nlev.max()) if len(nlev) else 1) // 4) * 4)
            sides.append(tables.SetTable.from_nested_arrays(ids, plen[:, :deepest], nlev, side, dev, categories=cat,
                                                            width=width, category_mode=cat_mode, partition=part))
        lt, rt = sides
    else:
        vocab = tables.Vocabulary()
        sl, sr = [as_set_levels(it) for it in levels_l], [as_set_levels(it) for it in levels_r]
        width = tables.pick_width(
            max((len(set(it[-1])) for it in sl if it), default=1), max((len(set(it[-1])) for it in sr if it), default=1)
        )
        lt = tables.SetTable.from_levels(sl, "left", dev, vocab, width=width, categories=cat_l, category_mode=cat_mode,
                                         partition=part)
        rt = tables.SetTable.from_levels(sr, "right", dev, vocab, width=width, categories=cat_r,
                                         category_mode=cat_mode, partition=part)
    if len(vocab) >= 1 << 25:
        raise NotImplementedError("vocabulary of 2^25 or more distinct tokens")
    # the library picks the inverted-index kernel from the threshold alone (it cannot see the vocabulary); the host
    # can: with a large vocabulary few pairs share an id and the index wins at every threshold (3 x 100k^2 items of
    # ~8 ids: 20k words 2.1 vs 2.2 ms at 0.7 and 4.2 vs 22 ms at 0.1; 2^17 words 1.0 vs 2.3 ms and 1.8 vs 20.9 ms)
    use_index = True if (threshold > 0 and len(vocab) >= 8192) else None
    return grid.jaccard_levels_grid(lt, rt, threshold, category_mode=cat_mode, index=use_index, defer=defer)


def _levels_grid(plugin, levels_l, levels_r, threshold, cat_l, cat_r, cat_mode=_lib.CAT_NONE, defer: bool = False):
    """Encode both sides' levels for ``plugin`` and run the levels grid on the current device.  ``defer``: when the
    whole grid goes through ONE fast kernel call, return its hits still on the device (``grid.PendingHits``: the sharded
    ``gen_comparable`` exchanges them without a host detour); grids that are split (wide / irregular items) return
    ``grid.Hits`` as always."""
    import torch

    if not torch.cuda.is_available():
        raise _lib.NsmLibraryError("the match loop runs on an MI355X (HIP device); there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device())
    from .. import wide

    sub = lambda seq, idx: [seq[k] for k in idx]
    cut = lambda cat, idx: None if cat is None else np.asarray(cat)[np.asarray(idx, dtype=np.int64)]
    if plugin.kind == "sets":
        as_set_levels = lambda it: [lv if isinstance(lv, list) else lv.split() for lv in it]
        def split_route(split):
            # items of more than 64 distinct tokens / more than 64 levels / levels that are not suffix-nested leave the fast
            # path (wide.py: general kernel, every step's level sets scored on their own); the rest is scored as always
            fast = lambda li, ri: _levels_grid(plugin, sub(levels_l, li), sub(levels_r, ri), threshold, cut(cat_l, li),
                                               cut(cat_r, ri), cat_mode)
            general = lambda li, ri: wide.jaccard_any_grid(
                [as_set_levels(levels_l[k]) for k in li], [as_set_levels(levels_r[k]) for k in ri], threshold, cut(cat_l, li),
                cut(cat_r, ri), cat_mode, device=dev)
            return wide.split_grid(split[0], split[1], fast, general)

        split = wide.wide_set_items([as_set_levels(it) for it in levels_l], [as_set_levels(it) for it in levels_r]) \
            if _may_be_wide_sets(levels_l, levels_r) else None
        if split is not None:
            return split_route(split)
        try:
            return _fast_jaccard_levels(levels_l, levels_r, as_set_levels, threshold, cat_l, cat_r, cat_mode, dev, defer)
        except tables.IrregularLevels:
            # the reference scores whatever its tokenizer yields per level (:283-299): find the items the nested layout
            # cannot hold and route only them
            split = wide.wide_set_items([as_set_levels(it) for it in levels_l], [as_set_levels(it) for it in levels_r],
                                        nesting=True)
            if split is None:
                raise
            return split_route(split)
    prep = lambda items: ComparableData._memoised(
        "fuzzy", items, lambda it: [score_functions.fuzzy_operand(lv) for lv in it])
    ops_l, ops_r = prep(levels_l), prep(levels_r)
    split = wide.wide_string_items(ops_l, ops_r)
    if split is not None:
        # level strings of more than 512 code units / a grid of more than 255 distinct code units (wide.py)
        def fast(li, ri):
            a, b, c, d = tables.encode_level_strings(sub(ops_l, li), sub(ops_r, ri), dev, cut(cat_l, li), cut(cat_r, ri), cat_mode)
            return grid.indel_levels_grid(a, b, c, d, threshold, category_mode=cat_mode)

        general = lambda li, ri: wide.indel_any_grid(sub(ops_l, li), sub(ops_r, ri), threshold, cut(cat_l, li), cut(cat_r, ri),
                                                     cat_mode, device=dev)
        return wide.split_grid(split[0], split[1], fast, general)
    li, ls, ri, rs = tables.encode_level_strings(ops_l, ops_r, dev, cat_l, cat_r, cat_mode)
    return grid.indel_levels_grid(li, ls, ri, rs, threshold, category_mode=cat_mode, defer=defer)
