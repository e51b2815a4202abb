"""The score_func plugin module, MI355X edition.

Same names, call convention and error behaviour as the reference's
``napkon_string_matching/compare/score_functions.py`` (:6-27): callables
``(left, right) -> float`` looked up BY NAME (types/comparable_data.py:150), plus
``join_sorted``.  Each plugin has two faces:

* calling it scores ONE pair -- as a 1 x 1 grid on the GPU, through the same kernels as the
  batched path (there is no CPU implementation of the arithmetic in this package);
* ``plugin.raw_grid(left_items, right_items, threshold)`` / the levels builders used by
  ``ComparableData.gen_comparable`` score N x M pairs in one launch.

``default_process`` / ``join_sorted`` are per-item string preparation and stay on the host; the
reference re-does them for every pair (score_functions.py:24-25 inside the hot loop).
"""
from __future__ import annotations

import re
from typing import Iterable, List, Sequence, Union

import torch

from .. import grid, tables

_NON_WORD = re.compile(r"\W", re.UNICODE)        # keeps "_": what rapidfuzz 2.1's pure-Python fallback does
_NON_ALNUM = re.compile(r"[\W_]", re.UNICODE)    # blanks "_" too: "non alphanumeric", the compiled implementation

# How ``default_process`` treats "_" -- the one point where rapidfuzz 2.1's two implementations of it differ:
# the C++ one (what a pip-installed wheel runs, and what its documentation describes: "removing all non
# alphanumeric characters") blanks it, the pure-Python fallback (``re.sub(r"(?ui)\W", " ", s)``) keeps it.
# rapidfuzz is not installable offline, so this cannot be pinned against the pinned version; the switch is
# explicit instead of buried in a regex.  Synthetic corpora only use ``[a-z0-9 ]``, a fixed point of both.
UNDERSCORE_POLICIES = ("blank", "keep")
UNDERSCORE_POLICY = "blank"

Operand = Union[str, List[str]]


def _device():
    if not torch.cuda.is_available():
        from .._lib import NsmLibraryError

        raise NsmLibraryError("score functions run on an MI355X (HIP device); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def join_sorted(value: Sequence[str]) -> str:
    """score_functions.py:16-17."""
    return " ".join(sorted(value, key=str.lower))


def default_process(text: str, underscore: str = None) -> str:
    """rapidfuzz 2.x ``utils.default_process`` (applied by ``fuzz.QRatio`` by default in the
    pinned 2.1 line): non-alphanumeric code points -> blank, strip, lower-case.  ``underscore``:
    "blank" or "keep" (default: the module's ``UNDERSCORE_POLICY``)."""
    policy = UNDERSCORE_POLICY if underscore is None else underscore
    if policy not in UNDERSCORE_POLICIES:
        raise ValueError(f"underscore policy must be one of {UNDERSCORE_POLICIES}")
    return (_NON_ALNUM if policy == "blank" else _NON_WORD).sub(" ", text).strip().lower()


def fuzzy_operand(value: Operand) -> str:
    """What ``fuzzy_match`` feeds to the Indel ratio for one operand (score_functions.py:24-25
    followed by QRatio's default processor)."""
    return default_process(join_sorted(value) if isinstance(value, list) else value)


def set_operand(value: Operand) -> Iterable[str]:
    """What ``intersection_vs_union`` turns one operand into (score_functions.py:10-11)."""
    return value if isinstance(value, list) else value.split()


class _IntersectionVsUnion:
    __name__ = "intersection_vs_union"
    kind = "sets"

    def __call__(self, left: Operand, right: Operand) -> float:
        hits = self.raw_grid([left], [right], float("-inf"))
        return float(hits.score[0])

    @staticmethod
    def raw_grid(left_items: Sequence[Operand], right_items: Sequence[Operand], threshold: float, device=None,
                 prune: bool = True) -> grid.Hits:
        from .. import wide

        dev = device or _device()
        l_rows = [list(set_operand(v)) for v in left_items]
        r_rows = [list(set_operand(v)) for v in right_items]

        def fast(li, ri):
            vocab = tables.Vocabulary()
            ls, rs = [l_rows[k] for k in li], [r_rows[k] for k in ri]
            width = tables.pick_width(max((len(set(r)) for r in ls), default=1), max((len(set(r)) for r in rs), default=1))
            lt = tables.SetTable.from_rows(ls, "left", dev, vocab, width=width)
            rt = tables.SetTable.from_rows(rs, "right", dev, vocab, width=width)
            return grid.jaccard_raw_grid(lt, rt, threshold, prune=prune)

        split = wide.wide_set_items([[r] for r in l_rows], [[r] for r in r_rows])
        if split is None:
            return fast(range(len(l_rows)), range(len(r_rows)))
        # items of more than 64 distinct tokens (the reference's set(...) has no size limit, score_functions.py:10-13)
        if any(not r for r in l_rows) and any(not r for r in r_rows):
            raise ZeroDivisionError("division by zero")  # (:13, checked on the whole grid before it is split)
        general = lambda li, ri: wide.jaccard_any_grid([[l_rows[k]] for k in li], [[r_rows[k]] for k in ri], threshold,
                                                       raw=True, device=dev)
        return wide.split_grid(split[0], split[1], fast, general)


class _FuzzyMatch:
    __name__ = "fuzzy_match"
    kind = "strings"

    def __call__(self, left: Operand, right: Operand) -> float:
        hits = self.raw_grid([left], [right], float("-inf"))
        return float(hits.score[0])

    @staticmethod
    def raw_grid(left_items: Sequence[Operand], right_items: Sequence[Operand], threshold: float, device=None,
                 prune: bool = True) -> grid.Hits:
        from .. import wide

        dev = device or _device()
        l_ops, r_ops = [fuzzy_operand(v) for v in left_items], [fuzzy_operand(v) for v in right_items]

        def fast(li, ri):
            lt, rt = tables.encode_strings([l_ops[k] for k in li], [r_ops[k] for k in ri], dev)
            return grid.indel_raw_grid(lt, rt, threshold, prune=prune)

        split = wide.wide_string_items([[s] for s in l_ops], [[s] for s in r_ops])
        if split is None:
            return fast(range(len(l_ops)), range(len(r_ops)))
        # strings of more than 512 code units / more than 255 distinct code units (rapidfuzz has no such limit, :27)
        general = lambda li, ri: wide.indel_any_grid([[l_ops[k]] for k in li], [[r_ops[k]] for k in ri], threshold, raw=True,
                                                     device=dev)
        return wide.split_grid(split[0], split[1], fast, general)


intersection_vs_union = _IntersectionVsUnion()
fuzzy_match = _FuzzyMatch()
