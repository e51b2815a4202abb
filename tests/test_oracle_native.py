"""The C oracle (oracle/c) must agree bit for bit with the Python oracle, which is pinned to the
reference by tests/test_oracle_golden.py."""
import random

import numpy as np
import pytest

from oracle import compare as oc
from oracle import native
from oracle import score_functions as osf


def _rand_sets(rng, n, vocab, kmax, allow_empty=False):
    return [sorted(rng.sample(range(vocab), rng.randint(0 if allow_empty else 1, kmax))) for _ in range(n)]


def test_jaccard_raw_matches_python_oracle():
    rng = random.Random(1)
    left, right = _rand_sets(rng, 60, 30, 8), _rand_sets(rng, 70, 30, 8)
    for thr in (0.0, 0.1, 0.34, 0.5, 1.0):
        want = oc.raw_grid_hits([[str(v) for v in s] for s in left], [[str(v) for v in s] for s in right],
                                "intersection_vs_union", thr)
        got = native.jaccard_raw(native.csr(left), native.csr(right), thr)
        assert got == want


def test_jaccard_raw_zero_division():
    with pytest.raises(ZeroDivisionError):
        native.jaccard_raw(native.csr([[1], []]), native.csr([[2], []]), 0.5)


def test_indel_raw_matches_python_oracle():
    rng = random.Random(2)
    alpha = "abcd efg"

    def word():
        return "".join(rng.choice(alpha) for _ in range(rng.randint(0, 14))).strip()

    left, right = [word() for _ in range(40)], [word() for _ in range(45)]
    for thr in (0.0, 0.3, 0.6, 0.8):
        want = oc.raw_grid_hits(left, right, "fuzzy_match", thr)
        got = native.indel_raw(native.csr([[ord(c) for c in s] for s in left]),
                               native.csr([[ord(c) for c in s] for s in right]), thr)
        assert got == want


def test_levels_match_python_oracle():
    rng = random.Random(3)

    def item():
        base, out = [], []
        for _ in range(rng.randint(1, 5)):
            base = sorted(set(base + rng.sample(range(12), rng.randint(0, 3))))
            out.append(list(base))
        if not out[-1]:
            out[-1] = [0]
        return out

    left, right = [item() for _ in range(30)], [item() for _ in range(35)]
    # avoid 0/0 pairs in this test: make level sets non-empty
    for it in left + right:
        for lv in it:
            if not lv:
                lv.append(11)
    as_str = lambda items: [[[str(v) for v in lv] for lv in it] for it in items]
    for thr in (0.0, 0.2, 0.5):
        want = oc.matcher_grid_hits(as_str(left), as_str(right), "intersection_vs_union", thr)
        got = native.levels(False, left, right, thr)
        assert got == want
    # Indel over per-level strings (levels given as code point lists)
    to_s = lambda lv: osf.default_process(osf.join_sorted([f"w{v}" for v in lv]))
    sl = [[to_s(lv) for lv in it] for it in left]
    sr = [[to_s(lv) for lv in it] for it in right]
    for thr in (0.0, 0.4, 0.7):
        want = oc.matcher_grid_hits(sl, sr, "fuzzy_match", thr)
        got = native.levels(True, [[[ord(c) for c in s] for s in it] for it in sl],
                            [[[ord(c) for c in s] for s in it] for it in sr], thr)
        assert got == want


def test_levels_errors_and_categories():
    with pytest.raises(ZeroDivisionError):
        native.levels(False, [[[], [], [1]]], [[[], []]], 0.0)
    with pytest.raises(IndexError):
        native.levels(False, [[]], [[[1]]], 0.0)
    assert native.levels(False, [[]], [[]], 0.0) == [(0.0, 0, 0)]
    lc, rc = np.array([1, 0, 2], np.uint64), np.array([1, 0], np.uint64)
    items_l, items_r = [[[1]], [[1]], [[1]]], [[[1]], [[1]]]
    assert [(i, j) for _, i, j in native.levels(False, items_l, items_r, 0.0, lc, rc, 2)] == [(0, 0), (1, 1)]
    assert [(i, j) for _, i, j in native.levels(False, items_l, items_r, 0.0, lc, rc, 1)] == [(0, 0)]
