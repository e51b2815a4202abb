"""GPU parity of the four grid kernels, called through the C ABI (ctypes), against the oracle.

Bar: Jaccard scores and hit indices bit-exact; Indel ratios within 1e-6 (they are in fact compared
bit-exact too, the tolerance only documents what BASELINE.json's north_star asks for).
"""
import random

import numpy as np
import torch
import pytest

pytestmark = pytest.mark.gpu

FUZZY_TOL = 1e-6


@pytest.fixture(scope="module")
def dev():
    import torch

    return torch.device("cuda:0")


def _same_hits(got, want, tol=0.0):
    got_t = got.as_tuples()
    assert len(got_t) == len(want), f"{len(got_t)} hits, oracle has {len(want)}"
    if tol == 0.0:
        assert got_t == want
    else:
        assert [(i, j) for _, i, j in got_t] == [(i, j) for _, i, j in want]
        for (s, _, _), (w, _, _) in zip(got_t, want):
            assert abs(s - w) <= tol


def _rand_padded(rng, n, width, vocab, kmax, allow_empty):
    ids = np.full((n, width), -1, dtype=np.int32)
    for r in range(n):
        k = rng.randint(0 if allow_empty else 1, kmax)
        ids[r, :k] = rng.sample(range(vocab), k)
    return ids


@pytest.mark.parametrize("width,kmax,vocab", [(16, 16, 60), (16, 6, 25), (32, 32, 90), (64, 64, 200), (64, 40, 5000)])
@pytest.mark.parametrize("prune", [False, True])
def test_jaccard_raw_random(dev, width, kmax, vocab, prune):
    from napkon_string_matching_amd import grid, tables
    from oracle import native

    rng = random.Random(width * 1000 + kmax)
    left = _rand_padded(rng, 333, width, vocab, kmax, allow_empty=False)
    right = _rand_padded(rng, 517, width, vocab, kmax, allow_empty=True)
    lt = tables.SetTable.from_padded(left, "left", dev, width=width)
    rt = tables.SetTable.from_padded(right, "right", dev, width=width)
    for thr in (0.0, 0.05, 0.2, 1 / 3, 0.5, 0.75, 1.0, 1.5):
        want = native.jaccard_raw(native.csr_from_padded(left), native.csr_from_padded(right), thr, cap=1 << 18)
        got = grid.jaccard_raw_grid(lt, rt, thr, prune=prune, capacity=1 << 12)  # small: exercises the retry
        _same_hits(got, want)


@pytest.mark.parametrize("width,kmax,vocab,n_left,n_right", [
    (16, 16, 60, 333, 517),         # tiny vocabulary: every pair is a candidate, the early tiles are dense
    (16, 10, 5000, 2100, 1700),     # sparse: most probes miss
    (16, 16, 100000, 900, 2500),    # all rows full: every tile is dense (indexed in two passes of 32 lanes)
    (32, 30, 700, 700, 900),
    (32, 12, 90, 300, 400),
])
def test_jaccard_raw_inverted_index(dev, width, kmax, vocab, n_left, n_right):
    """Candidate generation by inverted index (low thresholds) == the matrix kernel == the oracle, forced on
    and chosen by the library; partial last tile, empty sets, dense and sparse tiles."""
    from napkon_string_matching_amd import grid, tables
    from oracle import native

    rng = random.Random(width * 77 + vocab)
    left = _rand_padded(rng, n_left, width, vocab, kmax, allow_empty=False)  # (empty x empty is the host's ZeroDivisionError)
    right = _rand_padded(rng, n_right, width, vocab, kmax, allow_empty=True)
    if vocab == 100000:
        for row in right:
            row[:] = rng.sample(range(vocab), width)
    lt = tables.SetTable.from_padded(left, "left", dev, width=width)
    rt = tables.SetTable.from_padded(right, "right", dev, width=width)
    for thr in (0.01, 0.1, 0.25, 0.5, 1.0):
        want = native.jaccard_raw(native.csr_from_padded(left), native.csr_from_padded(right), thr, cap=1 << 20)
        forced = grid.jaccard_raw_grid(lt, rt, thr, index=True, capacity=1 << 12)
        _same_hits(forced, want)
        _same_hits(grid.jaccard_raw_grid(lt, rt, thr, capacity=1 << 12), want)               # the library's choice
        _same_hits(grid.jaccard_raw_grid(lt, rt, thr, index=False, capacity=1 << 12), want)  # never the index


@pytest.mark.parametrize("width,kmax,vocab,n_left,n_right", [
    (16, 16, 60, 333, 517),         # tiny vocabulary: long posting lists, most candidates survive the cheap filters
    (16, 10, 5000, 2100, 1700),     # sparse
    (16, 16, 100000, 900, 2500),    # every row full
    (16, 3, 40, 500, 700),          # tiny sets: prefix = the whole row
    (32, 30, 700, 700, 900),
    (32, 12, 90, 300, 400),
    (64, 64, 3000, 200, 300),
    (64, 40, 150, 150, 260),
])
def test_jaccard_raw_global_index(dev, width, kmax, vocab, n_left, n_right):
    """Candidate generation from the right table's GLOBAL inverted index (prefix filter: probe the posting lists of the
    first few ids of every left row) == the per-tile index == the signature kernel == the oracle, at thresholds from
    "one common id suffices" (prefix = the whole row) to "identical sets only" (prefix = one id), with planted
    near-duplicates, empty sets, ids the right side never uses, and a left table whose ids arrive unsorted."""
    from napkon_string_matching_amd import grid, tables
    from oracle import native

    rng = random.Random(width * 91 + vocab + kmax)
    left = _rand_padded(rng, n_left, width, vocab, kmax, allow_empty=False)
    right = _rand_padded(rng, n_right, width, vocab, kmax, allow_empty=True)
    for t in range(0, n_right, 7):  # near-duplicates: a left row with one id swapped or dropped
        row = [v for v in left[rng.randrange(n_left)] if v >= 0]
        if len(row) > 1 and rng.random() < 0.5:
            row[rng.randrange(len(row))] = vocab + rng.randrange(50)  # an id above everything else
        elif len(row) > 1 and rng.random() < 0.5:
            row.pop(rng.randrange(len(row)))
        rng.shuffle(row)
        right[t] = row + [-1] * (width - len(row))
    left[5][: kmax] = sorted((v for v in left[5][: kmax]), reverse=True)  # unsorted input rows
    lt = tables.SetTable.from_padded(left, "left", dev, width=width)
    rt = tables.SetTable.from_padded(right, "right", dev, width=width)
    assert rt.post is not None and lt.post is None and rt.vocab == int(np.max(right)) + 1
    plain = tables.SetTable.from_padded(right, "right", dev, width=width, index=False)
    assert plain.post is None
    assert rt.post_row_bits > 0 and rt.post_format == 2  # compact entries with the signature fold; the plain 64-bit ones:
    tables.COMPACT_POSTINGS = False
    try:
        rt64 = tables.SetTable.from_padded(right, "right", dev, width=width)
    finally:
        tables.COMPACT_POSTINGS = True
    assert rt64.post_row_bits == 0 and rt64.post_format == 0
    for thr in (0.01, 0.1, 0.25, 1 / 3, 0.5, 0.6, 0.75, 0.8, 0.9, 1.0):
        want = native.jaccard_raw(native.csr_from_padded(left), native.csr_from_padded(right), thr, cap=1 << 20)
        forced = grid.jaccard_raw_grid(lt, rt, thr, index=True, capacity=1 << 12)
        _same_hits(forced, want)
        _same_hits(grid.jaccard_raw_grid(lt, rt64, thr, index=True, capacity=1 << 12), want)
        _same_hits(grid.jaccard_raw_grid(lt, rt, thr, capacity=1 << 12), want)                 # the library's choice
        _same_hits(grid.jaccard_raw_grid(lt, rt, thr, index=False, capacity=1 << 12), want)    # all pairs
        if width <= 32:
            _same_hits(grid.jaccard_raw_grid(lt, rt, thr, index="tile", capacity=1 << 12), want)   # per-tile LDS index
            _same_hits(grid.jaccard_raw_grid(lt, plain, thr, index=True, capacity=1 << 12), want)  # no global index to force
    assert len(want) > 0  # (threshold 1.0: the identical planted rows)
    # a table that declares a posting format the kernels cannot decode for its rows is refused, not read
    from napkon_string_matching_amd import _lib
    for fmt, bits in ((7, rt.post_row_bits), (2, 0), (0, 24), (1, 3)):
        keep = (rt.post_format, rt.post_row_bits)
        rt.post_format, rt.post_row_bits = fmt, bits
        try:
            with pytest.raises(_lib.NsmLibraryError, match="post_format"):
                grid.jaccard_raw_grid(lt, rt, 0.5, index=True)
        finally:
            rt.post_format, rt.post_row_bits = keep
    # thresholds <= 0: every pair hits, an index cannot help and is not used
    want0 = native.jaccard_raw(native.csr_from_padded(left[:40]), native.csr_from_padded(right[:50]), 0.0, cap=1 << 20)
    lt0 = tables.SetTable.from_padded(left[:40], "left", dev, width=width)
    rt0 = tables.SetTable.from_padded([r for r in right[:50]], "right", dev, width=width)
    if not (lt0.has_empty and rt0.has_empty):
        _same_hits(grid.jaccard_raw_grid(lt0, rt0, 0.0, index=True), want0)


def test_jaccard_raw_caller_ids_beyond_row_count(dev):
    """Tables built with caller ids (``orig``) far above the row count -- a shard of a larger cohort -- through the radix
    sort (> 8192 hits): the sort's key width comes from the tables' id bound, not from their row counts, and ids with the
    sign bit set fall back to full-width keys."""
    from napkon_string_matching_amd import grid, tables
    from oracle import native

    rng = random.Random(77)
    left = _rand_padded(rng, 300, 16, 40, 12, allow_empty=False)
    right = _rand_padded(rng, 400, 16, 40, 12, allow_empty=False)
    want = native.jaccard_raw(native.csr_from_padded(left), native.csr_from_padded(right), 0.2, cap=1 << 20)
    assert len(want) > 8192
    for lo, ro in ((5_000_000, 2_000_000_000), (0, 70_000), (-7, 3)):
        li = np.arange(300, dtype=np.int32) * 3 + lo
        ri = np.arange(400, dtype=np.int32) + ro
        lt = tables.SetTable.from_padded(left, "left", dev, width=16, orig=li)
        rt = tables.SetTable.from_padded(right, "right", dev, width=16, orig=ri)
        assert lt.id_limit == (0 if lo < 0 else int(li.max()) + 1) and rt.id_limit == int(ri.max()) + 1
        got = grid.jaccard_raw_grid(lt, rt, 0.2)
        exp = sorted(((s, int(li[i]), int(ri[j])) for s, i, j in want), key=lambda h: (-h[0], h[1], h[2]))
        assert [(h[0], h[1], h[2]) for h in got.as_tuples()] == exp


def test_jaccard_raw_c2_shaped_low_threshold(dev):
    """configs[1]'s generator at the API's default threshold 0.1 (types/comparable_data.py:74), 20k x 20k: the
    index kernel against the exhaustive matrix kernel, and the collision-crafted ids of the signature test."""
    from napkon_string_matching_amd import grid, synthetic, tables

    left, right = synthetic.c2_corpus(20000, 20000)
    lt = tables.SetTable.from_padded(left, "left", dev)
    rt = tables.SetTable.from_padded(right, "right", dev)
    exhaustive = grid.jaccard_raw_grid(lt, rt, 0.1, prune=False, index=False).as_tuples()
    assert len(exhaustive) > 1000
    assert grid.jaccard_raw_grid(lt, rt, 0.1).as_tuples() == exhaustive
    assert grid.jaccard_raw_grid(lt, rt, 0.1, index=True).as_tuples() == exhaustive


def test_jaccard_raw_c2_shaped(dev):
    """BASELINE configs[1] generator at 6000 x 5000 (the oracle finishes in seconds)."""
    from napkon_string_matching_amd import grid, synthetic, tables
    from oracle import native

    left, right = synthetic.c2_corpus(6000, 5000)
    lt = tables.SetTable.from_padded(left, "left", dev)
    rt = tables.SetTable.from_padded(right, "right", dev)
    want = native.jaccard_raw(native.csr_from_padded(left), native.csr_from_padded(right), 0.5)
    assert len(want) >= 40  # the planted near-duplicates
    for prune in (False, True):
        _same_hits(grid.jaccard_raw_grid(lt, rt, 0.5, prune=prune), want)


def test_jaccard_raw_edges(dev):
    from napkon_string_matching_amd import grid, tables

    one = np.array([[5, 7, -1, -1]], dtype=np.int32)
    lt = tables.SetTable.from_padded(one, "left", dev)
    rt = tables.SetTable.from_padded(np.array([[7, 5, 9, -1], [1, 2, 3, 4]], dtype=np.int32), "right", dev)
    got = grid.jaccard_raw_grid(lt, rt, 0.5)
    assert got.as_tuples() == [(2 / 3, 0, 0)]
    # empty grid
    empty = tables.SetTable.from_padded(np.zeros((0, 4), dtype=np.int32), "left", dev, width=16)
    assert len(grid.jaccard_raw_grid(empty, rt, 0.0)) == 0
    # empty-vs-empty is the reference's ZeroDivisionError (score_functions.py:13)
    le = tables.SetTable.from_padded(np.array([[1, -1], [-1, -1]], dtype=np.int32), "left", dev)
    re_ = tables.SetTable.from_padded(np.array([[-1, -1], [1, 2]], dtype=np.int32), "right", dev)
    with pytest.raises(ZeroDivisionError):
        grid.jaccard_raw_grid(le, re_, 0.1)
    # one empty side only: scores 0.0, no error
    ro = tables.SetTable.from_padded(np.array([[1, 2]], dtype=np.int32), "right", dev)
    assert grid.jaccard_raw_grid(le, ro, 0.0).as_tuples() == [(0.5, 0, 0), (0.0, 1, 0)]
    with pytest.raises(ValueError):
        tables.SetTable.from_padded(np.array([[3, 3]], dtype=np.int32), "left", dev)
    with pytest.raises(ValueError):
        grid.jaccard_raw_grid(lt, lt, 0.5)  # both encoded as 'left': same padding value


def test_jaccard_signature_collisions(dev):
    """Rows whose ids all fall on ONE signature bit: up to 6 in-row collisions ride in the word's top
    bits, rows with more are encoded as all ones (always pass the bound).  RAW and levels grids."""
    from napkon_string_matching_amd import grid, tables
    from oracle import native

    cand = np.arange(400_000, dtype=np.int32)
    h16 = ((cand.astype(np.uint32) * np.uint32(0x9E3779B1)) >> np.uint32(16)) & np.uint32(0xFFFF)
    same = cand[((h16.astype(np.uint64) * np.uint64(58)) >> np.uint64(16)) == 5]  # every id lands on bit 5
    assert len(same) > 64
    rng = random.Random(5)
    rows = []
    for k in (2, 5, 7, 8, 12, 16):  # 1 .. 15 collisions inside the row
        base = rng.sample(list(map(int, same)), k)
        rows += [base, base[: max(1, k - 1)] + [int(same[-1])], base[: k // 2 + 1] + rng.sample(range(1000), 3)]
    rows += [rng.sample(range(5000), rng.randint(1, 12)) for _ in range(80)]
    left = [list(dict.fromkeys(r))[:16] for r in rows]
    right = [list(r) for r in left[::-1]] + [rng.sample(list(map(int, same)), 10) for _ in range(30)]
    sig = tables.signatures(np.array([left[15] + [-1] * (16 - len(left[15]))], dtype=np.int32), np.array([len(left[15])], np.int32))
    assert int(sig[0]) == (1 << 64) - 1  # 16 ids on one bit: 15 collisions -> all ones
    # (from_rows would renumber the tokens: encode the raw ids, so that the hash positions are the crafted ones)
    pad = lambda rr: np.array([r + [-1] * (16 - len(r)) for r in rr], dtype=np.int32)
    lt, rt = tables.SetTable.from_padded(pad(left), "left", dev), tables.SetTable.from_padded(pad(right), "right", dev)
    for thr in (0.2, 0.5, 0.8, 1.0):
        want = native.jaccard_raw(native.csr(left), native.csr(right), thr, cap=1 << 16)
        assert len(want) > 5
        for prune in (True, False):
            _same_hits(grid.jaccard_raw_grid(lt, rt, thr, prune=prune), want)
    # levels: each row as a two-level item (first half, whole row)
    items = lambda rr: [[r[: max(1, len(r) // 2)], r] for r in rr]
    nested = lambda rr, side: tables.SetTable.from_nested_arrays(
        pad(rr), np.array([[max(1, len(r) // 2), len(r)] for r in rr], dtype=np.uint8), np.full(len(rr), 2, np.int32), side, dev)
    lt2, rt2 = nested(left, "left"), nested(right, "right")
    for thr in (0.2, 0.5, 0.7):
        want = native.levels(False, items(left), items(right), thr, cap=1 << 16)
        assert len(want) > 5
        _same_hits(grid.jaccard_levels_grid(lt2, rt2, thr), want)


def _rand_strings(rng, n, alphabet, lo, hi):
    out = []
    for _ in range(n):
        k = rng.randint(lo, hi)
        out.append("".join(rng.choice(alphabet) for _ in range(k)))
    return out


@pytest.mark.parametrize("alphabet,lo,hi", [("ab", 0, 12), ("abcdefghijklmnopqrstuvwxyz0123456789 ", 0, 64),
                                             ("abcdefgh", 30, 64), ("".join(chr(0x100 + k) for k in range(200)), 1, 64)])
@pytest.mark.parametrize("prune", [False, True])
def test_indel_raw_random(dev, alphabet, lo, hi, prune):
    from napkon_string_matching_amd import grid, tables
    from oracle import native

    rng = random.Random(len(alphabet) * 100 + hi)
    left = _rand_strings(rng, 150, alphabet, lo, hi)
    right = _rand_strings(rng, 210, alphabet, lo, hi)
    right[7] = left[3]
    right[8] = left[4][:-1] + "a" if left[4] else "a"
    lt, rt = tables.encode_strings(left, right, dev)
    cp = lambda ss: native.csr([[ord(c) for c in s] for s in ss])
    for thr in (0.0, 0.3, 0.5, 0.8, 1.0):
        want = native.indel_raw(cp(left), cp(right), thr, cap=1 << 16)
        got = grid.indel_raw_grid(lt, rt, thr, prune=prune, capacity=1 << 10)
        _same_hits(got, want, FUZZY_TOL)
        _same_hits(got, want)  # and in fact bit-exact
        if prune and lt.hist16 is not None:  # the 32-bucket test for every pair (no 16-bucket first stage)
            _same_hits(grid.indel_raw_grid(lt, rt, thr, two_stage=False, capacity=1 << 10), want)


def test_indel_raw_two_stage_filter(dev):
    """The two-stage histogram filter of the RAW grid on 64-unit strings (16 buckets for every pair, 32 buckets for the
    pairs that pass, the LCS of the rest on the scalar unit) where its queue is busy: a 4-letter alphabet (most pairs pass
    the coarse test, entries carry several pairs and go back on the stack), strings of every length incl. empty ones, more
    right rows than two tiles per wave divide, thresholds down to 0 (every pair is a hit) -- against the oracle, the
    one-stage kernel and the exhaustive kernel."""
    from napkon_string_matching_amd import grid, tables
    from oracle import native

    rng = random.Random(4242)
    for alphabet, n_left, n_right, thresholds in (("abcd", 700, 531, (0.0, 0.35, 0.6, 0.8, 1.0)),
                                                  ("abcdefghijklmnopqrstuvwxyz0123456789 .", 900, 777, (0.3, 0.5, 0.7))):
        left = _rand_strings(rng, n_left, alphabet, 0, 64)
        right = _rand_strings(rng, n_right, alphabet, 0, 64)
        for k in range(0, n_right, 5):  # near-duplicates: a left string with one code unit changed or dropped
            src = left[rng.randrange(n_left)]
            if src:
                pos = rng.randrange(len(src))
                right[k] = src[:pos] + (rng.choice(alphabet) if rng.random() < 0.5 else "") + src[pos + 1:]
        lt, rt = tables.encode_strings(left, right, dev)
        assert lt.stride == 64 and lt.hist16 is not None and rt.hist16 is not None
        cp = lambda ss: native.csr([[ord(c) for c in s] for s in ss])
        base = native.indel_raw(cp(left), cp(right), min(thresholds), cap=1 << 20)
        for thr in thresholds:
            want = [h for h in base if h[0] >= thr]  # (the oracle's list is ordered by score already)
            assert len(want) > 0
            _same_hits(grid.indel_raw_grid(lt, rt, thr), want)
            _same_hits(grid.indel_raw_grid(lt, rt, thr, two_stage=False), want)
            _same_hits(grid.indel_raw_grid(lt, rt, thr, prune=False), want)


def test_indel_raw_c3_shaped(dev):
    from napkon_string_matching_amd import grid, synthetic, tables
    from oracle import native

    (lc, ll), (rc, rl) = synthetic.c3_corpus(1500, 1200)
    lt = tables.StrTable.from_codes(lc, ll, len(synthetic.STRING_ALPHABET), dev)
    rt = tables.StrTable.from_codes(rc, rl, len(synthetic.STRING_ALPHABET), dev)
    want = native.indel_raw(native.csr_from_codes(lc, ll), native.csr_from_codes(rc, rl), 0.8)
    assert len(want) >= 10
    for prune in (False, True):
        _same_hits(grid.indel_raw_grid(lt, rt, 0.8, prune=prune), want, FUZZY_TOL)
    _same_hits(grid.indel_raw_grid(lt, rt, 0.8, two_stage=False), want, FUZZY_TOL)


def test_indel_known_answers(dev):
    """Hand-derived ratios (fuzzy_match is parity unpinned: rapidfuzz is not available offline)."""
    from napkon_string_matching_amd import grid, tables

    pairs = [("kitten", "sitting", 8 / 13), ("lewenstein", "levenshtein", 18 / 21),
             ("dialyse", "dialyse nach entlassung", 14 / 30), ("this is a test", "this is a test", 1.0)]
    lt, rt = tables.encode_strings([p[0] for p in pairs], [p[1] for p in pairs], dev)
    got = {(i, j): s for s, i, j in grid.indel_raw_grid(lt, rt, 0.0).as_tuples()}
    assert len(got) == 16
    for k, (_, _, want) in enumerate(pairs):
        assert abs(got[(k, k)] - want) <= FUZZY_TOL
    with pytest.raises(NotImplementedError):
        tables.encode_strings(["x" * 513], ["y"], dev)


@pytest.mark.parametrize("hi", [100, 128, 200, 256, 300, 512])
@pytest.mark.parametrize("prune", [False, True])
def test_indel_raw_long_strings(dev, hi, prune):
    """Strings of 65..512 code units: the multi-word LCS (stride 128 / 256 / 512)."""
    from napkon_string_matching_amd import grid, tables
    from oracle import native

    rng = random.Random(hi)
    alphabet = "abcdefghijklmnopqrstuvwxyz "
    left = _rand_strings(rng, 90, alphabet, 0, hi)
    right = _rand_strings(rng, 140, alphabet, 0, hi)
    left[0] = "".join(rng.choice(alphabet) for _ in range(hi))
    right[5] = left[0]
    right[6] = left[0][: hi // 2] + "zz" + left[0][hi // 2:hi - 2]
    right[7] = left[1]
    lt, rt = tables.encode_strings(left, right, dev)
    assert lt.stride == rt.stride == (128 if hi <= 128 else 256 if hi <= 256 else 512)
    cp = lambda ss: native.csr([[ord(c) for c in s] for s in ss])
    for thr in (0.0, 0.4, 0.6, 0.9):
        want = native.indel_raw(cp(left), cp(right), thr, cap=1 << 16)
        got = grid.indel_raw_grid(lt, rt, thr, prune=prune, capacity=1 << 10)
        _same_hits(got, want, FUZZY_TOL)
        _same_hits(got, want)


def test_indel_raw_term_like_strings(dev):
    """Word-structured text with a skewed letter distribution (what real Term strings look like): the
    histogram bound is weak there, so the in-scan early exit of the multi-word LCS decides most pairs."""
    from napkon_string_matching_amd import grid, tables
    from oracle import native

    rng = random.Random(77)
    letters = "eeeeeennnniiisssrrraaatttddhhuullccggmmoobbwwffkkzzvvppjyxq"
    words = ["".join(rng.choice(letters) for _ in range(rng.randint(3, 12))) for _ in range(400)]
    text = lambda lo, hi: " ".join(rng.choice(words[: rng.choice([20, 400])]) for _ in range(rng.randint(lo, hi)))[:256].strip()
    left = [text(2, 30) for _ in range(300)]
    right = [text(2, 30) for _ in range(450)]
    for k in range(0, 450, 9):  # near-duplicates: one word replaced
        src = left[rng.randrange(300)].split(" ")
        src[rng.randrange(len(src))] = rng.choice(words)
        right[k] = " ".join(src)[:256].strip()
    lt, rt = tables.encode_strings(left, right, dev)
    assert lt.stride == 256
    cp = lambda ss: native.csr([[ord(c) for c in s] for s in ss])
    base = native.indel_raw(cp(left), cp(right), 0.55, cap=1 << 18)
    for thr in (0.55, 0.8, 0.95):
        want = [h for h in base if h[0] >= thr]  # the oracle's list is ordered by score already
        assert len(want) > 10
        for prune in (True, False):
            got = grid.indel_raw_grid(lt, rt, thr, prune=prune)
            _same_hits(got, want, FUZZY_TOL)
            _same_hits(got, want)


def _nested_item(rng, vocab, max_levels, max_new, allow_empty_levels=False):
    base, out = [], []
    for _ in range(rng.randint(1, max_levels)):
        new = rng.sample(range(vocab), rng.randint(0 if allow_empty_levels or base else 1, max_new))
        for v in new:
            if v not in base:
                base.append(v)
        out.append(list(base))
    return out


@pytest.mark.parametrize("vocab,max_levels,max_new", [(40, 5, 3), (200, 8, 2), (30, 20, 1), (300, 6, 9)])
def test_jaccard_levels_random(dev, vocab, max_levels, max_new):
    from napkon_string_matching_amd import _lib, grid, tables
    from oracle import native

    rng = random.Random(vocab + max_levels)
    left = [_nested_item(rng, vocab, max_levels, max_new) for _ in range(230)]
    right = [_nested_item(rng, vocab, max_levels, max_new) for _ in range(310)]
    for k in range(0, 300, 7):  # near-duplicates, so that the high thresholds have hits
        src = [list(lv) for lv in left[rng.randrange(len(left))]]
        if k % 14 == 0 and len(src) > 1:
            src = src[:-1]
        right[k] = src
    lcat = np.array([rng.choice([0, 1, 2, 3, 6]) for _ in left], dtype=np.uint64)
    rcat = np.array([rng.choice([0, 1, 2, 4, 5]) for _ in right], dtype=np.uint64)
    width = tables.pick_width(max(len(it[-1]) for it in left + right))
    for mode, partition in ((_lib.CAT_NONE, True), (_lib.CAT_INTERSECT, True), (_lib.CAT_INTERSECT_OR_BOTH_EMPTY, True),
                            (_lib.CAT_INTERSECT_OR_BOTH_EMPTY, False)):
        vocabulary = tables.Vocabulary()
        lt = tables.SetTable.from_levels(left, "left", dev, vocabulary, width=width, categories=lcat,
                                         category_mode=mode, partition=partition)
        rt = tables.SetTable.from_levels(right, "right", dev, vocabulary, width=width, categories=rcat,
                                         category_mode=mode, partition=partition)
        tables.COMPACT_POSTINGS = False  # (the same table with 64-bit posting entries)
        try:
            rt64 = tables.SetTable.from_levels(right, "right", dev, vocabulary, width=width, categories=rcat,
                                               category_mode=mode, partition=partition)
        finally:
            tables.COMPACT_POSTINGS = True
        assert rt.post_format == 1 and rt.post.element_size() == 4 and rt64.post_format == 0  # (levels tables: 32-bit entries)
        assert (lt.seg is not None) == (partition and mode != _lib.CAT_NONE)
        for thr in (0.0, 0.1, 0.3, 0.6, 0.8, 0.93):
            want = native.levels(False, left, right, thr, lcat, rcat, mode, cap=1 << 17)
            assert thr > 0.9 or len(want) > 0
            got = grid.jaccard_levels_grid(lt, rt, thr, category_mode=mode, capacity=1 << 12)
            _same_hits(got, want)
            if thr > 0.0:
                # candidates from the right table's global inverted index (postings by (category segment, id)) / from
                # the per-tile LDS index / from the signature filter over all pairs
                assert rt.post is not None and lt.post is None
                _same_hits(grid.jaccard_levels_grid(lt, rt, thr, category_mode=mode, capacity=1 << 12, index=True), want)
                _same_hits(grid.jaccard_levels_grid(lt, rt64, thr, category_mode=mode, capacity=1 << 12, index=True), want)
                _same_hits(grid.jaccard_levels_grid(lt, rt, thr, category_mode=mode, capacity=1 << 12, index=False), want)
                if width <= 32:
                    _same_hits(grid.jaccard_levels_grid(lt, rt, thr, category_mode=mode, capacity=1 << 12, index="tile"), want)


def test_jaccard_levels_identical_items(dev):
    """compare_terms on identical items gives 0.5 / 0.75 / 0.875 for 1 / 2 / 3 levels."""
    from napkon_string_matching_amd import grid, tables

    items = [[[1]], [[1], [1, 2]], [[1], [1, 2], [1, 2, 3]]]
    v = tables.Vocabulary()
    lt = tables.SetTable.from_levels(items, "left", dev, v)
    rt = tables.SetTable.from_levels(items, "right", dev, v)
    got = {(i, j): s for s, i, j in grid.jaccard_levels_grid(lt, rt, 0.0).as_tuples()}
    assert got[(0, 0)] == 0.5 and got[(1, 1)] == 0.75 and got[(2, 2)] == 0.875
    with pytest.raises(NotImplementedError):
        tables.SetTable.from_levels([[[1, 2], [2, 3]]], "left", dev, v)  # not suffix-nested


@pytest.mark.parametrize("max_levels", [1, 4, 9])
def test_indel_levels_random(dev, max_levels):
    from napkon_string_matching_amd import _lib, grid, tables
    from oracle import native

    rng = random.Random(77 + max_levels)
    alphabet = "abcdefghij klm"

    def item():
        return [
            "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 40))).strip()
            for _ in range(rng.randint(1, max_levels))
        ]

    left, right = [item() for _ in range(90)], [item() for _ in range(140)]
    for k in range(0, 140, 6):  # near-duplicates: one level string slightly changed
        src = list(left[rng.randrange(len(left))])
        src[-1] = src[-1][:-1] + "zz"
        right[k] = src
    lcat = np.array([rng.choice([0, 1, 2, 3]) for _ in left], dtype=np.uint64)
    rcat = np.array([rng.choice([0, 1, 2]) for _ in right], dtype=np.uint64)
    cps = lambda items: [[[ord(c) for c in s] for s in it] for it in items]
    for mode, partition in ((_lib.CAT_NONE, True), (_lib.CAT_INTERSECT_OR_BOTH_EMPTY, True), (_lib.CAT_INTERSECT, True),
                            (_lib.CAT_INTERSECT_OR_BOTH_EMPTY, False)):
        li, ls, ri, rs = tables.encode_level_strings(left, right, dev, lcat, rcat, mode, partition=partition)
        assert (li.seg is not None) == (partition and mode != _lib.CAT_NONE)
        for thr in (0.0, 0.25, 0.5, 0.8):
            want = native.levels(True, cps(left), cps(right), thr, lcat, rcat, mode, cap=1 << 16)
            got = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 11)
            _same_hits(got, want, FUZZY_TOL)
            _same_hits(got, want)
            # without the histogram bound (NSM_FLAG_PRUNE off: weights-only bounds) and wave-wide
            _same_hits(grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, prune=False), want)
            _same_hits(grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, wave_wide=True), want)


@pytest.mark.parametrize("thr", [0.0, 0.05, 0.1, 0.3])
def test_indel_levels_park_overflow(dev, thr):
    """Low thresholds keep nearly every pair alive, so the block-shared park of the one-word kernel
    overflows with uneven per-wave counts (partial last tile, category predicate): the cooperative kernel,
    the wave-wide kernel (NSM_FLAG_WAVE_WIDE) and the oracle must agree (round-1 ADVICE: the rolled-back
    reservation raced)."""
    from napkon_string_matching_amd import _lib, grid, tables
    from oracle import native

    rng = random.Random(4242)
    words = ["".join(rng.choice("abcdefgh") for _ in range(rng.randint(2, 5))) for _ in range(40)]

    def item():
        n = rng.randint(3, 6)
        toks = [rng.choice(words) for _ in range(n + 1)]
        return [" ".join(sorted(set(toks[: k + 2]))) for k in range(n)]  # suffix-nested, <= 64 code units

    left, right = [item() for _ in range(83)], [item() for _ in range(64 * 4 * 2 + 37)]
    assert max(len(s) for it in left + right for s in it) <= 64
    lcat = np.array([rng.choice([1, 2, 3, 4, 6]) for _ in left], dtype=np.uint64)
    rcat = np.array([rng.choice([1, 2, 3, 5]) for _ in right], dtype=np.uint64)
    cps = lambda items: [[[ord(c) for c in s] for s in it] for it in items]
    for mode, partition in ((_lib.CAT_INTERSECT, True), (_lib.CAT_INTERSECT, False), (_lib.CAT_NONE, False)):
        li, ls, ri, rs = tables.encode_level_strings(left, right, dev, lcat, rcat, mode, partition=partition)
        assert ls.stride == 64
        want = native.levels(True, cps(left), cps(right), thr, lcat, rcat, mode, cap=1 << 17)
        coop = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 16)
        plain = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 16, wave_wide=True)
        assert coop.as_tuples() == plain.as_tuples()
        _same_hits(coop, want)
        assert len(want) > 5000


@pytest.mark.parametrize("extra", [0, 6])  # 6: level strings of 33..64 code units on both sides (the finish kernel's two-sweep LCS)
@pytest.mark.parametrize("thr", [0.7, 0.8, 0.9])
def test_indel_levels_split_path(dev, thr, extra):
    """One-word level strings at thresholds >= 0.7 take the split path (scan kernel -> survivor queue in the CALLER's
    workspace -> finish kernel).  It must give the hits of the fused park kernel (NSM_FLAG_PARK), of the wave-wide kernel
    and of the oracle -- also when the queue overflows (a workspace of a few hundred bytes forces it: the hit counter is
    put back and the gated fused kernel redoes the grid), and when whole right tiles survive step 1 (the wave's LDS
    buffer is flushed in the middle of a batch)."""
    from napkon_string_matching_amd import _lib, grid, tables
    from oracle import native

    rng = random.Random(9091)
    words = ["".join(rng.choice("abcdefgh") for _ in range(rng.randint(2, 5))) for _ in range(40)]

    def item():
        n = rng.randint(1, 6)
        toks = [rng.choice(words) for _ in range(n + 1 + extra)]
        return [" ".join(sorted(set(toks[: k + 2 + extra])))[:64].strip() for k in range(n)]  # suffix-nested, <= 64 code units

    left = [item() for _ in range(83)]
    right = [item() for _ in range(64 * 4 + 37)]
    for k in range(300):  # three hundred near-copies of a few left items: rows whose 64 lanes all stay alive
        src = list(left[k % 7])
        if k % 3 == 0 and len(src[-1]) < 60:
            src[-1] = src[-1] + " zz"
        right.append(src)
    assert max(len(s) for it in left + right for s in it) <= 64
    lcat = np.array([rng.choice([1, 2, 3, 4, 6]) for _ in left], dtype=np.uint64)
    rcat = np.array([rng.choice([1, 2, 3, 5]) for _ in right], dtype=np.uint64)
    cps = lambda items: [[[ord(c) for c in s] for s in it] for it in items]
    lib = _lib.load()
    for mode, partition in ((_lib.CAT_INTERSECT, True), (_lib.CAT_INTERSECT, False), (_lib.CAT_NONE, False)):
        li, ls, ri, rs = tables.encode_level_strings(left, right, dev, lcat, rcat, mode, partition=partition)
        assert ls.stride == 64
        want = native.levels(True, cps(left), cps(right), thr, lcat, rcat, mode, cap=1 << 17)
        assert len(want) > 50
        fused = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 16, park=True)
        _same_hits(fused, want)
        asked = int(lib.nsm_indel_levels_workspace_bytes(li.struct(), ls.struct(), ri.struct(), rs.struct(), thr, _lib.FLAG_PRUNE, 0.0))
        assert asked > 512 + 16 * 65536  # the grid qualifies for the split path
        assert lib.nsm_indel_levels_workspace_bytes(li.struct(), ls.struct(), ri.struct(), rs.struct(), 0.5, _lib.FLAG_PRUNE, 0.0) == 0
        assert lib.nsm_indel_levels_workspace_bytes(li.struct(), ls.struct(), ri.struct(), rs.struct(), thr,
                                                    _lib.FLAG_PRUNE | _lib.FLAG_PARK, 0.0) == 0
        # None = what the library asks for; 0 = no workspace (single-kernel path); the small ones overflow (queue halves of
        # 32, 100 and 5000 entries), 1024 bytes is the smallest workspace the library uses at all
        for nbytes in (None, 0, 1024, 512 + 16 * 100, 512 + 16 * 5000):
            flag = []
            split = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 16, workspace=nbytes,
                                           return_overflow=flag)
            assert split.as_tuples() == fused.as_tuples(), (mode, partition, nbytes)
            if nbytes is None:
                assert flag == [0]  # the recommended size never overflows here
            elif nbytes == 1024:
                assert flag == [1]  # 32 entries per half: overflow, the gated fused kernel redid the grid
            small = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=64, workspace=nbytes)  # hit buffer grown once
            assert small.as_tuples() == fused.as_tuples()
            # a hit buffer that already holds records: the overflow path puts the counter back to THEIR number
            buf = grid.HitBuffer(1 << 16, dev)
            buf.reset()
            buf.count.fill_(5)
            cm = li.category_mode if li.category_mode is not None else mode
            ws = torch.empty(max(1, (nbytes if nbytes is not None else asked) // 8), dtype=torch.int64, device=dev)
            _lib.check(lib.nsm_indel_levels_grid(li.struct(), ls.struct(), ri.struct(), rs.struct(), thr, int(cm), _lib.FLAG_PRUNE,
                                                 buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(), ws.data_ptr(),
                                                 nbytes if nbytes is not None else asked, 0.0,
                                                 torch.cuda.current_stream(dev).cuda_stream), "nsm_indel_levels_grid")
            assert int(buf.count.item()) == 5 + len(want)
    assert lib.nsm_release(torch.cuda.current_stream(dev).cuda_stream) == 0  # the side stream and events of this stream
    again = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 16)  # ... are re-created on demand
    assert again.as_tuples() == fused.as_tuples()
    assert lib.nsm_release_all() == 0


@pytest.mark.parametrize("partition", [True, False])
def test_indel_levels_split_path_many_rounds(dev, partition):
    """The split path in SEVERAL rounds that do not overflow -- what every large grid runs (configs[4] at 500k rows: 3
    rounds).  A workspace of a fraction of the expected survivors forces >= 3 rounds over the left slices (slice_base /
    slices_total, both queue halves, the scanned / finished events re-used from round 2 on, scan and finish kernels
    appending hits side by side); the overflow word must stay 0, so the gated fused kernel did NOT redo the grid and the
    hits are the rounds' own.  (round-3 advice: the overflowing caps masked the rounds, the large ones ran a single round.)
    Without a partition: 3000 x 3500 against the C oracle.  With one: 20 000 x 20 000 (a category's row range must be
    long enough for every one of its 64 slices to hold rows) against the fused kernel, which the oracle pins above."""
    from napkon_string_matching_amd import _lib, grid, synthetic, tables
    from oracle import native

    n_l, n_r = (20_000, 20_000) if partition else (3000, 3500)
    hap = synthetic.c5_cohort(n_l, 21)
    pop = synthetic.c5_cohort(n_r, 22, plant_from=hap, plant_fraction=0.05)
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY if partition else _lib.CAT_NONE
    li, ls, ri, rs = tables.encode_level_codes(synthetic.c5_level_codes(hap), synthetic.c5_level_codes(pop),
                                               len(synthetic.C5_ALPHABET), dev, hap["cat"] if partition else None,
                                               pop["cat"] if partition else None, mode)
    assert (li.seg is not None) == partition
    fused = grid.indel_levels_grid(li, ls, ri, rs, 0.7, category_mode=mode, park=True)
    assert len(fused) > 100
    if not partition:
        cps = lambda c: [[[ord(ch) for ch in " ".join(level)] for level in item] for item in synthetic.c5_level_token_lists(c)]
        _same_hits(fused, native.levels(True, cps(hap), cps(pop), 0.7, None, None, mode, cap=1 << 20))
    # expected survivors (the library's estimate): 2 % of the pairs of table ROWS (a partitioned table has one row per item
    # and category), 1/16 of that with a partition; configs[4]'s corpus measures 0.9 % of the pairs a partition visits
    expect = li.n * ri.n * 0.02 * (1 / 16 if partition else 1.0)
    lib = _lib.load()
    cm = li.category_mode if li.category_mode is not None else mode
    seen_rounds = set()
    for rounds_wanted in (3, 5, 8):
        entries = int(expect / (rounds_wanted - 0.5))
        nbytes = 512 + 16 * entries
        ws = torch.full((nbytes // 8,), -1, dtype=torch.int64, device=dev)
        buf = grid.HitBuffer(1 << 20, dev)
        buf.reset()
        _lib.check(lib.nsm_indel_levels_grid(li.struct(), ls.struct(), ri.struct(), rs.struct(), 0.7, int(cm), _lib.FLAG_PRUNE,
                                             buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(), ws.data_ptr(), nbytes, 0.0,
                                             torch.cuda.current_stream(dev).cuda_stream), "nsm_indel_levels_grid")
        n = int(buf.count.item())
        ctl = ws[:64].cpu().numpy()
        per_round = [int(v) for v in ctl[2:64] if v > 0]
        assert int(ctl[1]) & 0xFFFFFFFF == 0, ("the queue overflowed: the rounds were not what produced the hits", per_round, entries)
        assert len(per_round) == rounds_wanted and max(per_round) <= entries, (per_round, entries)
        seen_rounds.add(len(per_round))
        got = grid.sort_hits_device(buf, n)
        assert got.as_tuples() == fused.as_tuples(), (partition, rounds_wanted)
    assert max(seen_rounds) >= 8


def test_indel_levels_probe_and_route(dev):
    """The host MEASURES how many pairs outlive step 1 on a sample of the left rows (NSM_FLAG_PROBE: scan kernel only, the
    queue counters are the result, no hit is written) and routes by it: few survivors -> the split path at ANY threshold
    (NSM_FLAG_SPLIT, queue sized to the measurement), many -> the shared-tile kernel (NSM_FLAG_TILE).  Whatever the route,
    the hits are the fused kernel's (which the oracle pins elsewhere): word-like text at 0.55 (split), digit strings at
    0.6 (tile), both forced routes on one grid."""
    from napkon_string_matching_amd import _lib, grid, synthetic, tables

    lib = _lib.load()
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY
    lex = synthetic.word_vocabulary(5000)
    cases = {}
    for name, kw in (("words", dict(lex=lex)), ("digits", dict())):
        hap = synthetic.c5_cohort(9000, 31, **kw)
        pop = synthetic.c5_cohort(16000, 32, plant_from=hap, plant_fraction=0.02, **kw)
        cases[name] = tables.encode_level_codes(synthetic.c5_level_codes(hap), synthetic.c5_level_codes(pop),
                                                len(synthetic.c5_alphabet(hap)), dev, hap["cat"], pop["cat"], mode)
    for name, thr, want_path in (("words", 0.55, "split"), ("words", 0.7, "split"), ("digits", 0.6, "tile"), ("digits", 0.7, "split")):
        li, ls, ri, rs = cases[name]
        fused = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, park=True, capacity=1 << 22)
        assert len(fused) > 50
        # the probe leaves the caller's hit counter alone and reports a sensible rate
        expected, visited = grid.probe_survival(li, ls, ri, rs, thr, li.category_mode)
        assert visited > 1e6 and 0.0 <= expected <= visited
        route = []
        got = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 22, probe=True, route=route)
        assert route and route[0]["path"] == want_path, (name, thr, route)
        assert got.as_tuples() == fused.as_tuples(), (name, thr, route)
        # the measured expectation sizes the queue: no overflow on the split route
        if want_path == "split":
            flag = []
            again = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 22, probe=True, return_overflow=flag)
            assert flag == [0] and again.as_tuples() == fused.as_tuples()
    # both routes forced on ONE grid, at a threshold where the library by itself would take neither
    li, ls, ri, rs = cases["words"]
    fused = grid.indel_levels_grid(li, ls, ri, rs, 0.6, category_mode=mode, park=True, capacity=1 << 22).as_tuples()
    for extra in (_lib.FLAG_SPLIT, _lib.FLAG_TILE):
        buf = grid.HitBuffer(1 << 22, dev)
        buf.reset()
        ws = torch.empty((512 + 16 * (1 << 20)) // 8, dtype=torch.int64, device=dev)
        _lib.check(lib.nsm_indel_levels_grid(li.struct(), ls.struct(), ri.struct(), rs.struct(), 0.6, int(li.category_mode),
                                             _lib.FLAG_PRUNE | extra, buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(),
                                             ws.data_ptr(), ws.numel() * 8, 0.0, torch.cuda.current_stream(dev).cuda_stream),
                   "nsm_indel_levels_grid")
        n = int(buf.count.item())
        assert grid.sort_hits_device(buf, n).as_tuples() == fused, extra
    # NSM_FLAG_PROBE without NSM_FLAG_SPLIT / a workspace is an argument error, not a silent full run
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    rc = lib.nsm_indel_levels_grid(li.struct(), ls.struct(), ri.struct(), rs.struct(), 0.6, int(li.category_mode),
                                   _lib.FLAG_PRUNE | _lib.FLAG_PROBE, 0, 0, cnt.data_ptr(), 0, 0, 0.0,
                                   torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 10001


def test_indel_levels_split_path_under_graph_capture(dev):
    """The split path's queue lives in the caller's workspace; its finish kernels run on a library-owned side stream
    in eager mode.  Captured into a hipGraph the call stays on the capturing stream alone (no stream or event is created
    under capture): the replay must give the eager call's hits, with a workspace (split path, one stream) and without
    (single-kernel path)."""
    from napkon_string_matching_amd import _lib, grid, tables

    rng = random.Random(515)
    words = ["".join(rng.choice("abcdefgh") for _ in range(rng.randint(2, 5))) for _ in range(40)]

    def item():
        n = rng.randint(1, 5)
        toks = [rng.choice(words) for _ in range(n + 1)]
        return [" ".join(sorted(set(toks[: k + 2]))) for k in range(n)]

    left = [item() for _ in range(150)]
    right = [item() for _ in range(700)] + [list(left[k % 9]) for k in range(200)]
    li, ls, ri, rs = tables.encode_level_strings(left, right, dev, None, None, _lib.CAT_NONE)
    assert ls.stride == 64
    want = grid.indel_levels_grid(li, ls, ri, rs, 0.7, park=True).as_tuples()
    assert len(want) > 100
    lib = _lib.load()
    asked = int(lib.nsm_indel_levels_workspace_bytes(li.struct(), ls.struct(), ri.struct(), rs.struct(), 0.7, _lib.FLAG_PRUNE, 0.0))
    assert asked > 0

    def hits_of(buf):
        n = int(buf.count.item())
        rec = grid.sort_hits_device(buf, n)
        return rec.as_tuples()

    for with_ws in (True, False):
        ws = torch.empty(asked // 8, dtype=torch.int64, device=dev) if with_ws else None

        def launch(buf, stream):
            buf.count.zero_()
            _lib.check(lib.nsm_indel_levels_grid(li.struct(), ls.struct(), ri.struct(), rs.struct(), 0.7, _lib.CAT_NONE,
                                                 _lib.FLAG_PRUNE, buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(),
                                                 ws.data_ptr() if ws is not None else 0, asked if ws is not None else 0, 0.0,
                                                 stream.cuda_stream), "nsm_indel_levels_grid")

        stream = torch.cuda.Stream(dev)
        buf = grid.HitBuffer(1 << 14, dev)
        buf.reset()
        with torch.cuda.stream(stream):
            launch(buf, stream)  # eager (with a workspace: the side stream of this stream is created here)
            stream.synchronize()
            assert hits_of(buf) == want
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                launch(buf, stream)
            for _ in range(2):
                g.replay()
                torch.cuda.synchronize(dev)
                assert hits_of(buf) == want, with_ws
            if ws is not None:
                assert int(ws[1].item()) & 0xFFFFFFFF == 0
        lib.nsm_release(stream.cuda_stream)


@pytest.mark.parametrize("hi", [90, 230, 480])
def test_indel_levels_long_strings(dev, hi):
    from napkon_string_matching_amd import _lib, grid, tables
    from oracle import native

    rng = random.Random(1000 + hi)
    alphabet = "abcdefghij klm"

    def item():
        return ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, hi))).strip() for _ in range(rng.randint(1, 4))]

    left, right = [item() for _ in range(40)], [item() for _ in range(70)]
    right[3] = list(left[2])
    lcat = np.array([rng.choice([1, 2, 3]) for _ in left], dtype=np.uint64)
    rcat = np.array([rng.choice([0, 1, 2]) for _ in right], dtype=np.uint64)
    cps = lambda items: [[[ord(c) for c in s] for s in it] for it in items]
    for mode in (_lib.CAT_NONE, _lib.CAT_INTERSECT):
        li, ls, ri, rs = tables.encode_level_strings(left, right, dev, lcat, rcat, mode)
        assert ls.stride == rs.stride and ls.stride >= 128
        for thr in (0.0, 0.3, 0.6):
            want = native.levels(True, cps(left), cps(right), thr, lcat, rcat, mode, cap=1 << 14)
            got = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 10)
            _same_hits(got, want, FUZZY_TOL)
            _same_hits(got, want)
            _same_hits(grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, prune=False), want)
            _same_hits(grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, wave_wide=True), want)


def test_sort_hits_large(dev):
    """nsm_sort_hits beyond the single-workgroup limit (one radix sort over 128-bit keys built in place; the bitonic
    network on a capturing stream) against numpy's lexsort: with and without the caller's hints (n_hint: geometry from
    the live count instead of the capacity; id_limit: fewer key bits), ids with the sign bit set, many equal scores."""
    import torch

    from napkon_string_matching_amd import _lib, grid

    lib = _lib.load()
    for n, cap in ((1000, 1 << 10), (5000, 1 << 13), (8193, 1 << 14), (10_000, 1 << 20), (70_001, 70_001),
                   (200_000, 1 << 18), (300_001, 300_001), (1_500_000, 1 << 21)):
        rng = np.random.default_rng(n)
        score = rng.integers(0, 50, n).astype(np.float64) / 49.0
        if n == 70_001:
            score = rng.random(n)  # all different, incl. tiny values
            score[:5] = [0.0, 1.0, 5e-324, 1e-300, 0.5]
        i = rng.integers(0, 1000, n).astype(np.int32)
        j = np.arange(n, dtype=np.int32)
        rec = np.zeros((cap, 2), dtype=np.float64)
        rec[:n, 0] = score
        rec.view(np.int32).reshape(cap, 4)[:n, 2] = i
        rec.view(np.int32).reshape(cap, 4)[:n, 3] = j
        order = np.lexsort((j, i, -score))
        for id_limit in (0, max(n, 1000), 1 << 31):
            buf = grid.HitBuffer(cap, dev)
            buf.records.copy_(torch.from_numpy(rec))
            buf.count.fill_(n)
            got = grid.sort_hits_device(buf, n, id_limit)
            assert np.array_equal(got.score, score[order]), (n, cap, id_limit)
            assert np.array_equal(got.i, i[order]) and np.array_equal(got.j, j[order])
        # no hint at all: the geometry follows the capacity
        buf = grid.HitBuffer(cap, dev)
        buf.records.copy_(torch.from_numpy(rec))
        buf.count.fill_(n)
        buf.scratch = torch.empty_like(buf.records)
        _lib.check(lib.nsm_sort_hits(buf.records.data_ptr(), buf.scratch.data_ptr(), buf.capacity, buf.count.data_ptr(), 0, 0,
                                     torch.cuda.current_stream(dev).cuda_stream), "nsm_sort_hits")
        host = buf.records[:n].cpu().numpy()
        assert np.array_equal(host[:, 0], score[order]) and np.array_equal(host.view(np.int32).reshape(n, 4)[:, 2], i[order])
    # negative ids (the sign bit is part of the order when no id_limit is promised)
    n, cap = 20_000, 1 << 15
    rng = np.random.default_rng(7)
    score = rng.integers(0, 3, n).astype(np.float64) / 2.0
    i = rng.integers(-500, 500, n).astype(np.int32)
    j = rng.permutation(n).astype(np.int32) - 10_000
    rec = np.zeros((cap, 2), dtype=np.float64)
    rec[:n, 0] = score
    rec.view(np.int32).reshape(cap, 4)[:n, 2] = i
    rec.view(np.int32).reshape(cap, 4)[:n, 3] = j
    buf = grid.HitBuffer(cap, dev)
    buf.records.copy_(torch.from_numpy(rec))
    buf.count.fill_(n)
    got = grid.sort_hits_device(buf, n)
    order = np.lexsort((j, i, -score))
    assert np.array_equal(got.score, score[order]) and np.array_equal(got.i, i[order]) and np.array_equal(got.j, j[order])
    # on a capturing stream nothing may be allocated: the bitonic network, replayed
    n, cap = 50_000, 1 << 16
    score = rng.integers(0, 9, n).astype(np.float64) / 8.0
    i = rng.integers(0, 300, n).astype(np.int32)
    j = np.arange(n, dtype=np.int32)
    rec = np.zeros((cap, 2), dtype=np.float64)
    rec[:n, 0] = score
    rec.view(np.int32).reshape(cap, 4)[:n, 2] = i
    rec.view(np.int32).reshape(cap, 4)[:n, 3] = j
    buf = grid.HitBuffer(cap, dev)
    buf.scratch = torch.empty_like(buf.records)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        buf.records.copy_(torch.from_numpy(rec))
        buf.count.fill_(n)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            _lib.check(lib.nsm_sort_hits(buf.records.data_ptr(), buf.scratch.data_ptr(), buf.capacity, buf.count.data_ptr(), n, n,
                                         stream.cuda_stream), "nsm_sort_hits")
        g.replay()
        torch.cuda.synchronize(dev)
    host = buf.records[:n].cpu().numpy()
    order = np.lexsort((j, i, -score))
    assert np.array_equal(host[:, 0], score[order]) and np.array_equal(host.view(np.int32).reshape(n, 4)[:, 3], j[order])


@pytest.mark.parametrize("entries,words,n_left,n_right,stride", [
    ((2, 4), (1, 3), 150, 330, 128),    # 128-unit rows (K = 2): 16 waves per block
    ((3, 5), (2, 5), 170, 300, 256),    # the reference's Term shape (K = 4)
    ((3, 8), (2, 5), 120, 200, 256),    # items deeper than the resident images: levels read from global memory
    ((4, 8), (3, 6), 70, 130, 512),     # 512-unit rows (K = 8), deep items
])
def test_indel_levels_term_like(dev, entries, words, n_left, n_right, stride):
    """Multi-word level strings through the shared-tile kernel (round 3: all waves of a block share one right tile
    whose level strings stay resident in LDS): suffix-nested Term-shaped items, several tiles with a partial last one,
    thresholds from "everything survives every step" to "nearly nothing survives step 1", with and without a category
    partition -- against the C oracle (O(nm) DP per level pair), the round-2 park kernel and the wave-wide kernel."""
    from napkon_string_matching_amd import _lib, grid, synthetic, tables
    from napkon_string_matching_amd.compare import score_functions as sf
    from oracle import native

    rng = random.Random(entries[1] * 100 + words[1])
    lt_items = synthetic.term_cohort(n_left, 31, vocab=300, entries=entries, words=words)
    rt_items = synthetic.term_cohort(n_right, 32, vocab=300, entries=entries, words=words, plant_from=lt_items,
                                     plant_fraction=0.05)
    ops = lambda items: [[sf.fuzzy_operand(lv) for lv in it] for it in synthetic.term_levels(items)]
    left, right = ops(lt_items), ops(rt_items)
    left[3] = left[3][:1]      # single-level items: every step compares level 0
    right[5] = right[5][:1]
    right[7] = list(left[3])
    lcat = np.array([rng.choice([1, 2, 3, 4, 6]) for _ in left], dtype=np.uint64)
    rcat = np.array([rng.choice([1, 2, 3, 5, 0]) for _ in right], dtype=np.uint64)
    cps = lambda items: [[[ord(c) for c in s] for s in it] for it in items]
    every = native.levels(True, cps(left), cps(right), 0.0, None, None, 0, cap=n_left * n_right)
    assert len(every) == n_left * n_right
    # items WITHOUT levels, in a category of their own on both sides (types/comparable_data.py:255-258: two such items
    # score 0 -- a hit exactly when the threshold is <= 0; against an item with levels the reference raises IndexError,
    # which the host checks before the launch, so the category filter must keep them apart here).  The last left item is
    # one of them: its first level row points one past the string table.
    zl, zr = [n_left - 1, 11], [n_right - 1, 13]
    for k in zl:
        left[k], lcat[k] = [], np.uint64(1 << 40)
    for k in zr:
        right[k], rcat[k] = [], np.uint64(1 << 40)
    every = [(s_, i, j) for (s_, i, j) in every if i not in zl and j not in zr] + [(0.0, i, j) for i in zl for j in zr]
    every.sort(key=lambda h: (-h[0], h[1], h[2]))
    for mode, partition in ((_lib.CAT_NONE, True), (_lib.CAT_INTERSECT, True), (_lib.CAT_INTERSECT_OR_BOTH_EMPTY, False)):
        li, ls, ri, rs = tables.encode_level_strings(left, right, dev, lcat, rcat, mode, partition=partition)
        assert ls.stride == rs.stride == stride
        if mode == _lib.CAT_NONE:  # (no category filter: the zero-level items would meet items with levels)
            drop_l, drop_r = set(zl), set(zr)
            l2 = [it for k, it in enumerate(left) if k not in drop_l]
            r2 = [it for k, it in enumerate(right) if k not in drop_r]
            li, ls, ri, rs = tables.encode_level_strings(l2, r2, dev)
            lmap = [k for k in range(n_left) if k not in drop_l]
            rmap = [k for k in range(n_right) if k not in drop_r]
            for thr in (0.0, 0.45, 0.7):
                want = [(s_, i, j) for (s_, i, j) in every if s_ >= thr and i not in drop_l and j not in drop_r]
                got = [(s_, lmap[i], rmap[j]) for (s_, i, j) in grid.indel_levels_grid(li, ls, ri, rs, thr, capacity=1 << 17).as_tuples()]
                assert sorted(got, key=lambda h: (-h[0], h[1], h[2])) == want
            continue
        if mode == _lib.CAT_NONE:
            keep = lambda i, j: True
        elif mode == _lib.CAT_INTERSECT:
            keep = lambda i, j: bool(int(lcat[i]) & int(rcat[j]))
        else:
            keep = lambda i, j: bool(int(lcat[i]) & int(rcat[j])) or (not lcat[i] and not rcat[j])
        for thr in (0.0, 0.2, 0.45, 0.7, 0.9):
            want = [(s, i, j) for (s, i, j) in every if s >= thr and keep(i, j)]
            got = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 17)
            _same_hits(got, want)
            assert thr > 0.8 or len(want) > 0
            if thr in (0.2, 0.7):
                _same_hits(grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, capacity=1 << 17, prune=False), want)
    # the round-1 / round-2 kernels behind NSM_FLAG_WAVE_WIDE / NSM_FLAG_PARK do not take items without levels
    # (include/nsm_hip.h; the Python host never passes them down): compared on the grid without them
    keep_l = [k for k in range(n_left) if k not in zl]
    keep_r = [k for k in range(n_right) if k not in zr]
    li, ls, ri, rs = tables.encode_level_strings([left[k] for k in keep_l], [right[k] for k in keep_r], dev,
                                                 lcat[keep_l], rcat[keep_r], _lib.CAT_INTERSECT)
    for thr in (0.2, 0.7):
        tile = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=_lib.CAT_INTERSECT, capacity=1 << 17)
        park = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=_lib.CAT_INTERSECT, capacity=1 << 17, park=True)
        wide = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=_lib.CAT_INTERSECT, capacity=1 << 17, wave_wide=True)
        assert tile.as_tuples() == park.as_tuples() == wide.as_tuples() and (thr > 0.5 or len(tile) > 0)


@pytest.mark.parametrize("vocab", [20_000, 1 << 17])
def test_jaccard_levels_c5_shaped_low_threshold(dev, vocab):
    """configs[4]-shaped cohorts (4 levels, ~8 ids, 1-2 of 32 categories, category partition) at the API's default
    threshold 0.1 (types/comparable_data.py:75), 25k x 25k: the inverted-index kernel (chosen by the library below
    0.45), the signature-filter kernel and the C oracle agree bit for bit."""
    from napkon_string_matching_amd import _lib, grid, synthetic, tables
    from oracle import native

    n = 25_000
    hap = synthetic.c5_cohort(n, 41, vocab=vocab)
    pop = synthetic.c5_cohort(n, 42, vocab=vocab, plant_from=hap)
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY
    hap["cat"][:40] = 0
    pop["cat"][:60] = 0
    lt = tables.SetTable.from_nested_arrays(hap["ids"], hap["plen"], hap["nlev"], "left", dev, categories=hap["cat"], width=16,
                                            category_mode=mode)
    rt = tables.SetTable.from_nested_arrays(pop["ids"], pop["plen"], pop["nlev"], "right", dev, categories=pop["cat"], width=16,
                                            category_mode=mode)
    assert lt.seg is not None
    auto = grid.jaccard_levels_grid(lt, rt, 0.1, category_mode=mode, capacity=1 << 18)
    matrix = grid.jaccard_levels_grid(lt, rt, 0.1, category_mode=mode, capacity=1 << 18, index=False)
    forced = grid.jaccard_levels_grid(lt, rt, 0.1, category_mode=mode, capacity=1 << 18, index=True)      # global index
    tile = grid.jaccard_levels_grid(lt, rt, 0.1, category_mode=mode, capacity=1 << 18, index="tile")       # per-tile index
    assert rt.post is not None
    assert auto.as_tuples() == matrix.as_tuples() == forced.as_tuples() == tile.as_tuples() and len(auto) > 200
    ids = lambda c: [[[int(t[1:]) for t in level] for level in item] for item in synthetic.c5_level_token_lists(c)]
    want = native.levels(False, ids(hap), ids(pop), 0.1, hap["cat"], pop["cat"], mode, cap=1 << 18)
    _same_hits(auto, want)
    # without a partition (per-lane predicate) and without categories
    for m, part in ((mode, False), (_lib.CAT_NONE, False)):
        lt2 = tables.SetTable.from_nested_arrays(hap["ids"][:6000], hap["plen"][:6000], hap["nlev"][:6000], "left", dev,
                                                 categories=hap["cat"][:6000], width=16, category_mode=m, partition=part)
        rt2 = tables.SetTable.from_nested_arrays(pop["ids"][:7000], pop["plen"][:7000], pop["nlev"][:7000], "right", dev,
                                                 categories=pop["cat"][:7000], width=16, category_mode=m, partition=part)
        a = grid.jaccard_levels_grid(lt2, rt2, 0.1, category_mode=m, capacity=1 << 18, index=True)
        b = grid.jaccard_levels_grid(lt2, rt2, 0.1, category_mode=m, capacity=1 << 18, index=False)
        c = grid.jaccard_levels_grid(lt2, rt2, 0.1, category_mode=m, capacity=1 << 18, index="tile")
        assert a.as_tuples() == b.as_tuples() == c.as_tuples() and len(a) > 0
