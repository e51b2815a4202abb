"""BASELINE.json full-size grids checked through size-independent properties (the oracle cannot
score 2.5e9 .. 4e10 pairs): prune == no prune, symmetry under swapping the sides, threshold
monotonicity, recall of the planted near-duplicates, exactness of the scores of the hits (each hit is
re-scored by the oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    return torch.device("cuda:0")


def _tuples(h):
    return h.as_tuples()


def test_c2_full_size_properties(dev):
    """configs[1]: 50k x 50k token-id sets, intersection_vs_union, threshold 0.5."""
    from napkon_string_matching_amd import grid, synthetic, tables
    from oracle import score_functions as osf

    left, right = synthetic.c2_corpus()  # 50 000 x 50 000, 1 % planted
    lt, rt = tables.SetTable.from_padded(left, "left", dev), tables.SetTable.from_padded(right, "right", dev)
    pruned = _tuples(grid.jaccard_raw_grid(lt, rt, 0.5, prune=True))
    exhaustive = _tuples(grid.jaccard_raw_grid(lt, rt, 0.5, prune=False))
    assert pruned == exhaustive and len(pruned) >= 400
    # every reported score is the reference's arithmetic on that pair
    for s, i, j in pruned:
        a = [str(v) for v in left[i] if v >= 0]
        b = [str(v) for v in right[j] if v >= 0]
        assert s == osf.intersection_vs_union(a, b) and s >= 0.5
    # symmetry: J(A, B) == J(B, A)
    lt2, rt2 = tables.SetTable.from_padded(right, "left", dev), tables.SetTable.from_padded(left, "right", dev)
    swapped = {(j, i): s for s, i, j in _tuples(grid.jaccard_raw_grid(lt2, rt2, 0.5))}
    assert swapped == {(i, j): s for s, i, j in pruned}
    # monotone in the threshold
    high = _tuples(grid.jaccard_raw_grid(lt, rt, 0.8))
    assert set(high) == {h for h in pruned if h[0] >= 0.8} and 0 < len(high) < len(pruned)
    # recall: a right row equal to a left row must be reported with score 1.0
    index = {tuple(r): k for k, r in enumerate(map(tuple, left))}
    exact = [(index[tuple(r)], j) for j, r in enumerate(map(tuple, right)) if tuple(r) in index]
    assert len(exact) >= 100
    found = {(i, j) for s, i, j in pruned if s == 1.0}
    for i, j in exact:
        assert (i, j) in found or any(s == 1.0 and jj == j for s, _, jj in pruned)


def test_c3_full_size_properties(dev):
    """configs[2]: 200k x 200k strings, fuzzy_match, threshold 0.8."""
    from napkon_string_matching_amd import grid, synthetic, tables
    from oracle import score_functions as osf

    (lc, ll), (rc, rl) = synthetic.c3_corpus()
    alpha = len(synthetic.STRING_ALPHABET)
    lt, rt = tables.StrTable.from_codes(lc, ll, alpha, dev), tables.StrTable.from_codes(rc, rl, alpha, dev)
    pruned = _tuples(grid.indel_raw_grid(lt, rt, 0.8, prune=True))
    exhaustive = _tuples(grid.indel_raw_grid(lt, rt, 0.8, prune=False))
    assert pruned == exhaustive and len(pruned) >= 1500
    sample = pruned[:: max(1, len(pruned) // 300)]
    ls = synthetic.decode_strings(lc[[i for _, i, _ in sample]], ll[[i for _, i, _ in sample]])
    rs = synthetic.decode_strings(rc[[j for _, _, j in sample]], rl[[j for _, _, j in sample]])
    for (s, _, _), a, b in zip(sample, ls, rs):
        assert abs(s - osf.fuzzy_match(a, b)) <= 1e-6 and s >= 0.8
    # symmetry of the Indel ratio
    lt2, rt2 = tables.StrTable.from_codes(rc, rl, alpha, dev), tables.StrTable.from_codes(lc, ll, alpha, dev)
    swapped = {(j, i): s for s, i, j in _tuples(grid.indel_raw_grid(lt2, rt2, 0.8))}
    assert swapped == {(i, j): s for s, i, j in pruned}
    high = _tuples(grid.indel_raw_grid(lt, rt, 0.95))
    assert set(high) == {h for h in pruned if h[0] >= 0.95}


def test_c5_shaped_levels_properties(dev):
    """configs[4]-shaped cohorts (60k items, 4 levels, categories): self-grid diagonal, planted pairs,
    agreement between the category-partitioned and the per-lane-predicate fuzzy grids."""
    from napkon_string_matching_amd import _lib, grid, synthetic, tables
    from napkon_string_matching_amd.compare import score_functions as sf

    n = 60_000
    hap = synthetic.c5_cohort(n, 11)
    pop = synthetic.c5_cohort(n, 12, plant_from=hap)
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY
    mk = lambda c, side: tables.SetTable.from_nested_arrays(c["ids"], c["plen"], c["nlev"], side, dev,
                                                             categories=c["cat"], width=16, category_mode=mode)
    # self grid: every item matches itself with the maximal score 1 - 2^-4
    self_hits = grid.jaccard_levels_grid(mk(hap, "left"), mk(hap, "right"), 0.9, category_mode=mode)
    diag = {(i, j) for s, i, j in self_hits.as_tuples() if i == j}
    assert len(diag) == n and all(s == 0.9375 for s, i, j in self_hits.as_tuples() if i == j)
    hits = grid.jaccard_levels_grid(mk(hap, "left"), mk(pop, "right"), 0.7, category_mode=mode).as_tuples()
    same = {(i, j) for s, i, j in hits if s == 0.9375}
    tok_index = {tuple(r): k for k, r in enumerate(map(tuple, hap["tok"]))}
    planted = [(tok_index[tuple(r)], j) for j, r in enumerate(map(tuple, pop["tok"])) if tuple(r) in tok_index]
    assert len(planted) >= 200 and all(p in same for p in planted)
    # the category-partitioned grid and the per-lane predicate agree
    mk2 = lambda c, side: tables.SetTable.from_nested_arrays(c["ids"], c["plen"], c["nlev"], side, dev,
                                                              categories=c["cat"], width=16, category_mode=mode,
                                                              partition=False)
    assert grid.jaccard_levels_grid(mk2(hap, "left"), mk2(pop, "right"), 0.7, category_mode=mode).as_tuples() == hits
    # fuzzy levels on a 6k corner: partitioned == unpartitioned
    m = 6000
    lv = lambda c: [[sf.fuzzy_operand(x) for x in it] for it in synthetic.c5_level_token_lists(c, slice(0, m))]
    la, lb = lv(hap), lv(pop)
    res = []
    for part in (True, False):
        li, ls, ri, rs = tables.encode_level_strings(la, lb, dev, hap["cat"][:m], pop["cat"][:m], mode, partition=part)
        res.append(grid.indel_levels_grid(li, ls, ri, rs, 0.7, category_mode=mode).as_tuples())
    assert res[0] == res[1] and len(res[0]) > 0


def test_c4_full_size_sharded_equals_whole(dev):
    """configs[3]: 1M x 1M token-id sets, threshold 0.8.  The eight left-row blocks of the 8-GPU
    layout, run one after the other here, reproduce the one-grid result; prune == no prune."""
    from napkon_string_matching_amd import distributed, grid, synthetic, tables
    from oracle import score_functions as osf

    n = 1_000_000
    left, right = synthetic.c2_corpus(n, n)
    rt = tables.SetTable.from_padded(right, "right", dev)
    whole = _tuples(grid.jaccard_raw_grid(tables.SetTable.from_padded(left, "left", dev), rt, 0.8))
    assert len(whole) >= 2000
    parts = []
    for rank in range(8):
        lo, hi = distributed.shard_bounds(n, rank, 8)
        block = tables.SetTable.from_padded(left[lo:hi], "left", dev, orig=np.arange(lo, hi, dtype=np.int32))
        got = grid.jaccard_raw_grid(block, rt, 0.8)
        assert all(lo <= i < hi for i in got.i.tolist())
        parts += _tuples(got)
        if rank == 3:  # one block also exhaustively (1.25e11 pairs)
            assert _tuples(grid.jaccard_raw_grid(block, rt, 0.8, prune=False)) == _tuples(got)
    assert sorted(parts, key=lambda h: (-h[0], h[1], h[2])) == whole
    for s, i, j in whole[:: max(1, len(whole) // 500)]:
        a = [str(v) for v in left[i] if v >= 0]
        b = [str(v) for v in right[j] if v >= 0]
        assert s == osf.intersection_vs_union(a, b) and s >= 0.8


def test_c5_full_size_cohort_pair(dev):
    """configs[4]: one 500k x 500k cohort pair, 4 levels, 32 categories, filter_categories, Jaccard
    levels at the config's score threshold 0.7: planted items found with the maximal score, the
    category partition agrees with the per-lane predicate, hits are monotone in the threshold."""
    from napkon_string_matching_amd import _lib, grid, synthetic, tables
    from oracle import compare as ocmp
    from oracle import score_functions as osf

    n = 500_000
    hap = synthetic.c5_cohort(n, 21)
    pop = synthetic.c5_cohort(n, 22, plant_from=hap)
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY

    def mk(c, side, partition=True):
        return tables.SetTable.from_nested_arrays(c["ids"], c["plen"], c["nlev"], side, dev, categories=c["cat"],
                                                  width=16, category_mode=mode, partition=partition)

    hits = grid.jaccard_levels_grid(mk(hap, "left"), mk(pop, "right"), 0.7, category_mode=mode).as_tuples()
    plain = grid.jaccard_levels_grid(mk(hap, "left", False), mk(pop, "right", False), 0.7, category_mode=mode)
    assert plain.as_tuples() == hits and len(hits) >= 2000
    cached = grid.jaccard_levels_grid(mk(hap, "left"), mk(pop, "right"), 0.5, category_mode=mode).as_tuples()
    assert [h for h in cached if h[0] >= 0.7] == hits  # cache_threshold 0.5 then score_threshold 0.7
    tok_index = {tuple(r): k for k, r in enumerate(map(tuple, hap["tok"]))}
    planted = [(tok_index[tuple(r)], j) for j, r in enumerate(map(tuple, pop["tok"])) if tuple(r) in tok_index]
    same = {(i, j) for s, i, j in hits if s == 0.9375}
    cats_ok = lambda i, j: bool(hap["cat"][i] & pop["cat"][j]) or (hap["cat"][i] == 0 and pop["cat"][j] == 0)
    assert len(planted) >= 2000 and all(p in same for p in planted if cats_ok(*p))
    # a sample of hits re-scored by the oracle's compare_terms on the decoded level lists
    sample = hits[:: max(1, len(hits) // 200)]
    for s, i, j in sample:
        a = synthetic.c5_level_token_lists(hap, slice(i, i + 1))[0]
        b = synthetic.c5_level_token_lists(pop, slice(j, j + 1))[0]
        assert s == ocmp.compare_terms(a, b, osf.intersection_vs_union) and cats_ok(i, j)


def test_c5_full_size_fuzzy_pair(dev):
    """configs[4], the fuzzy_match leg at full size: one 500k x 500k cohort pair, 4 levels, 32 categories,
    filter_categories, compare_terms x fuzzy_match through nsm_indel_levels_grid at the config's score
    threshold 0.7 (reference: types/comparable_data.py:223-232, :248-265).  Size-independent properties:
    planted copies found with the maximal score 1 - 2^-4; the hits of a lower device threshold filtered to
    0.7 are the 0.7 run (the cache_threshold / score_threshold relation of compare(), :123-126 -- 0.65 here:
    at the config's 0.5 half of all same-category pairs of this corpus are hits, 10^10 records); sampled hits
    re-scored by the oracle's compare_terms; a 50k corner equals the un-partitioned grid."""
    from napkon_string_matching_amd import _lib, grid, synthetic, tables
    from napkon_string_matching_amd.compare import score_functions as sf
    from oracle import compare as ocmp
    from oracle import score_functions as osf

    n = 500_000
    hap = synthetic.c5_cohort(n, 21)
    pop = synthetic.c5_cohort(n, 22, plant_from=hap)
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY
    lv = lambda c, rows=slice(None): [[sf.fuzzy_operand(x) for x in it] for it in synthetic.c5_level_token_lists(c, rows)]
    la, lb = lv(hap), lv(pop)
    li, ls, ri, rs = tables.encode_level_strings(la, lb, dev, hap["cat"], pop["cat"], mode)
    assert li.seg is not None and ls.stride == 64
    hits = grid.indel_levels_grid(li, ls, ri, rs, 0.7, category_mode=mode, capacity=1 << 16).as_tuples()
    assert len(hits) >= 2000 and all(s >= 0.7 for s, _, _ in hits)
    cats_ok = lambda i, j: bool(hap["cat"][i] & pop["cat"][j]) or (hap["cat"][i] == 0 and pop["cat"][j] == 0)
    # (i) planted copies: identical items score 1 - 2^-4 = 0.9375 exactly
    tok_index = {tuple(r): k for k, r in enumerate(map(tuple, hap["tok"]))}
    planted = [(tok_index[tuple(r)], j) for j, r in enumerate(map(tuple, pop["tok"])) if tuple(r) in tok_index]
    same = {(i, j) for s, i, j in hits if s == 0.9375}
    assert len(planted) >= 2000 and all(pq in same for pq in planted if cats_ok(*pq))
    # (ii) lower device threshold, filtered on the host
    low = grid.indel_levels_grid(li, ls, ri, rs, 0.65, category_mode=mode, capacity=1 << 22).as_tuples()
    assert len(low) > len(hits) and [h for h in low if h[0] >= 0.7] == hits
    # (iii) sampled hits (both runs) against the oracle's compare_terms x fuzzy_match
    sample = hits[:: max(1, len(hits) // 150)] + low[:: max(1, len(low) // 100)]
    assert len(sample) >= 200
    for s, i, j in sample:
        a = synthetic.c5_level_token_lists(hap, slice(i, i + 1))[0]
        b = synthetic.c5_level_token_lists(pop, slice(j, j + 1))[0]
        want = ocmp.compare_terms(a, b, osf.fuzzy_match)
        assert abs(s - want) <= 1e-6 and s == want and cats_ok(i, j)
    # (iv) a 50k x 50k corner: partitioned == per-lane predicate == wave-wide kernel
    m = 50_000
    res = []
    for part in (True, False):
        ci, cs, di, ds = tables.encode_level_strings(la[:m], lb[:m], dev, hap["cat"][:m], pop["cat"][:m], mode, partition=part)
        res.append(grid.indel_levels_grid(ci, cs, di, ds, 0.7, category_mode=mode).as_tuples())
        if part:
            res.append(grid.indel_levels_grid(ci, cs, di, ds, 0.7, category_mode=mode, wave_wide=True).as_tuples())
    assert res[0] == res[1] == res[2] and len(res[0]) > 10
    assert res[0] == [h for h in hits if h[1] < m and h[2] < m]


def test_fuzzy_levels_cooperative_equals_wave_wide(dev):
    """The block-cooperative late steps of the one-word fuzzy levels kernel against the wave-wide kernel it
    replaces (NSM_FLAG_WAVE_WIDE), 25k x 25k C5-shaped items, three thresholds, with and without partition."""
    from napkon_string_matching_amd import _lib, grid, synthetic, tables
    from napkon_string_matching_amd.compare import score_functions as sf

    n = 25_000
    hap = synthetic.c5_cohort(n, 31)
    pop = synthetic.c5_cohort(n, 32, plant_from=hap)
    lv = lambda c: [[sf.fuzzy_operand(x) for x in it] for it in synthetic.c5_level_token_lists(c)]
    la, lb = lv(hap), lv(pop)
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY
    for part in (True, False):
        li, ls, ri, rs = tables.encode_level_strings(la, lb, dev, hap["cat"], pop["cat"], mode, partition=part)
        for thr in (0.5, 0.7, 0.9):
            coop = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode).as_tuples()
            plain = grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, wave_wide=True).as_tuples()
            assert coop == plain and len(coop) > 100
