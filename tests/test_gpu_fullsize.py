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
