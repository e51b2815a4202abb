"""Device-side table builders (nsm_build_set_table / nsm_build_str_table / nsm_build_level_items, C ABI) against
the numpy encoders of tables.py they replace: every column byte-equal on random tables, RAW and levels mode,
both category predicates, with and without the partition; unsorted input comes out sorted; data errors are
reported, not swallowed."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    return torch.device("cuda:0")


def _same(a, b, what):
    if a is None or b is None:
        assert a is None and b is None, what
        return
    x, y = a.cpu().numpy(), b.cpu().numpy()
    assert x.shape == y.shape, (what, x.shape, y.shape)
    assert x.tobytes() == y.tobytes(), what


def _same_set_tables(g, c):
    assert (g.n, g.width, g.max_levels, g.has_empty, g.side, g.category_mode) == (c.n, c.width, c.max_levels, c.has_empty, c.side, c.category_mode)
    for col in ("ids", "cnt", "sig", "sig2", "orig", "size_start", "nlev", "plen", "cat", "filt", "seg", "seg_start", "post",
                "post_start"):
        _same(getattr(g, col), getattr(c, col), col)
    assert (g.vocab, tuple(g.post_sq), g.post_row_bits, g.post_format) == (c.vocab, tuple(c.post_sq), c.post_row_bits, c.post_format)


def _rand_ids(rng, n, width, vocab, kmax, allow_empty):
    ids = np.full((n, width), -1, dtype=np.int32)
    for r in range(n):
        k = rng.randint(0 if allow_empty else 1, kmax)
        ids[r, :k] = rng.sample(range(vocab), k)
    return ids


@pytest.mark.parametrize("width,kmax,vocab", [(16, 16, 60), (32, 30, 500), (64, 64, 100000)])
@pytest.mark.parametrize("side", ["left", "right"])
def test_set_table_raw(dev, width, kmax, vocab, side):
    from napkon_string_matching_amd import tables

    rng = random.Random(width * 7 + len(side))
    ids = _rand_ids(rng, 3001, width, vocab, kmax, allow_empty=True)
    orig = np.arange(3001, dtype=np.int32)[::-1] * 3
    for o in (None, orig):
        _same_set_tables(tables.SetTable.from_padded(ids, side, dev, width=width, orig=o),
                         tables.SetTable.from_padded(ids, side, "cpu", width=width, orig=o))
    empty = np.full((0, width), -1, dtype=np.int32)
    _same_set_tables(tables.SetTable.from_padded(empty, side, dev, width=width), tables.SetTable.from_padded(empty, side, "cpu", width=width))
    # the global inverted index (built for right tables by default; here on both sides): rows come out ascending, postings
    # sorted by (id, position), five offsets per id, the squared-length statistics on the host
    g = tables.SetTable.from_padded(ids, side, dev, width=width, index=True)
    c = tables.SetTable.from_padded(ids, side, "cpu", width=width, index=True)
    assert g.post is not None and g.vocab == int(ids.max()) + 1 and g.post_sq[4] > 0
    assert g.post_row_bits == c.post_row_bits == 32 - 2 * (width.bit_length() - 1) and g.post_format == 2  # (RAW: entry + signature fold)
    _same_set_tables(g, c)
    tables.COMPACT_POSTINGS = False  # the 64-bit entries of tables with more rows than a 32-bit entry can name
    try:
        g64 = tables.SetTable.from_padded(ids, side, dev, width=width, index=True)
        c64 = tables.SetTable.from_padded(ids, side, "cpu", width=width, index=True)
    finally:
        tables.COMPACT_POSTINGS = True
    assert g64.post_row_bits == 0 and g64.post_format == 0
    _same_set_tables(g64, c64)
    _same(g64.post_start, g.post_start, "post_start of both entry formats")
    rows = g.ids.cpu().numpy()
    live = rows >= 0
    assert (np.diff(np.where(live, rows, np.iinfo(np.int32).max).astype(np.int64), axis=1) >= 0).all()
    assert (tables.SetTable.from_padded(ids, "right", dev, width=width).post is not None) and \
           (tables.SetTable.from_padded(ids, "left", dev, width=width).post is None)


@pytest.mark.parametrize("n_cat", [3, 32, 63])
def test_set_table_levels(dev, n_cat):
    from napkon_string_matching_amd import _lib, synthetic, tables

    c = synthetic.c5_cohort(5000, 7 + n_cat, vocab=300, n_categories=n_cat)
    cat = c["cat"].copy()
    cat[::17] = 0  # items without a category
    nlev = c["nlev"].copy()
    nlev[::5] = 2  # shallower items
    for mode in (_lib.CAT_NONE, _lib.CAT_INTERSECT, _lib.CAT_INTERSECT_OR_BOTH_EMPTY):
        for part in (True, False):
            kw = dict(categories=cat, width=16, category_mode=mode, partition=part, orig=np.arange(5000, dtype=np.int32) + 11)
            _same_set_tables(tables.SetTable.from_nested_arrays(c["ids"], c["plen"], nlev, "left", dev, **kw),
                             tables.SetTable.from_nested_arrays(c["ids"], c["plen"], nlev, "left", "cpu", **kw))
            # right tables carry the global inverted index: postings by (category segment, id) with a partition
            g = tables.SetTable.from_nested_arrays(c["ids"], c["plen"], nlev, "right", dev, **kw)
            assert g.post is not None and g.vocab == int(c["ids"].max()) + 1
            assert g.post_start.shape[0] == 5 * g.vocab * (64 if g.seg is not None else 1) + 1
            _same_set_tables(g, tables.SetTable.from_nested_arrays(c["ids"], c["plen"], nlev, "right", "cpu", **kw))
    with pytest.raises(ValueError):  # all 64 bits are real categories: no room for the "both empty" category
        bad = cat.copy()
        bad[3] |= np.uint64(1) << np.uint64(63)
        tables.SetTable.from_nested_arrays(c["ids"], c["plen"], nlev, "left", dev, categories=bad, width=16,
                                           category_mode=_lib.CAT_INTERSECT_OR_BOTH_EMPTY)


@pytest.mark.parametrize("stride", [64, 128, 512])
@pytest.mark.parametrize("sort", [True, False])
def test_str_table(dev, stride, sort):
    from napkon_string_matching_amd import tables

    rng = np.random.default_rng(stride)
    n, alphabet = 4001, 37
    lengths = rng.integers(0, stride + 1, size=n).astype(np.int32)
    codes = rng.integers(0, alphabet, size=(n, stride)).astype(np.uint8)  # garbage past the length must not matter
    if stride == 512:
        codes[:50, :] = 5  # histogram counts saturate at 255
        lengths[:50] = 512
    g = tables.StrTable.from_codes(codes, lengths, alphabet, dev, sort=sort)
    c = tables.StrTable.from_codes(codes, lengths, alphabet, "cpu", sort=sort)
    assert (g.n, g.stride, g.alphabet, g.has_empty) == (c.n, c.stride, c.alphabet, c.has_empty)
    for col in ("codes", "len", "orig", "hist", "hist16", "len_start"):
        _same(getattr(g, col), getattr(c, col), col)
    assert (c.hist16 is not None) == (sort and stride == 64)  # the 16-bucket column: RAW tables of 64-unit strings
    bad = codes.copy()
    bad[7, 0] = 200
    lengths[7] = 3
    with pytest.raises(ValueError):
        tables.StrTable.from_codes(bad, lengths, alphabet, dev, sort=sort)


def test_level_items(dev):
    from napkon_string_matching_amd import _lib, tables

    rng = random.Random(99)
    alphabet = "abcdefghij klm"
    item = lambda: ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 30))).strip() for _ in range(rng.randint(1, 6))]
    left, right = [item() for _ in range(700)], [item() for _ in range(900)]
    lcat = np.array([rng.choice([0, 1, 2, 3, 5, 1 << 40]) for _ in left], dtype=np.uint64)
    rcat = np.array([rng.choice([0, 1, 2, 6]) for _ in right], dtype=np.uint64)
    for mode in (_lib.CAT_NONE, _lib.CAT_INTERSECT, _lib.CAT_INTERSECT_OR_BOTH_EMPTY):
        for part in (True, False):
            g = tables.encode_level_strings(left, right, dev, lcat, rcat, mode, partition=part, left_offset=5)
            c = tables.encode_level_strings(left, right, "cpu", lcat, rcat, mode, partition=part, left_offset=5)
            for gi, ci in ((g[0], c[0]), (g[2], c[2])):
                assert (gi.n, gi.category_mode) == (ci.n, ci.category_mode)
                for col in ("first", "nlev", "orig", "cat", "seg", "seg_start"):
                    _same(getattr(gi, col), getattr(ci, col), col)
            for gs, cs in ((g[1], c[1]), (g[3], c[3])):
                for col in ("codes", "len", "hist"):
                    _same(getattr(gs, col), getattr(cs, col), col)


def test_builder_rejects_bad_rows(dev):
    """Data errors surface as NSM_E_BADARG with a message (through ctypes: the raw C ABI)."""
    import ctypes

    import torch

    from napkon_string_matching_amd import _lib

    lib = _lib.load()
    ids = torch.tensor([[1, 2, 2, -1], [3, -1, -1, -1]], dtype=torch.int32, device=dev)  # duplicate id in row 0
    n, width = 2, 16
    cols = dict(ids=torch.empty((n, width), dtype=torch.int32, device=dev), cnt=torch.empty(n, dtype=torch.int32, device=dev),
                sig=torch.empty(n, dtype=torch.int64, device=dev), sig2=torch.empty(n, dtype=torch.int64, device=dev),
                orig=torch.empty(n, dtype=torch.int32, device=dev), size_start=torch.empty(width + 2, dtype=torch.int32, device=dev))
    st = _lib.NsmSetTable(cols["ids"].data_ptr(), cols["cnt"].data_ptr(), cols["sig"].data_ptr(), cols["sig2"].data_ptr(),
                          cols["orig"].data_ptr(), cols["size_start"].data_ptr(), None, None, None, None, None, None, n, width, 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    # RAW rows are SETS (the reference builds set(...) per pair): a repeated id counts once, flag or no flag
    for flags in (_lib.BUILD_VALIDATE, 0):
        st.n = n
        rc = lib.nsm_build_set_table(ids.data_ptr(), n, 4, 0, None, None, None, None, 0, flags, ctypes.byref(st), stream)
        assert rc == 0 and st.n == 2
        assert cols["cnt"].cpu().tolist() == [2, 1] and cols["orig"].cpu().tolist() == [0, 1]  # sorted by size, descending
        assert cols["ids"].cpu()[0, :3].tolist() == [1, 2, -1]
    # a levels row cannot drop a repeat (its prefix lengths count the caller's slots): data error, always
    lev = dict(nlev=torch.ones(n, dtype=torch.int32, device=dev), plen=torch.tensor([[3], [1]], dtype=torch.uint8, device=dev),
               nlev_o=torch.empty(n, dtype=torch.int32, device=dev), plen_o=torch.empty((n, 1), dtype=torch.uint8, device=dev),
               filt=torch.empty((n, 8), dtype=torch.int32, device=dev))
    st2 = _lib.NsmSetTable(cols["ids"].data_ptr(), cols["cnt"].data_ptr(), cols["sig"].data_ptr(), cols["sig2"].data_ptr(),
                           cols["orig"].data_ptr(), cols["size_start"].data_ptr(), lev["nlev_o"].data_ptr(), lev["plen_o"].data_ptr(),
                           None, lev["filt"].data_ptr(), None, None, n, width, 1)
    rc = lib.nsm_build_set_table(ids.data_ptr(), n, 4, 0, lev["nlev"].data_ptr(), lev["plen"].data_ptr(), None, None, 0, 0,
                                 ctypes.byref(st2), stream)
    assert rc == 10001 and b"duplicate id" in lib.nsm_last_error()
    st.n = 1  # output columns too short
    rc = lib.nsm_build_set_table(ids.data_ptr(), n, 4, 0, None, None, None, None, 0, 0, ctypes.byref(st), stream)
    assert rc == 10001 and b"rows" in lib.nsm_last_error()
