"""CPU-only checks (run with -m "not gpu"): the C ABI library loads and exports every symbol that
include/nsm_hip.h declares, the host encoders / containers behave, the product refuses to score
without a GPU, and the N > 1 exchange step works with gloo at world_size 2."""
import ctypes
import os
import re
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_library_exports_every_declared_symbol():
    from napkon_string_matching_amd import _lib

    header = (ROOT / "include" / "nsm_hip.h").read_text()
    declared = set(re.findall(r"^(?:int|uint64_t|const char\*)\s+(nsm_[a-z_]+)\s*\(", header, re.M))
    assert declared == set(_lib.EXPORTS)
    if not _lib.LIB_PATH.exists():
        pytest.skip("libnsm_hip.so not built (run __graft_entry__.build())")
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.nsm_abi_version() == _lib.ABI_VERSION
    # struct layouts must match the header (sizes on LP64)
    assert ctypes.sizeof(_lib.NsmHit) == 16
    assert ctypes.sizeof(_lib.NsmSetTable) == 12 * 8 + 4 * 4 + 2 * 8 + 5 * 8 + 8  # (ABI 5: post_row_bits + padding)
    assert ctypes.sizeof(_lib.NsmStrTable) == 5 * 8 + 3 * 4 + 4 + 8  # (ABI 5: hist16)
    assert ctypes.sizeof(_lib.NsmLevelItems) == 6 * 8 + 4 + 4


def test_argument_validation_without_gpu():
    """Bad arguments are rejected by the launcher before anything touches a device."""
    from napkon_string_matching_amd import _lib

    if not _lib.LIB_PATH.exists():
        pytest.skip("libnsm_hip.so not built")
    lib = _lib.load()
    a = _lib.NsmSetTable(None, None, None, None, None, None, None, None, None, None, None, None, 3, 16, 0)
    b = _lib.NsmSetTable(None, None, None, None, None, None, None, None, None, None, None, None, 3, 32, 0)
    cnt = ctypes.c_ulonglong(0)
    rc = lib.nsm_jaccard_raw_grid(a, b, 0.5, 0, None, 0, ctypes.addressof(cnt), None)
    assert rc == 10001 and b"width" in lib.nsm_last_error()
    s = _lib.NsmStrTable(None, None, None, None, None, 1, 1024, 10)  # rows of 1024 code units: unsupported
    rc = lib.nsm_indel_raw_grid(s, s, 0.5, 0, None, 0, ctypes.addressof(cnt), None)
    assert rc == 10002
    with pytest.raises(NotImplementedError):
        _lib.check(rc, "nsm_indel_raw_grid")


def test_no_cpu_fallback():
    import torch

    from napkon_string_matching_amd import _lib
    from napkon_string_matching_amd.compare.score_functions import fuzzy_match, intersection_vs_union

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.NsmLibraryError):
        intersection_vs_union(["a"], ["a"])
    with pytest.raises(_lib.NsmLibraryError):
        fuzzy_match("a", "a")


def test_product_never_imports_oracle():
    pkg = ROOT / "napkon-string-matching_amd"
    for path in pkg.rglob("*.py"):
        text = path.read_text()
        assert "oracle" not in text.replace("oracle/", "").replace("oracle.score_functions", "") or "import oracle" not in text, path
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), path


def test_set_table_encoding_cpu():
    import torch

    from napkon_string_matching_amd import tables

    ids = np.array([[5, -1, 9, 2], [-1, -1, -1, -1], [7, 8, -1, -1]], dtype=np.int32)
    t = tables.SetTable.from_padded(ids, "right", "cpu")
    assert t.width == 16 and t.n == 3 and t.has_empty
    assert t.cnt.tolist() == [3, 2, 0] and t.orig.tolist() == [0, 2, 1]
    ss = t.size_start.tolist()
    assert len(ss) == 18 and ss[0] == 0 and ss[-1] == 3 and ss[13] == 0 and ss[14] == 1 and ss[15] == 2 and ss[16] == 2
    row0 = t.ids[0].tolist()
    assert row0[:3] == [2, 5, 9] and set(row0[3:]) == {-2}  # RAW rows: ids ascending (the inverted index's global order)
    # right tables carry the global inverted index: one entry per (row, id) sorted by (id, position), 5 offsets per id
    assert t.vocab == 10 and t.post_start.shape[0] == 51 and t.post_sq == (2, 4, 5, 5, 5)
    # 3 rows of width 16, RAW: format 2 -- row | position << 24 | (size - 1) << 28 in the low word, above it the 27-bit fold of
    # the row's signature word
    assert t.post_row_bits == 24 and t.post_format == 2 and t.post.dtype == torch.int64
    raw = t.post[:5].numpy().view(np.uint64)
    entries = [(int(e) & 0xFFFFFF, (int(e) >> 24) & 0xF, ((int(e) >> 28) & 0xF) + 1) for e in raw]
    want = [(0, 0, 3), (0, 1, 3), (1, 0, 2), (1, 1, 2), (0, 2, 3)]  # ids 2, 5, 7, 8, 9 -> (row, position, size)
    assert entries == want
    folds = tables.sig_fold27(t.sig.numpy().view(np.uint64))
    assert [int(e) >> 32 for e in raw] == [int(folds[r]) for r, _, _ in want]
    sg = int(t.sig.numpy().view(np.uint64)[0])  # the fold keeps every hash bit: same ids, hash taken mod 27
    assert int(folds[0]) == ((sg & ((1 << 27) - 1)) | ((sg >> 27) & ((1 << 27) - 1)) | ((sg >> 54) & 0xF)) and 1 <= bin(int(folds[0])).count("1") <= 3
    tables.COMPACT_POSTINGS = False  # the 64-bit entries of tables with too many rows: row | position << 32 | size << 40
    try:
        t64 = tables.SetTable.from_padded(ids, "right", "cpu")
    finally:
        tables.COMPACT_POSTINGS = True
    assert t64.post_row_bits == 0 and t64.post_format == 0 and t64.post.dtype == torch.int64
    assert [(int(e) & 0xFFFFFFFF, (int(e) >> 32) & 0xFF, (int(e) >> 40) & 0xFF) for e in t64.post[:5].numpy().view(np.uint64)] == want
    assert t64.post_start.tolist() == t.post_start.tolist() and t64.post_sq == t.post_sq
    assert tables.post_row_bits(1 << 24, 16) == 24 and tables.post_row_bits((1 << 24) + 1, 16) == 0 and tables.post_row_bits(1 << 20, 64) == 20
    assert t.post_start[5 * 2: 5 * 2 + 6].tolist() == [0, 1, 1, 1, 1, 1] and int(t.post_start[-1]) == 5
    assert tables.SetTable.from_padded(ids, "left", "cpu").post is None
    assert tables.SetTable.from_padded(ids, "left", "cpu").ids[2].tolist() == [-1] * 16
    sig = tables.signatures(np.array([[1, 1 + (1 << 20)]], dtype=np.int32), np.array([2], np.int32))
    assert bin(int(sig[0])).count("1") in (1, 2)
    with pytest.raises(NotImplementedError):
        tables.pick_width(65)
    v = tables.Vocabulary()
    lv = tables.SetTable.from_levels([[["b"], ["b", "a"], ["c", "a", "b"]], [["z"]]], "left", "cpu", v)
    assert lv.nlev.tolist() == [3, 1] and lv.plen[0, :4].tolist() == [1, 2, 3, 3] and lv.plen[1, :2].tolist() == [1, 1]
    assert lv.ids[0, :3].tolist() == [v.id("b"), v.id("a"), v.id("c")]


def test_str_table_encoding_cpu():
    from napkon_string_matching_amd import tables

    lt, rt = tables.encode_strings(["abc", "", "ba"], ["cab"], "cpu")
    assert lt.alphabet == rt.alphabet == 3 and lt.has_empty
    assert lt.len.tolist() == [3, 2, 0] and lt.orig.tolist() == [0, 2, 1]
    assert lt.codes[0, :4].tolist() == [0, 1, 2, 3]  # pad code == alphabet size
    assert lt.hist.shape == (3, 32) and lt.hist[0, :4].tolist() == [1, 1, 1, 0] and int(lt.hist[2].sum()) == 0
    li, ls, ri, rs = tables.encode_level_strings([["a", "ab"], ["b"]], [["c"]], "cpu")
    assert li.nlev.tolist() == [2, 1] and li.first.tolist() == [0, 2] and ls.n == 3 and rs.n == 1


def test_containers():
    from napkon_string_matching_amd.types.comparable import Comparable
    from napkon_string_matching_amd.types.comparable_data import ComparableData, flatten_mapping
    from napkon_string_matching_amd.types.mapping import Mapping

    frame = pd.DataFrame({"HapIdentifier": ["h0", "h1", "h2"], "PopIdentifier": ["p0", "p1", "p2"],
                          "HapVariable": ["a", "b", "c"], "PopVariable": ["x", "y", "z"],
                          "MatchScore": [0.5, 0.75, 0.5]}, index=[7, 3, 2])
    c = Comparable(frame, "Hap", "Pop")
    assert list(c.match_variable) == ["a", "b", "c"] and list(c.variable) == ["x", "y", "z"]  # match_* -> LEFT
    c.sort_by_score()
    assert list(c.dataframe().index) == [3, 2, 7]
    again = Comparable(data={"left_name": "Hap", "right_name": "Pop", "data": c.dataframe().to_dict(orient="records")})
    assert list(again.match_score) == [0.75, 0.5, 0.5]
    with pytest.raises(AttributeError):
        Comparable(data={"x": 1})
    m = Mapping({"u1": {"hap": ["h1", "h2"], "pop": ["p1"]}, "u2": {"hap": ["h3"], "suep": ["s1"]}})
    assert flatten_mapping("hap", "pop", m) == [("h1", "p1"), ("h2", "p1")]
    with pytest.raises(KeyError):
        m.filter_by_group("pop")
    assert ComparableData.gen_comp_value(["a b", "c"]) == [["c"], ["a", "b", "c"]]
    assert ComparableData.gen_comp_value("abca") == [["a"], ["a", "c"], ["a", "b", "c"], ["a", "b", "c"]]


def test_matcher_pair_enumeration(monkeypatch):
    from napkon_string_matching_amd.matcher import Matcher
    from napkon_string_matching_amd.types.comparable_data import ComparableData

    calls = []

    def fake_compare(self, other, **kwargs):
        calls.append((kwargs["left_name"], kwargs["right_name"], kwargs["compare_column"], kwargs["score_threshold"]))
        return "result"

    monkeypatch.setattr(ComparableData, "compare", fake_compare)
    q = {name: ComparableData(pd.DataFrame({"Identifier": [name]})) for name in ("suep", "Hap", "pop")}
    cfg = {"matching": {"score_threshold": 0.7, "compare_column": "Term", "score_func": "fuzzy_match",
                        "variable_score_threshold": 0.9}}
    m = Matcher(None, cfg, questionnaires=q, gecco=ComparableData(pd.DataFrame({"Identifier": ["g"]})))
    m.match_questionnaires()
    assert list(m.results.results) == ["Hap vs suep", "pop vs suep", "Hap vs pop"] or sorted(m.results.results) == [
        "Hap vs pop", "Hap vs suep", "pop vs suep"]
    assert {c[:2] for c in calls} == {("Hap", "suep"), ("Hap", "pop"), ("pop", "suep")}
    calls.clear()
    m.clear_results()
    m.match_questionnaires_variables()
    assert all(c[2] == "Variable" and c[3] == 0.9 for c in calls) and len(calls) == 3
    assert all(k.startswith("var_") for k in m.results.results)
    m.match_gecco_with_questionnaires()
    assert {"gecco vs suep", "gecco vs Hap", "gecco vs pop"} <= set(m.results.results)


def test_shard_bounds():
    from napkon_string_matching_amd.distributed import shard_bounds

    for n in (0, 1, 7, 8, 9, 1000):
        for w in (1, 2, 3, 8):
            blocks = [shard_bounds(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(hi - lo for lo, hi in blocks) <= -(-n // w) if n else True


_WORKER = r'''
import os, sys
sys.path.insert(0, {pkg!r})
import numpy as np, torch, torch.distributed as dist
from napkon_string_matching_amd.distributed import all_gather_hits, shard_bounds, world
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, size = world()
rng = np.random.default_rng(5)
n = 41
score = rng.integers(0, 6, n) / 5.0
i = rng.integers(0, 20, n); j = np.arange(n)
lo, hi = shard_bounds(20, rank, size)
mine = (i >= lo) & (i < hi)
s, gi, gj = all_gather_hits(score[mine], i[mine], j[mine])
order = np.lexsort((j, i, -score))
assert np.array_equal(s, score[order]) and np.array_equal(gi, i[order]) and np.array_equal(gj, j[order]), rank
# an empty contribution from one rank must work too
s2, _, _ = all_gather_hits(score[:0] if rank else score, i[:0] if rank else i, j[:0] if rank else j)
assert len(s2) == n
# the device-resident form of the same exchange (gen_comparable without a blacklist): every rank hands over its hit
# BUFFER with ids relative to its own sub-grid (left shard's items that have levels x right items that have levels)
from napkon_string_matching_amd import grid
from napkon_string_matching_amd.distributed import agree_all, all_gather_pending
assert agree_all(True) is True and agree_all(rank == 0) is False
n_left, n_right = 23, 9
nlev_l = np.array([0 if k % 5 == 2 else 3 for k in range(n_left)])
nlev_r = np.array([0 if k == 4 else 2 for k in range(n_right)])
keep_r = np.flatnonzero(nlev_r > 0)
want = []
pending = None
for r in range(size):
    lo, hi = shard_bounds(n_left, r, size)
    keep_l = lo + np.flatnonzero(nlev_l[lo:hi] > 0)
    rr = np.random.default_rng(100 + r)
    m = 0 if r == 1 and size > 1 and False else 17 + r
    li, lj = rr.integers(0, len(keep_l), m), rr.integers(0, len(keep_r), m)
    sc = rr.integers(0, 4, m) / 3.0
    want += [(float(a), int(keep_l[b]), int(keep_r[c])) for a, b, c in zip(sc, li, lj)]
    if r == rank:
        buf = grid.HitBuffer(64, "cpu")
        rec = buf.records.numpy()
        rec[:m, 0] = sc
        rec.view(np.int32).reshape(-1, 4)[:m, 2] = li
        rec.view(np.int32).reshape(-1, 4)[:m, 3] = lj
        buf.count.fill_(m)
        pending = grid.PendingHits(buf, m, 0)
s, gi, gj = all_gather_pending(pending, n_left, nlev_l, nlev_r, size)
want.sort(key=lambda t: (-t[0], t[1], t[2]))
assert [(float(a), int(b), int(c)) for a, b, c in zip(s, gi, gj)] == want, rank
# a rank without anything to score contributes an empty buffer
s3, _, _ = all_gather_pending(pending if rank == 0 else None, n_left, nlev_l, nlev_r, size)
assert len(s3) == 17
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_all_gather_hits_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(pkg=str(ROOT / "napkon-string-matching_amd")))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()


def test_category_partition_encoding_cpu():
    """The host side of the category partition: rows per category, lowest-common-category dedupe is
    the kernel's job, 'both empty' becomes category 63."""
    from napkon_string_matching_amd import _lib, tables

    cat = np.array([0b101, 0, 0b010, 0b100], dtype=np.uint64)
    items = [[["a"]], [["b"]], [["c"], ["c", "d"]], [["e"]]]
    v = tables.Vocabulary()
    t = tables.SetTable.from_levels(items, "left", "cpu", v, categories=cat, category_mode=_lib.CAT_INTERSECT_OR_BOTH_EMPTY)
    assert t.category_mode == _lib.CAT_INTERSECT and t.n == 5  # item 0 twice, the empty item in category 63
    assert t.seg.tolist() == [0, 1, 2, 2, 63]
    assert sorted(t.orig[t.seg == 2].tolist()) == [0, 3] and t.orig[-1].item() == 1
    ss = t.seg_start.tolist()
    assert ss[0] == 0 and ss[1] == 1 and ss[2] == 2 and ss[3] == 4 and ss[63] == 4 and ss[64] == 5
    assert t.filt.shape == (5, 8)
    plain = tables.SetTable.from_levels(items, "left", "cpu", v, categories=cat, category_mode=_lib.CAT_INTERSECT_OR_BOTH_EMPTY,
                                        partition=False)
    assert plain.seg is None and plain.n == 4 and plain.category_mode == _lib.CAT_INTERSECT_OR_BOTH_EMPTY
    none = tables.SetTable.from_levels(items, "left", "cpu", v)
    assert none.category_mode == _lib.CAT_NONE and none.cat is None
    li, ls, ri, rs = tables.encode_level_strings([["ab"], ["c"]], [["ab"]], "cpu", cat[:2], cat[:1], _lib.CAT_INTERSECT)
    assert li.seg.tolist() == [0, 2] and li.n == 2 and ri.seg.tolist() == [0, 2]  # the empty item is dropped


def test_results_writers(tmp_path):
    from napkon_string_matching_amd.matcher import Matcher
    from napkon_string_matching_amd.types.comparable import Comparable, ComparisonResults

    frame = pd.DataFrame({"HapIdentifier": ["h0"], "PopIdentifier": ["p0"], "HapVariable": ["gec_a"],
                          "PopVariable": ["x"], "MatchScore": [0.9]}, index=[4])
    res = ComparisonResults({"hap vs pop": Comparable(frame, "Hap", "Pop")})
    res.write_csv_dir(tmp_path / "csv")
    assert (tmp_path / "csv" / "hap_vs_pop.csv").read_text().startswith(",HapIdentifier")
    m = Matcher(None, {"matching": {"score_threshold": 0.7, "compare_column": "Term", "score_func": "fuzzy_match"},
                       "output_dir": str(tmp_path)})
    m.results = res
    m.write_results()  # xlsx if an engine is installed, else the CSV directory
    produced = list(tmp_path.glob("result_0.7_Term_fuzzy-match*"))
    assert produced
    assert m._analyse() == {"hap vs pop": {"matched": "1/1", "gecco": "0/1"}}


def test_signature_word_layout():
    """include/nsm_hip.h: 58 hash bits, in-row collisions in unary in the top 6 bits, all ones beyond 6."""
    from napkon_string_matching_amd import tables

    cand = np.arange(400_000, dtype=np.int32)
    h16 = ((cand.astype(np.uint32) * np.uint32(0x9E3779B1)) >> np.uint32(16)) & np.uint32(0xFFFF)
    pos = (h16.astype(np.uint64) * np.uint64(58)) >> np.uint64(16)
    on5, on9 = cand[pos == 5], cand[pos == 9]
    for k in range(1, 10):
        row = np.full((1, 16), -1, dtype=np.int32)
        row[0, :k] = on5[:k]
        row[0, k] = on9[0]
        word = int(tables.signatures(row, np.array([k + 1], np.int32))[0])
        if k - 1 > 6:
            assert word == (1 << 64) - 1
        else:
            assert word & ((1 << 58) - 1) == (1 << 5) | (1 << 9)
            assert word >> 58 == (1 << (k - 1)) - 1  # k ids on one bit: k - 1 collisions
    # an id beyond cnt is ignored
    row = np.array([[int(on5[0]), int(on9[0])] + [-1] * 14], dtype=np.int32)
    assert int(tables.signatures(row, np.array([1], np.int32))[0]) == 1 << 5


def test_comparable_frame_conveniences():
    from napkon_string_matching_amd.types.comparable import Comparable

    frame = pd.DataFrame({"HapIdentifier": ["h0", "h1"], "PopIdentifier": ["p0", None], "HapVariable": ["a", "b"],
                          "PopVariable": ["x", "y"], "Extra": [1, 2], "MatchScore": [0.9, 0.4]})
    comp = Comparable(frame, "Hap", "Pop")
    assert str(comp) == str(frame) and len(comp.dropna()) == 1 and comp.dropna().left_name == "Hap"
    assert list(comp.drop(columns=["Extra"]).dataframe().columns) == [c for c in frame.columns if c != "Extra"]
    merged = comp.merge(pd.DataFrame({"HapIdentifier": ["h0"], "Note": ["n"]}), on="HapIdentifier")
    assert isinstance(merged, Comparable) and list(merged["Note"]) == ["n"] and merged.right_name == "Pop"
    comp.drop_superfluous_columns()
    assert "Extra" not in comp.dataframe().columns and "MatchScore" in comp.dataframe().columns
    assert list(comp.match_variable) == ["a", "b"] and list(comp.variable) == ["x", "y"]  # match_<col> is the LEFT side


_CACHE_WORKER = r'''
import os, sys
sys.path.insert(0, {pkg!r})
import pandas as pd
import torch.distributed as dist
from napkon_string_matching_amd.types.comparable import Comparable
from napkon_string_matching_amd.types.comparable_data import CACHE_FILE_PATTERN, ComparableData
from napkon_string_matching_amd.types.questionnaire import Questionnaire

rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)
frame = lambda tag: pd.DataFrame({{"Identifier": [tag + "1", tag + "2"], "Sheet": ["s", "s"], "Category": [["c"], ["c"]],
                                  "Variable": ["v1", "v2"], "Term": [["x"], ["y"]], "Tokens": [["a", "b"], ["c"]]}})
left, right = Questionnaire(frame("l")), Questionnaire(frame("r"))
kw = dict(score_func="intersection_vs_union", left_name="hap", right_name="pop", filter_categories=False)
# every rank has its OWN cache directory (a cache_dir that is not shared between nodes); only rank 0's holds the file
cache_dir = os.path.join({tmp!r}, "rank%d" % rank)
os.makedirs(cache_dir, exist_ok=True)
key = left._hash_compare_args(right, None, None, "Tokens", 0.2, kw)
if rank == 0:
    rows = pd.DataFrame({{"HapIdentifier": ["l1"], "PopIdentifier": ["r1"], "MatchScore": [0.5]}})
    Comparable(rows, "Hap", "Pop").write_json(os.path.join(cache_dir, CACHE_FILE_PATTERN.format(key)))


def must_not_run(*a, **k):
    raise AssertionError("rank %d entered gen_comparable although rank 0 holds the cached result" % rank)


ComparableData.gen_comparable = must_not_run
got = left.compare(right, None, None, compare_column="Tokens", score_threshold=0.3, cache_threshold=0.2,
                   cache_dir=cache_dir, **kw)
assert len(got) == 1 and got.match_score.tolist() == [0.5] and got.left_name == "Hap", (rank, got)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_compare_cache_decision_is_collective_gloo_world2(tmp_path):
    """Round-2 advice: with a rank-local ``exists()`` a cache that only one rank can see sends the ranks down
    different paths and the ones that miss hang in gen_comparable's collectives.  Rank 0 decides for everybody and
    hands the cached rows to the ranks that cannot read the file."""
    script = tmp_path / "cache_worker.py"
    script.write_text(_CACHE_WORKER.format(pkg=str(ROOT / "napkon-string-matching_amd"), tmp=str(tmp_path)))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out.decode()


def test_wide_item_routing_cpu():
    """napkon_string_matching_amd/wide.py: which items leave the fast kernels, and that the three sub-grids
    (regular x regular, wide x all, regular x wide) together cover every pair exactly once, merged in canonical order."""
    from napkon_string_matching_amd import grid, wide

    long_s = "x" * 600
    items_l = [["ab", "ab cd"], [long_s], ["ef"]]
    items_r = [["gh"], ["ab", long_s + "y"], ["ij"], ["kl"]]
    wl, wr = wide.wide_string_items(items_l, items_r)
    assert wl.tolist() == [False, True, False] and wr.tolist() == [False, True, False, False]
    assert wide.wide_string_items([["ab"]], [["cd"]]) is None
    # more than 255 distinct code units: the items holding one of the rarest symbols become wide
    many = "".join(chr(0x4E00 + k) for k in range(300))
    wl, wr = wide.wide_string_items([["aaaa " * 50], [many]], [["aaaa"]])
    assert wl.tolist() == [False, True] and wr.tolist() == [False]
    sets_l = [[["t%d" % k for k in range(70)]], [["a", "b"]]]
    sets_r = [[["a"]], [["t%d" % k for k in range(10)] * 9]]  # 90 tokens, 10 distinct: not wide
    wl, wr = wide.wide_set_items(sets_l, sets_r)
    assert wl.tolist() == [True, False] and wr.tolist() == [False, False]

    seen = []

    def fake(tag):
        def run(li, ri):
            li, ri = list(li), list(ri)
            seen.extend((tag, int(a), int(b)) for a in li for b in ri)
            n = len(li) * len(ri)
            i = np.repeat(np.arange(len(li)), len(ri)).astype(np.int32)
            j = np.tile(np.arange(len(ri)), len(li)).astype(np.int32)
            score = np.array([1.0 / (1 + li[a] + 10 * ri[b]) for a, b in zip(i, j)], dtype=np.float64)
            return grid.Hits(score, i, j) if n else grid.Hits(np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.int32))
        return run

    wl, wr = np.array([False, True, False, True]), np.array([True, False, False])
    hits = wide.split_grid(wl, wr, fake("fast"), fake("any"))
    pairs = sorted((a, b) for _, a, b in seen)
    assert pairs == [(a, b) for a in range(4) for b in range(3)]  # every pair once
    assert {t for t, a, b in seen if not wl[a] and not wr[b]} == {"fast"}
    assert {t for t, a, b in seen if wl[a] or wr[b]} == {"any"}
    want = sorted(((1.0 / (1 + a + 10 * b), a, b) for a in range(4) for b in range(3)), key=lambda h: (-h[0], h[1], h[2]))
    assert hits.as_tuples() == want
