#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running THE REFERENCE ITSELF.

Run in the build container only (it needs /root/reference):

    python tests/golden/make_golden.py

The reference's hot path imports two third-party packages that are absent offline:
``rapidfuzz`` (score_functions.py:3) and ``nltk`` (comparable_data.py:7-10,20-21, which
also calls ``nltk.download`` at import time).  They are replaced by INERT stand-ins:

* ``rapidfuzz.fuzz.QRatio`` raises -- no ``fuzzy_match`` number in any fixture comes
  from a stand-in; that score function stays "parity unpinned".
* ``nltk.download`` does nothing, ``stopwords.words`` is empty, ``word_tokenize`` is
  ``str.split``.  All fixture inputs are built from ``[A-Za-z0-9]+`` words that are not
  German stop words, for which punkt gives the same tokens.

Everything written here is DATA (inputs and the reference's outputs); no reference
source text is stored.  The fixtures are what ``tests/test_oracle_golden.py`` pins the
oracle with, and what the GPU parity tests replay through the HIP path.
"""
import json
import random
import sys
import tempfile
import types
import zlib
from pathlib import Path

HERE = Path(__file__).resolve().parent
REFERENCE = Path("/root/reference")


def install_stand_ins():
    rf = types.ModuleType("rapidfuzz")
    fz = types.ModuleType("rapidfuzz.fuzz")

    def _absent(*_a, **_k):
        raise RuntimeError("rapidfuzz is not installed; fuzzy_match is parity unpinned")

    fz.QRatio = _absent
    rf.fuzz = fz
    nl = types.ModuleType("nltk")
    nl.download = lambda *_a, **_k: None
    corpus = types.ModuleType("nltk.corpus")

    class _Stop:
        @staticmethod
        def words(_language):
            return []

    corpus.stopwords = _Stop()
    tok = types.ModuleType("nltk.tokenize")
    tok.word_tokenize = lambda text: text.split()
    nl.corpus, nl.tokenize = corpus, tok
    sys.modules.update(
        {"rapidfuzz": rf, "rapidfuzz.fuzz": fz, "nltk": nl, "nltk.corpus": corpus, "nltk.tokenize": tok}
    )
    sys.path.insert(0, str(REFERENCE))


def dump(name, payload):
    path = HERE / name
    path.write_text(json.dumps(payload, indent=1) + "\n", encoding="utf-8")
    print("wrote", path.name)


def call(fn, *args):
    try:
        return {"value": fn(*args)}
    except Exception as exc:  # the exception TYPE is part of the contract
        return {"raises": type(exc).__name__}


# ----------------------------------------------------------------------------- inputs
def cohort(rng, name, n, vocab=500, max_tokens=8, n_categories=8, none_every=0):
    """C1-shaped cohort: Tokens = k in [1, max_tokens] words ``t<u>``."""
    rows = []
    for k in range(n):
        toks = [f"t{rng.randrange(vocab)}" for _ in range(rng.randint(1, max_tokens))]
        cats = sorted(rng.sample([f"cat{c}" for c in range(n_categories)], rng.randint(1, 2)))
        rows.append(
            {
                "Identifier": f"{name}#sheet{k % 7}#{k}",
                "Variable": f"{name}_var_{k}",
                "Sheet": f"sheet{k % 7}",
                "Category": cats,
                "Term": [f"header{k % 5}", f"question {k}"],
                "Tokens": None if none_every and k % none_every == none_every - 1 else toks,
                "Parameter": f"param{k}",
            }
        )
    return rows


def run_compare(Questionnaire, Mapping, case):
    import pandas as pd

    left = Questionnaire(pd.DataFrame(case["left"]))
    right = Questionnaire(pd.DataFrame(case["right"]))
    wl = Mapping(data=case["whitelist"])
    bl = Mapping(data=case["blacklist"])
    out = {}
    try:
        g = left.gen_comparable(right, wl, bl, **case["gen_kwargs"])
        frame = g.dataframe() if hasattr(g, "dataframe") else g
        while not isinstance(frame, pd.DataFrame):
            frame = frame._data
        out["gen_comparable"] = {
            "index": [int(v) for v in frame.index],
            "columns": list(frame.columns),
            "records": json.loads(frame.drop(columns=["MatchScore"]).to_json(orient="records")),
            "scores": [float(v) for v in frame["MatchScore"]],
        }
    except Exception as exc:
        out["gen_comparable"] = {"raises": type(exc).__name__}
    if "compare_kwargs" in case:
        with tempfile.TemporaryDirectory() as tmp:
            try:
                c = left.compare(right, wl, bl, cache_dir=tmp, cached=False, **case["compare_kwargs"])
                frame = c.data
                while not isinstance(frame, pd.DataFrame):
                    frame = frame._data
                out["compare"] = {
                    "index": [int(v) for v in frame.index],
                    "scores": [float(v) for v in frame["MatchScore"]],
                    "left_name": c.left_name,
                    "right_name": c.right_name,
                }
            except Exception as exc:
                out["compare"] = {"raises": type(exc).__name__}
    return out


def main():
    install_stand_ins()
    from napkon_string_matching.compare.score_functions import intersection_vs_union, join_sorted
    from napkon_string_matching.types.comparable_data import (
        ComparableData,
        categories_matching,
        flatten_list,
        flatten_mapping,
    )
    from napkon_string_matching.types.mapping import Mapping
    from napkon_string_matching.types.questionnaire import Questionnaire

    rng = random.Random(20240)

    # -- 1. score functions -------------------------------------------------------------
    ivu_cases = [
        [["a", "b", "c"], ["b", "c", "d", "e"]],
        ["a b c", "b c d e"],
        [["a", "a", "b"], ["a"]],
        [[], ["a"]],
        [["a"], []],
        [[], []],
        ["", ""],
        [["x"], ["x"]],
        [["x"], "x y"],
        ["  spaced   out ", ["out", "spaced"]],
        [["A"], ["a"]],
    ]
    for _ in range(200):
        a = [f"w{rng.randrange(12)}" for _ in range(rng.randrange(0, 9))]
        b = [f"w{rng.randrange(12)}" for _ in range(rng.randrange(0, 9))]
        ivu_cases.append([a, b])
    js_cases = [["b", "A", "c", "B"], [], ["z"], ["b", "B", "a", "A"], ["10", "9", "a1", "A0"]]
    dump(
        "score_functions.json",
        {
            "intersection_vs_union": [{"args": c, **call(intersection_vs_union, *c)} for c in ivu_cases],
            "join_sorted": [{"args": [c], **call(join_sorted, c)} for c in js_cases],
        },
    )

    # -- 2. compare_terms / gen_comp_value ----------------------------------------------
    trace = []

    def tracing(l, r):
        trace.append([l, r])
        return intersection_vs_union(l, r)

    ct_cases = []
    shapes = [(1, 1), (2, 2), (3, 3), (3, 2), (1, 4), (3, 1), (4, 1), (2, 5), (0, 0), (0, 2), (2, 0)]
    for nl_, nr_ in shapes + [(rng.randint(1, 6), rng.randint(1, 6)) for _ in range(120)]:
        def levels(n):
            base, out = [], []
            for _ in range(n):
                base = sorted(set(base + [f"k{rng.randrange(10)}" for _ in range(rng.randint(0, 3))]))
                out.append(list(base))
            return out

        l, r = levels(nl_), levels(nr_)
        if nl_ and nr_ and not l[-1] and not r[-1]:
            l[-1] = ["k0"]
            l = [sorted(set(x)) for x in l]
        trace.clear()
        res = call(ComparableData.compare_terms, l, r, tracing)
        ct_cases.append({"left": l, "right": r, "trace": [list(t) for t in trace], **res})
    # identical items: 0.5 / 0.75 / 0.875
    for n in (1, 2, 3, 4):
        lv = [[f"q{m}" for m in range(k + 1)] for k in range(n)]
        ct_cases.append({"left": lv, "right": lv, **call(ComparableData.compare_terms, lv, lv, intersection_vs_union)})
    gcv_cases = [
        ["a b", "c"],
        ["only"],
        [],
        ["x y z", "x", "W q"],
        [["nested", "list"], "tail word"],
        "abca",
        ["b A", "d C e"],  # (case-variant duplicates like 'a'/'A' tie under casefold and come out in
        # set-iteration order, which changes with PYTHONHASHSEED: not used in fixtures)
        ["p . q", "r , s ( t )"],
    ]
    dump(
        "compare_terms.json",
        {
            "compare_terms": ct_cases,
            "gen_comp_value": [{"args": [c], **call(ComparableData.gen_comp_value, c)} for c in gcv_cases],
            "flatten_list": [
                {"args": [c], **call(flatten_list, c)} for c in (["a", ["b", "c"], "d"], "xyz", [], [["a"], ["b"]])
            ],
        },
    )

    # -- 3. categories / blacklist predicates -------------------------------------------
    import pandas as pd

    cat_cases = []
    for left_vals, right_vals in [
        ([["a"], ["b"], [], ["a", "c"]], [["a", "b"], [], ["c"], []]),
        (["a", "b", "c", "a"], [["a", "b"], [], ["c"], ["b"]]),
        ([["a"], ["b"], [], ["a", "c"]], ["a", "b", "c", "a"]),
        (["a", "b", None, "a"], ["a", "c", None, "b"]),
    ]:
        df = pd.DataFrame({"L": left_vals}).merge(pd.DataFrame({"R": right_vals}), how="cross")
        try:
            kept = {"kept": [int(v) for v in categories_matching(df, "L", "R").index]}
        except Exception as exc:  # list x scalar hashes a list (:471) -> TypeError
            kept = {"raises": type(exc).__name__}
        cat_cases.append({"left": left_vals, "right": right_vals, **kept})
    try:
        categories_matching(pd.DataFrame({"L": [], "R": []}), "L", "R")
        empty = {"value": None}
    except Exception as exc:
        empty = {"raises": type(exc).__name__}
    bl_map = {
        "u1": {"hap": ["h1", "h2"], "pop": ["p1"]},
        "u2": {"hap": ["h3"], "suep": ["s1"]},
        "u3": {"pop": ["p2"], "hap": ["h4"], "suep": ["s2", "s3"]},
    }
    flat = {
        f"{a}|{b}": [list(t) for t in flatten_mapping(a, b, Mapping(data=bl_map))]
        for a, b in (("hap", "pop"), ("pop", "hap"), ("hap", "suep"), ("pop", "suep"), ("hap", "nope"))
    }
    dump("predicates.json", {"categories_matching": cat_cases, "categories_matching_empty": empty,
                             "flatten_mapping": {"mapping": bl_map, "pairs": flat}})

    # -- 4. pair grids -------------------------------------------------------------------
    grids = {}

    # 4a. the 4x4 case of SURVEY.md 8c(5): a None compare value, one blacklisted pair, list categories
    l4 = [
        {"Identifier": "h0", "Variable": "hv0", "Sheet": "s", "Category": ["c1"], "Term": ["A", "b c"], "Tokens": ["a b", "c"], "Parameter": "p"},
        {"Identifier": "h1", "Variable": "hv1", "Sheet": "s", "Category": ["c2"], "Term": ["D"], "Tokens": None, "Parameter": "p"},
        {"Identifier": "h2", "Variable": "hv2", "Sheet": "s", "Category": ["c1", "c2"], "Term": ["E", "f"], "Tokens": ["x", "c"], "Parameter": "p"},
        {"Identifier": "h3", "Variable": "hv3", "Sheet": "t", "Category": [], "Term": [["G", "h"], "i"], "Tokens": ["a", "b c"], "Parameter": "p"},
    ]
    r4 = [
        {"Identifier": "p0", "Variable": "pv0", "Sheet": "s", "Category": ["c1"], "Term": ["A"], "Tokens": ["a b", "c"], "Parameter": "p"},
        {"Identifier": "p1", "Variable": "pv1", "Sheet": "s", "Category": ["c3"], "Term": ["B"], "Tokens": ["c"], "Parameter": "p"},
        {"Identifier": "p2", "Variable": "pv2", "Sheet": "s", "Category": [], "Term": ["C"], "Tokens": ["q", "r s"], "Parameter": "p"},
        {"Identifier": "p3", "Variable": "pv3", "Sheet": "u", "Category": ["c2"], "Term": ["D"], "Tokens": ["b", "a", "c"], "Parameter": "p"},
    ]
    base_kwargs = dict(score_func="intersection_vs_union", compare_column="Tokens", left_name="hap", right_name="pop")
    for name, extra, cmp_extra in [
        ("small4_plain", dict(score_threshold=0.0), None),
        ("small4_thr", dict(score_threshold=0.1), dict(score_threshold=0.3)),
        ("small4_categories", dict(score_threshold=0.1, filter_categories=True), dict(score_threshold=0.3, filter_categories=True)),
        ("small4_cache_thr", dict(score_threshold=0.1), dict(score_threshold=0.6, cache_threshold=0.2)),
    ]:
        case = {
            "left": l4, "right": r4, "whitelist": {},
            "blacklist": {"b1": {"hap": ["h0"], "pop": ["p1"]}},
            "gen_kwargs": {**base_kwargs, **extra},
        }
        if cmp_extra is not None:
            case["compare_kwargs"] = {**base_kwargs, **cmp_extra}
        case["expected"] = run_compare(Questionnaire, Mapping, case)
        grids[name] = case

    # 4b. whitelist removal, and the KeyError that silently skips it
    wl_ok = {"w1": {"hap": ["h0"], "pop": ["p3"]}, "w2": {"hap": ["h2"], "pop": []}}
    wl_keyerror = {"w1": {"hap": ["h0"], "pop": ["p3"]}, "w2": {"hap": ["h2"]}}
    for name, wl in (("small4_whitelist", wl_ok), ("small4_whitelist_keyerror", wl_keyerror)):
        case = {"left": l4, "right": r4, "whitelist": wl, "blacklist": {},
                "gen_kwargs": {**base_kwargs, "score_threshold": 0.0}}
        case["expected"] = run_compare(Questionnaire, Mapping, case)
        grids[name] = case

    # 4c. variables: compare_column is a str -> character-suffix levels
    case = {"left": l4, "right": r4, "whitelist": {}, "blacklist": {},
            "gen_kwargs": {**base_kwargs, "compare_column": "Variable", "score_threshold": 0.2},
            "compare_kwargs": {**base_kwargs, "compare_column": "Variable", "score_threshold": 0.5}}
    case["expected"] = run_compare(Questionnaire, Mapping, case)
    grids["small4_variable"] = case

    # 4d. error surfaces: empty-vs-empty level -> ZeroDivisionError; zero-level item -> IndexError
    lz = [dict(l4[0], Tokens=[".", "a"]), dict(l4[2], Tokens=[".", ","])]
    rz = [dict(r4[0], Tokens=["?", "!"]), dict(r4[1], Tokens=["c"])]
    case = {"left": lz, "right": rz, "whitelist": {}, "blacklist": {}, "gen_kwargs": {**base_kwargs, "score_threshold": 0.0}}
    case["expected"] = run_compare(Questionnaire, Mapping, case)
    grids["error_zero_division"] = case
    case = {"left": lz, "right": rz, "whitelist": {}, "blacklist": {"b": {"hap": ["h2"], "pop": ["p0"]}},
            "gen_kwargs": {**base_kwargs, "score_threshold": 0.0}}
    case["expected"] = run_compare(Questionnaire, Mapping, case)
    grids["error_zero_division_blacklisted_away"] = case
    case = {"left": [dict(l4[0], Tokens=[])], "right": r4[:2], "whitelist": {}, "blacklist": {},
            "gen_kwargs": {**base_kwargs, "score_threshold": 0.0}}
    case["expected"] = run_compare(Questionnaire, Mapping, case)
    grids["error_index"] = case
    case = {"left": l4, "right": r4, "whitelist": {},
            "blacklist": {"b": {"hap": ["h0", "h2", "h3"], "pop": ["p0", "p1", "p2", "p3"]}},
            "gen_kwargs": {**base_kwargs, "score_threshold": 0.0, "filter_categories": True}}
    case["expected"] = run_compare(Questionnaire, Mapping, case)
    grids["error_categories_on_empty_grid"] = case

    # 4e. GECCO-shaped left side: Category is a str -> the `x in set(y)` branch
    lg = [dict(row, Category=(row["Category"] or ["c9"])[0]) for row in l4]
    case = {"left": lg, "right": r4, "whitelist": {}, "blacklist": {},
            "gen_kwargs": {**base_kwargs, "left_name": "gecco", "score_threshold": 0.0, "filter_categories": True}}
    case["expected"] = run_compare(Questionnaire, Mapping, case)
    grids["small4_scalar_category"] = case

    # 4f. medium random grids
    for name, n, m, thr, cats, none_every in [
        ("rand_30x40", 30, 40, 0.05, False, 7),
        ("rand_40x30_categories", 40, 30, 0.05, True, 0),
    ]:
        r2 = random.Random(zlib.crc32(name.encode()) % 1000 + 5)  # not hash(): str hashes change per process
        left, right = cohort(r2, "hap", n, vocab=40, max_tokens=5, none_every=none_every), cohort(r2, "suep", m, vocab=40, max_tokens=5)
        bl = {f"b{k}": {"hap": [left[r2.randrange(n)]["Identifier"]], "suep": [right[r2.randrange(m)]["Identifier"] for _ in range(2)]} for k in range(6)}
        kw = dict(score_func="intersection_vs_union", compare_column="Tokens", left_name="hap", right_name="suep", filter_categories=cats)
        case = {"left": left, "right": right, "whitelist": {}, "blacklist": bl,
                "gen_kwargs": {**kw, "score_threshold": thr}, "compare_kwargs": {**kw, "score_threshold": 0.3, "cache_threshold": 0.1}}
        case["expected"] = run_compare(Questionnaire, Mapping, case)
        grids[name] = case
    dump("pair_grids.json", grids)

    # -- 5. C1: hap x pop, 100 items each, Tokens, threshold 0.1 (BASELINE.json configs[0]) ----
    r0 = random.Random(0)
    left, right = cohort(r0, "hap", 100), cohort(r0, "pop", 100)
    kw = dict(score_func="intersection_vs_union", compare_column="Tokens", left_name="hap", right_name="pop")
    case = {"left": left, "right": right, "whitelist": {}, "blacklist": {},
            "gen_kwargs": {**kw, "score_threshold": 0.1}, "compare_kwargs": {**kw, "score_threshold": 0.1}}
    case["expected"] = run_compare(Questionnaire, Mapping, case)
    dump("c1_hap_pop_100.json", case)
    print("C1 hits:", len(case["expected"]["gen_comparable"]["index"]))

    # -- 6. result consumers: Matcher._analyse (matcher.py:290-312) on results the reference computed ----------
    # The Matcher module pulls in ingestion back ends that are absent offline (psycopg2, bs4, openpyxl): inert
    # stand-ins, none of them is reached -- the object is built with __new__ and only `results` is set.
    for name in ("psycopg2", "psycopg2.extras", "bs4", "openpyxl"):
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    from napkon_string_matching.matcher import Matcher
    from napkon_string_matching.types.comparable import ComparisonResults

    r6 = random.Random(66)

    def with_gecco_variables(rows, every):
        out = []
        for k, row in enumerate(rows):
            row = dict(row)
            row["Variable"] = ("gec_" if k % every == 0 else "var_") + row["Identifier"].lower() + ("" if k % 5 else "_x")
            if k % 4 == 3:  # several items share a variable name: nunique < number of rows
                row["Variable"] = out[k - 1]["Variable"]
            out.append(row)
        return out

    cohorts6 = {
        "hap": with_gecco_variables(cohort(r6, "hap", 25, vocab=30, max_tokens=4), 3),
        "pop": with_gecco_variables(cohort(r6, "pop", 30, vocab=30, max_tokens=4), 4),
        "suep": with_gecco_variables(cohort(r6, "suep", 20, vocab=30, max_tokens=4), 1000003),  # one gec_ variable only
    }
    kw6 = dict(score_func="intersection_vs_union", compare_column="Tokens", score_threshold=0.25)
    steps = [("hap", "pop", 0.25), ("hap", "suep", 0.25), ("pop", "suep", 0.25), ("pop", "hap", 1.1)]  # the last: no hit
    results, expected_rows = {}, {}
    for a, b, thr in steps:
        with tempfile.TemporaryDirectory() as tmp:
            comp = Questionnaire(pd.DataFrame(cohorts6[a])).compare(
                Questionnaire(pd.DataFrame(cohorts6[b])), Mapping(data={}), Mapping(data={}), cache_dir=tmp, cached=False,
                left_name=a, right_name=b, **{**kw6, "score_threshold": thr})
        key = f"{a} vs {b}"
        results[key] = comp
        frame = comp.data
        while not isinstance(frame, pd.DataFrame):
            frame = frame._data
        expected_rows[key] = len(frame)
    matcher = Matcher.__new__(Matcher)
    matcher.results = ComparisonResults(comp_dict=results)
    dump("analyse.json", {"cohorts": cohorts6, "compare_kwargs": kw6, "steps": [list(s_) for s_ in steps],
                          "rows": expected_rows, "analysis": matcher._analyse()})


if __name__ == "__main__":
    main()
