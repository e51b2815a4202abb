"""Pin the CPU oracle to the reference: every fixture under tests/golden/ was produced by
importing the reference itself (tests/golden/make_golden.py)."""
import pandas as pd
import pytest

from oracle import compare as oc
from oracle import score_functions as osf


def _check(case, fn, *args):
    if "raises" in case:
        with pytest.raises(Exception) as err:
            fn(*args)
        assert type(err.value).__name__ == case["raises"]
    else:
        assert fn(*args) == case["value"]


def test_intersection_vs_union(golden):
    cases = golden("score_functions.json")["intersection_vs_union"]
    assert len(cases) > 200
    for case in cases:
        _check(case, osf.intersection_vs_union, *case["args"])


def test_join_sorted(golden):
    for case in golden("score_functions.json")["join_sorted"]:
        _check(case, osf.join_sorted, *case["args"])


def test_compare_terms_values_and_trace(golden):
    cases = golden("compare_terms.json")["compare_terms"]
    for case in cases:
        trace = []

        def tracing(l, r):
            trace.append([l, r])
            return osf.intersection_vs_union(l, r)

        _check(case, oc.compare_terms, case["left"], case["right"], tracing)
        if "trace" in case and "value" in case:
            assert trace == case["trace"]
            pairs = oc.level_index_pairs(len(case["left"]), len(case["right"]))
            assert [[case["left"][a], case["right"][b]] for a, b in pairs] == case["trace"]


def test_gen_comp_value_and_flatten(golden):
    g = golden("compare_terms.json")
    for case in g["gen_comp_value"]:
        _check(case, oc.gen_comp_value, *case["args"])
    for case in g["flatten_list"]:
        _check(case, oc.flatten_list, *case["args"])


def test_categories_predicate(golden):
    g = golden("predicates.json")
    for case in g["categories_matching"]:
        left, right = case["left"], case["right"]
        pred = oc.categories_predicate(left[0], right[0])

        def run():
            return [i * len(right) + j for i, x in enumerate(left) for j, y in enumerate(right) if pred(x, y)]

        if "raises" in case:
            with pytest.raises(Exception) as err:
                run()
            assert type(err.value).__name__ == case["raises"]
        else:
            assert run() == case["kept"]


def test_blacklist_pairs(golden):
    g = golden("predicates.json")["flatten_mapping"]
    for key, pairs in g["pairs"].items():
        a, b = key.split("|")
        assert [list(t) for t in oc.blacklist_pairs(a, b, g["mapping"])] == pairs


def _run_grid(case):
    left, right = pd.DataFrame(case["left"]), pd.DataFrame(case["right"])
    exp = case["expected"]["gen_comparable"]
    if "raises" in exp:
        with pytest.raises(Exception) as err:
            oc.gen_comparable(left, right, case["whitelist"], case["blacklist"], **case["gen_kwargs"])
        assert type(err.value).__name__ == exp["raises"]
    else:
        got = oc.gen_comparable(left, right, case["whitelist"], case["blacklist"], **case["gen_kwargs"])
        assert list(got.index) == exp["index"]
        assert list(got.columns) == exp["columns"]
        assert list(got["MatchScore"]) == exp["scores"]  # bit-exact doubles
        for rec, want in zip(got.drop(columns=["MatchScore"]).to_dict(orient="records"), exp["records"]):
            assert rec == want
    if "compare" in case["expected"]:
        exp = case["expected"]["compare"]
        if "raises" in exp:
            with pytest.raises(Exception):
                oc.compare(left, right, case["whitelist"], case["blacklist"], **case["compare_kwargs"])
            return
        got = oc.compare(left, right, case["whitelist"], case["blacklist"], **case["compare_kwargs"])
        assert list(got["MatchScore"]) == exp["scores"]
        # the reference's tie order is unspecified (unstable quicksort): compare per score
        by_score_got, by_score_exp = {}, {}
        for lab, s in zip(got.index, got["MatchScore"]):
            by_score_got.setdefault(s, set()).add(lab)
        for lab, s in zip(exp["index"], exp["scores"]):
            by_score_exp.setdefault(s, set()).add(lab)
        assert by_score_got == by_score_exp


def test_pair_grids(golden):
    grids = golden("pair_grids.json")
    assert len(grids) >= 14
    for name, case in grids.items():
        _run_grid(case)


def test_c1_hap_pop_100(golden):
    case = golden("c1_hap_pop_100.json")
    assert len(case["expected"]["gen_comparable"]["index"]) == 90
    _run_grid(case)


def test_fuzzy_known_answers():
    """Hand-derived (fuzzy_match is parity unpinned, see oracle/__init__.py)."""
    assert osf.fuzzy_match("kitten", "sitting") == pytest.approx(8 / 13, abs=1e-12)
    assert osf.fuzzy_match("lewenstein", "levenshtein") == pytest.approx(18 / 21, abs=1e-12)
    assert osf.fuzzy_match("Dialyse", "Dialyse nach Entlassung") == pytest.approx(14 / 30, abs=1e-12)
    assert osf.fuzzy_match("this is a test", "THIS is a test!") == 1.0
    assert osf.fuzzy_match("abc", "") == 0.0
    assert osf.fuzzy_match("", "") == 0.0
    assert osf.fuzzy_match(["b", "A"], "a b") == 1.0
    assert osf.default_process("  Hello, World!_x ", underscore="keep") == "hello  world _x"
    assert osf.default_process("  Hello, World!_x ", underscore="blank") == "hello  world  x"
    assert osf.default_process("  Hello, World!_x ") == osf.default_process("  Hello, World!_x ", underscore=osf.UNDERSCORE_POLICY)
    assert osf.lcs_length("AGGTAB", "GXTXAYB") == 4


def test_rapidfuzz_published_answers():
    """Values rapidfuzz publishes for the 2.x line (fuzzy_match's arithmetic lives there, pinned ~=2.1.4 by the
    reference's pyproject.toml:17; the package is not installable offline, so these are the only numbers of
    its own that can anchor the restatement):
      README, "Simple Ratio":  fuzz.ratio("this is a test", "this is a test!") -> 96.55...  (= 100 * 28 / 29)
      rapidfuzz.distance.Indel docs:  distance("lewenstein", "levenshtein") -> 3, similarity -> 18,
        normalized_distance -> 0.14285714285714285, normalized_similarity -> 0.8571428571428572
      README, "Process"/QRatio default processor: upper case and punctuation do not matter."""
    assert osf.ratio("this is a test", "this is a test!") == pytest.approx(96.55172413793103, abs=1e-9)
    a, b = "lewenstein", "levenshtein"
    lcs = osf.lcs_length(a, b)
    assert len(a) + len(b) - 2 * lcs == 3 and lcs * 2 == 18
    assert osf.indel_ratio_from_lcs(len(a), len(b), lcs) == pytest.approx(0.8571428571428572, abs=1e-15)
    assert 1.0 - osf.indel_ratio_from_lcs(len(a), len(b), lcs) == pytest.approx(0.14285714285714285, abs=1e-15)
    assert osf.q_ratio("this is a test", "THIS is a test!") == 100.0


def test_default_process_two_implementations_agree():
    """The product restates default_process as a regular expression, the oracle per code point with
    str.isalnum: compared here on strings with "_", punctuation, upper case, non-ASCII letters and digits,
    under BOTH readings of the underscore (rapidfuzz 2.1's C++ and pure-Python implementations differ there)."""
    import random

    from napkon_string_matching_amd.compare import score_functions as sf

    rng = random.Random(5)
    pool = list("abcXYZ019 _-.,;:!?()[]/\\'\"\t\n") + list("äöüÄÖÜßéèñçøÅ") + list("αβγДЖ中文٣४①²½") + ["\u00a0", "\u2003", "İ", "ǅ"]
    samples = ["", "_", "__a__", "a_b-c", "  Hello, World!_x ", "Größe_(cm)", "İstanbul_2", "x\u00a0y", "a\tb_c\n"]
    samples += ["".join(rng.choice(pool) for _ in range(rng.randint(0, 24))) for _ in range(3000)]
    for policy in ("blank", "keep"):
        for text in samples:
            assert sf.default_process(text, underscore=policy) == osf.default_process(text, underscore=policy), (policy, text)
    assert sf.UNDERSCORE_POLICY == osf.UNDERSCORE_POLICY
    assert sf.default_process("a_b") == ("a b" if sf.UNDERSCORE_POLICY == "blank" else "a_b")
    with pytest.raises(ValueError):
        sf.default_process("x", underscore="maybe")
