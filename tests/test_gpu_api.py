"""The reference-shaped host API (score_func plugins, ComparableData, Matcher) driven through the
HIP path, against (a) the golden fixtures produced by the reference itself and (b) the oracle."""
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _check(case, fn, *args):
    if "raises" in case:
        with pytest.raises(Exception) as err:
            fn(*args)
        assert type(err.value).__name__ == case["raises"]
    else:
        assert fn(*args) == case["value"]


def test_plugins_by_name():
    from napkon_string_matching_amd.compare import score_functions as sf

    assert getattr(sf, "intersection_vs_union") is sf.intersection_vs_union
    assert getattr(sf, "fuzzy_match") is sf.fuzzy_match
    with pytest.raises(AttributeError):
        getattr(sf, "no_such_score_func")


def test_intersection_vs_union_golden(golden):
    from napkon_string_matching_amd.compare.score_functions import intersection_vs_union, join_sorted

    g = golden("score_functions.json")
    for case in g["intersection_vs_union"]:
        _check(case, intersection_vs_union, *case["args"])
    for case in g["join_sorted"]:
        _check(case, join_sorted, *case["args"])


def test_fuzzy_match_scalar_against_oracle(monkeypatch):
    """Scalar plugin calls and one batched grid against the oracle, under both readings of "_" in
    default_process (the product restates it as a regex, the oracle per code point)."""
    import random

    from napkon_string_matching_amd.compare import score_functions as sf
    from oracle import score_functions as osf

    cases = [("kitten", "sitting"), ("Dialyse", "Dialyse nach Entlassung"), ("this is a test", "THIS is a test!"),
             ("abc", ""), ("", ""), (["b", "A"], "a b"), (["Zeta", "alpha", "Beta"], ["beta", "ALPHA"]),
             ("a_b-c", "a b c"), ("Größe_(cm)", "groesse cm"), ("x_1", "X 1"), ("__", "_")]
    rng = random.Random(11)
    pool = list("abcdeXYZ0123 _-.,;!?()/") + list("äöüÄÖÜßéñ") + list("αβД中٣")
    texts = ["".join(rng.choice(pool) for _ in range(rng.randint(0, 30))) for _ in range(60)]
    for policy in ("blank", "keep"):
        monkeypatch.setattr(sf, "UNDERSCORE_POLICY", policy)
        monkeypatch.setattr(osf, "UNDERSCORE_POLICY", policy)
        for a, b in cases:
            assert abs(sf.fuzzy_match(a, b) - osf.fuzzy_match(a, b)) <= 1e-6, (policy, a, b)
        got = {(i, j): sc for sc, i, j in sf.fuzzy_match.raw_grid(texts, texts[::-1], float("-inf")).as_tuples()}
        assert len(got) == len(texts) ** 2
        for i, a in enumerate(texts):
            for j, b in enumerate(texts[::-1]):
                want = osf.fuzzy_match(a, b)
                assert got[i, j] == want and abs(got[i, j] - want) <= 1e-6, (policy, a, b)
    monkeypatch.setattr(sf, "UNDERSCORE_POLICY", "keep")
    monkeypatch.setattr(osf, "UNDERSCORE_POLICY", "keep")
    assert sf.fuzzy_match("a_b", "a b") < 1.0
    monkeypatch.setattr(sf, "UNDERSCORE_POLICY", "blank")
    assert sf.fuzzy_match("a_b", "a b") == 1.0
    # rapidfuzz's own published numbers (see tests/test_oracle_golden.py::test_rapidfuzz_published_answers),
    # through the RAW kernel on the unprocessed strings
    from napkon_string_matching_amd import grid, tables
    import torch

    lt, rt = tables.encode_strings(["this is a test", "lewenstein"], ["this is a test!", "levenshtein"], torch.device("cuda:0"))
    raw = {(i, j): sc for sc, i, j in grid.indel_raw_grid(lt, rt, -1.0).as_tuples()}
    assert abs(raw[0, 0] * 100 - 96.55172413793103) <= 1e-9 and abs(raw[1, 1] - 0.8571428571428572) <= 1e-15


def test_compare_terms_golden(golden):
    from napkon_string_matching_amd.compare.score_functions import intersection_vs_union
    from napkon_string_matching_amd.types.comparable_data import ComparableData

    cases = golden("compare_terms.json")
    for case in cases["compare_terms"]:
        _check(case, ComparableData.compare_terms, case["left"], case["right"], intersection_vs_union)
    for case in cases["gen_comp_value"]:
        _check(case, ComparableData.gen_comp_value, *case["args"])


def _run_case(case):
    from napkon_string_matching_amd.types.questionnaire import Questionnaire

    left, right = Questionnaire(pd.DataFrame(case["left"])), Questionnaire(pd.DataFrame(case["right"]))
    exp = case["expected"]["gen_comparable"]
    if "raises" in exp:
        with pytest.raises(Exception) as err:
            left.gen_comparable(right, case["whitelist"], case["blacklist"], **case["gen_kwargs"])
        assert type(err.value).__name__ == exp["raises"]
    else:
        got = left.gen_comparable(right, case["whitelist"], case["blacklist"], **case["gen_kwargs"]).dataframe()
        assert list(got.index) == exp["index"]
        assert list(got.columns) == exp["columns"]
        assert list(got["MatchScore"]) == exp["scores"]  # bit-exact doubles
        assert got.drop(columns=["MatchScore"]).to_dict(orient="records") == exp["records"]
    if "compare" in case["expected"]:
        exp = case["expected"]["compare"]
        if "raises" in exp:
            with pytest.raises(Exception):
                left.compare(right, case["whitelist"], case["blacklist"], **case["compare_kwargs"])
            return
        comp = left.compare(right, case["whitelist"], case["blacklist"], **case["compare_kwargs"])
        assert (comp.left_name, comp.right_name) == (exp["left_name"], exp["right_name"])
        assert list(comp.match_score) == exp["scores"]
        per_got, per_exp = {}, {}  # the reference's tie order is unspecified
        for lab, s in zip(comp.dataframe().index, comp.match_score):
            per_got.setdefault(s, set()).add(lab)
        for lab, s in zip(exp["index"], exp["scores"]):
            per_exp.setdefault(s, set()).add(lab)
        assert per_got == per_exp


def test_pair_grids_golden(golden):
    grids = golden("pair_grids.json")
    for name, case in grids.items():
        _run_case(case)


def test_c1_hap_pop_100_golden(golden):
    """BASELINE.json configs[0]: hap vs pop, 100 items each, Tokens, threshold 0.1."""
    _run_case(golden("c1_hap_pop_100.json"))


@pytest.mark.parametrize("score_func,column,thr", [("fuzzy_match", "Tokens", 0.3), ("fuzzy_match", "Variable", 0.3),
                                                    ("intersection_vs_union", "Variable", 0.2)])
def test_gen_comparable_against_oracle(golden, score_func, column, thr):
    """No reference output exists for fuzzy_match (rapidfuzz absent): compare with the oracle."""
    from napkon_string_matching_amd.types.questionnaire import Questionnaire
    from oracle import compare as oc

    case = golden("pair_grids.json")["rand_40x30_categories"]
    left, right = pd.DataFrame(case["left"]), pd.DataFrame(case["right"])
    kw = dict(score_func=score_func, compare_column=column, left_name="hap", right_name="suep",
              filter_categories=True, score_threshold=thr)
    want = oc.gen_comparable(left, right, {}, case["blacklist"], **kw)
    got = Questionnaire(left).gen_comparable(Questionnaire(right), {}, case["blacklist"], **kw).dataframe()
    assert list(got.index) == list(want.index) and len(want) > 3
    assert list(got.columns) == list(want.columns)
    assert np.allclose(got["MatchScore"].to_numpy(), want["MatchScore"].to_numpy(), rtol=0, atol=1e-6)
    if score_func == "intersection_vs_union":
        assert list(got["MatchScore"]) == list(want["MatchScore"])


@pytest.mark.parametrize("n_labels", [64, 70])
@pytest.mark.parametrize("score_func", ["intersection_vs_union", "fuzzy_match"])
def test_sixty_four_categories(golden, score_func, n_labels):
    """64 distinct category labels fill the mask; with list categories the empty-vs-empty rule then has
    no spare bit to ride on, and only one side may use the last label: the partition decision is joint."""
    from napkon_string_matching_amd.types.questionnaire import Questionnaire
    from oracle import compare as oc

    case = golden("pair_grids.json")["rand_40x30_categories"]
    left, right = pd.DataFrame(case["left"]), pd.DataFrame(case["right"])
    labels = [f"c{k}" for k in range(n_labels)]  # 70: more labels than mask bits, the predicate moves to the host
    left["Category"] = [[] if k % 7 == 0 else [labels[(3 * k) % 63], labels[(5 * k + 1) % 63]] for k in range(len(left))]
    right["Category"] = [[] if k % 5 == 0 else [labels[(3 * k + 6) % 63]] for k in range(len(right))]
    left.at[left.index[1], "Category"] = labels          # every label, the last ones only on this side
    kw = dict(score_func=score_func, compare_column="Tokens", left_name="hap", right_name="suep",
              filter_categories=True, score_threshold=0.15)
    want = oc.gen_comparable(left, right, {}, {}, **kw)
    got = Questionnaire(left).gen_comparable(Questionnaire(right), {}, {}, **kw).dataframe()
    assert list(got.index) == list(want.index) and len(want) > 3
    assert np.allclose(got["MatchScore"].to_numpy(), want["MatchScore"].to_numpy(), rtol=0, atol=1e-6)


def test_matcher_end_to_end(golden):
    """Matcher over three cohorts: pair enumeration, result keys, overrides; results vs the oracle."""
    from napkon_string_matching_amd import matching, synthetic
    from napkon_string_matching_amd.types.comparable import Comparable
    from napkon_string_matching_amd.types.mapping import Mapping
    from napkon_string_matching_amd.types.questionnaire import Questionnaire
    from oracle import compare as oc

    frames = {
        "suep": pd.DataFrame(synthetic.cohort_records("suep", 40, 3, vocab=30, max_entries=4, tokens_per_entry=2)),
        "hap": pd.DataFrame(synthetic.cohort_records("hap", 50, 1, vocab=30, max_entries=4, tokens_per_entry=2)),
        "Pop": pd.DataFrame(synthetic.cohort_records("pop", 45, 2, vocab=30, max_entries=4, tokens_per_entry=2)),
    }
    config = {
        "matching": {"score_threshold": 0.3, "cache_threshold": 0.2, "compare_column": "Tokens",
                     "score_func": "intersection_vs_union", "calculate_tokens": False, "filter_column": "Variable",
                     "filter_prefix": "gec_", "tokens": {"timeout": 30}, "variable_score_threshold": 0.6,
                     "filter_categories": True},
        "steps": ["variables", "questionnaires"],
    }
    blacklist = {"b": {"hap": [frames["hap"]["Identifier"][0]], "Pop": list(frames["Pop"]["Identifier"][:5])}}
    m = matching.match(config, write=False, questionnaires={k: Questionnaire(v) for k, v in frames.items()},
                       mappings_blacklist=Mapping(blacklist))
    assert sorted(m.results.results) == sorted(
        ["var_hap vs Pop", "var_hap vs suep", "var_Pop vs suep", "hap vs Pop", "hap vs suep", "Pop vs suep"])
    for key, comp in m.results.items():
        assert isinstance(comp, Comparable)
        is_var = key.startswith("var_")
        a, b = key[4:].split(" vs ") if is_var else key.split(" vs ")
        kw = {**config["matching"]}
        if is_var:
            kw.update(compare_column="Variable", score_threshold=0.6)
        want = oc.compare(frames[a], frames[b], {}, blacklist, left_name=a, right_name=b, **kw)
        assert list(comp.dataframe().index) == list(want.index), key
        assert list(comp.match_score) == list(want["MatchScore"]), key
        assert (comp.left_name, comp.right_name) == (a.title(), b.title())
    assert len(m.results["hap vs Pop"]) > 0
    analysis = m._analyse()
    assert set(analysis["hap vs Pop"]) == {"matched", "gecco"}


def test_matcher_gecco_step(golden):
    """``match_gecco_with_questionnaires``: GECCO items (scalar category, Variable := Identifier) against every
    cohort, keys ``gecco vs <name>``; results vs the oracle."""
    from napkon_string_matching_amd import matching, synthetic
    from napkon_string_matching_amd.types.questionnaire import GeccoDefinition, Questionnaire
    from oracle import compare as oc

    frames = {
        "hap": pd.DataFrame(synthetic.cohort_records("hap", 50, 1, vocab=30, max_entries=4, tokens_per_entry=2)),
        "pop": pd.DataFrame(synthetic.cohort_records("pop", 45, 2, vocab=30, max_entries=4, tokens_per_entry=2)),
    }
    gecco = pd.DataFrame(synthetic.cohort_records("gec", 30, 5, vocab=30, max_entries=4, tokens_per_entry=2))
    gecco["Category"] = [cats[0] for cats in gecco["Category"]]  # GECCO: one label per item
    gecco["Parameter"] = [f"p{k}" for k in range(len(gecco))]
    gecco["Choices"] = [None if k % 3 else "ja nein" for k in range(len(gecco))]
    gd = GeccoDefinition(gecco.copy())
    gd.add_terms()
    assert gd["Term"][0] == [gecco["Category"][0], "p0", "ja nein"] and gd["Term"][1] == [gecco["Category"][1], "p1"]
    config = {"matching": {"score_threshold": 0.2, "cache_threshold": None, "compare_column": "Tokens",
                           "score_func": "intersection_vs_union", "filter_categories": True,
                           "variable_score_threshold": 0.6}, "steps": ["gecco"]}
    m = matching.match(config, write=False, questionnaires={k: Questionnaire(v) for k, v in frames.items()},
                       gecco=GeccoDefinition(gecco.copy()))
    assert sorted(m.results.results) == ["gecco vs hap", "gecco vs pop"]
    left = gecco.copy()
    left["Variable"] = left["Identifier"]
    for name, frame in frames.items():
        want = oc.compare(left, frame, {}, {}, left_name="gecco", right_name=name, **config["matching"])
        got = m.results[f"gecco vs {name}"].dataframe()
        assert list(got.index) == list(want.index) and len(want) > 0
        assert list(got["MatchScore"]) == list(want["MatchScore"])
        assert list(got["GeccoVariable"]) == list(want["GeccoVariable"]) == list(got["GeccoIdentifier"])


def test_mesh_get_matches_and_add_tokens(golden):
    """Row f1: MeshProvider.get_matches / MatchPreparator.add_tokens through the RAW fuzzy grid."""
    import random

    from napkon_string_matching_amd.prepare.match_preparator import MatchPreparator
    from napkon_string_matching_amd.terminology.mesh import MeshProvider, TerminologyProvider
    from napkon_string_matching_amd.types.questionnaire import Questionnaire
    from oracle import terminology as oterm

    refs = pd.DataFrame(golden("mesh_references.json")["references"])
    provider = MeshProvider(None, synonyms=refs)
    # the reference's own smoke test (tests/terminology/test_mesh.py:26-33) passes a str
    results = provider.get_matches("Dialyse nach Entlassung")
    assert results and "Dialyse" in results[0][1] and results[0][2] > 0
    want = oterm.get_matches(list(refs["Id"]), list(refs["Term"]), "Dialyse nach Entlassung", 0.1)
    assert [(a, b) for a, b, _ in results] == [(a, b) for a, b, _ in want]
    assert all(abs(x[2] - y[2]) <= 1e-6 for x, y in zip(results, want))

    rng = random.Random(4)
    words = ["dialyse", "niere", "herz", "lunge", "fieber", "husten", "impfung", "therapie", "nach", "vor", "bei"]
    syn = pd.DataFrame({"Id": [f"D{rng.randrange(25):03d}" for _ in range(120)],
                        "Term": [" ".join(rng.sample(words, rng.randint(1, 4))).title() for _ in range(120)]})
    items = [rng.sample(words, rng.randint(1, 5)) for _ in range(60)]
    provider = MeshProvider(None, synonyms=syn)
    got = provider.get_matches_batch(items, 0.55)
    for term, rows in zip(items, got):
        want = oterm.get_matches(list(syn["Id"]), list(syn["Term"]), term, 0.55)
        assert [(a, b) for a, b, _ in rows] == [(a, b) for a, b, _ in want]
        assert all(abs(x[2] - y[2]) <= 1e-6 for x, y in zip(rows, want))
        assert len({a for a, _, _ in rows}) == len(rows)  # one row per Id
    cs = Questionnaire(pd.DataFrame({"Identifier": [f"i{k}" for k in range(len(items))], "Term": items}))
    MatchPreparator(None, TerminologyProvider(None, [provider])).add_tokens(cs, score_threshold=0.55)
    assert list(cs["TokenMatch"])[0] == (got[0] if got[0] else None)
    first = next(k for k, rows in enumerate(got) if rows)
    assert cs["TokenIds"][first] == tuple(a for a, _, _ in got[first])
    assert cs["Tokens"][first] == tuple(b for _, b, _ in got[first])


def test_compare_cache_stable_key(golden, tmp_path, monkeypatch):
    """Row f2: the compare cache hits on identical content, misses on a different score function."""
    from napkon_string_matching_amd.types.comparable_data import ComparableData
    from napkon_string_matching_amd.types.questionnaire import Questionnaire

    case = golden("pair_grids.json")["rand_30x40"]
    left, right = Questionnaire(pd.DataFrame(case["left"])), Questionnaire(pd.DataFrame(case["right"]))
    kw = dict(case["compare_kwargs"])
    first = left.compare(right, case["whitelist"], case["blacklist"], cache_dir=tmp_path, **kw)
    files = list(tmp_path.glob("compared__score_*.json"))
    assert len(files) == 1
    cached_rows = pd.DataFrame(__import__("json").loads(files[0].read_text())["data"])
    assert (cached_rows["MatchScore"] >= kw["cache_threshold"]).all() and len(cached_rows) >= len(first)

    def boom(*_a, **_k):
        raise AssertionError("gen_comparable must not run on a cache hit")

    monkeypatch.setattr(ComparableData, "gen_comparable", boom)
    again = Questionnaire(pd.DataFrame(case["left"])).compare(
        Questionnaire(pd.DataFrame(case["right"])), case["whitelist"], case["blacklist"], cache_dir=tmp_path, **kw)
    assert list(again.match_score) == list(first.match_score)
    assert list(again.dataframe()["HapIdentifier"]) == list(first.dataframe()["HapIdentifier"])
    with pytest.raises(AssertionError):  # another score function is another key
        left.compare(right, case["whitelist"], case["blacklist"], cache_dir=tmp_path, **{**kw, "score_func": "fuzzy_match"})


_DIST_WORKER = r'''
import json, os, sys
sys.path.insert(0, {pkg!r}); sys.path.insert(0, {root!r})
import pandas as pd, torch, torch.distributed as dist
from napkon_string_matching_amd.types.questionnaire import Questionnaire
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
torch.cuda.set_device(0)
case = json.load(open({fixture!r}))["rand_40x30_categories"]
left, right = Questionnaire(pd.DataFrame(case["left"])), Questionnaire(pd.DataFrame(case["right"]))
out = {{}}
from napkon_string_matching_amd import distributed
seen = []
real = distributed.all_gather_pending
distributed.all_gather_pending = lambda *a, **k: (seen.append("device"), real(*a, **k))[1]
for func in ("intersection_vs_union", "fuzzy_match"):
    for label, blacklist in (("", case["blacklist"]), ("/no blacklist", None)):
        kw = dict(case["compare_kwargs"], score_func=func, score_threshold=0.2, cache_threshold=None)
        comp = left.compare(right, case["whitelist"], blacklist, **kw)
        out[func + label] = [list(map(int, comp.dataframe().index)), [float(v) for v in comp.match_score]]
out["device_resident_exchanges"] = len(seen)
json.dump(out, open({out!r} + str(dist.get_rank()), "w"))
dist.destroy_process_group()
'''


def test_sharded_compare_world2(golden, tmp_path):
    """N > 1 path end to end: two ranks (gloo, sharing this GPU) each score their block of left rows,
    all-gather the hits and return the same Comparable as a single process -- with a blacklist (the hits are filtered
    on the host first, then exchanged) and without one (the hit buffers go to the all-gather as they are on the device:
    ``distributed.all_gather_pending``, the exchange bench.py times)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from pathlib import Path

    from napkon_string_matching_amd.types.questionnaire import Questionnaire

    root = Path(__file__).resolve().parent.parent
    script = tmp_path / "worker.py"
    script.write_text(_DIST_WORKER.format(pkg=str(root / "napkon-string-matching_amd"), root=str(root),
                                          fixture=str(root / "tests" / "golden" / "pair_grids.json"),
                                          out=str(tmp_path / "out")))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=300)
        assert p.returncode == 0, out.decode()[-2000:]
    case = golden("pair_grids.json")["rand_40x30_categories"]
    left, right = Questionnaire(pd.DataFrame(case["left"])), Questionnaire(pd.DataFrame(case["right"]))
    for func in ("intersection_vs_union", "fuzzy_match"):
        for label, blacklist in (("", case["blacklist"]), ("/no blacklist", None)):
            kw = dict(case["compare_kwargs"], score_func=func, score_threshold=0.2, cache_threshold=None)
            single = left.compare(right, case["whitelist"], blacklist, **kw)
            want = [list(map(int, single.dataframe().index)), [float(v) for v in single.match_score]]
            assert len(want[0]) > 5
            for rank in range(2):
                got = json.load(open(str(tmp_path / "out") + str(rank)))
                assert got[func + label] == want, (func, label, rank)
                assert got["device_resident_exchanges"] == 2  # the two runs without a blacklist


def test_integration_md_level2_stub():
    """INTEGRATION.md's level-2 binding, executed AS WRITTEN (raw ctypes on the C ABI: nsm_build_set_table +
    nsm_jaccard_levels_grid, nsm_build_str_table + nsm_build_level_items + nsm_indel_levels_grid; nothing of the
    package's tables.py), against the oracle on a C5-shaped cohort pair with the list x list category predicate."""
    import re

    from napkon_string_matching_amd import _lib, synthetic
    from oracle import native

    text = (ROOT / "INTEGRATION.md").read_text(encoding="utf-8")
    level2 = text[text.index("## Level 2"):text.index("Entry points and what they replace")]
    blocks = re.findall(r"```python\n(.*?)```", level2, flags=re.S)
    assert len(blocks) == 2 and "napkon_string_matching_amd" not in "".join(blocks)
    ns = {}
    exec(compile("\n".join(blocks).replace('"libnsm_hip.so"', repr(str(_lib.LIB_PATH))), "INTEGRATION.md", "exec"), ns)

    hap = synthetic.c5_cohort(700, 21)
    pop = synthetic.c5_cohort(900, 22, plant_from=hap)
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY
    hap["cat"][:5] = 0  # items without a category: "both empty" matches (types/comparable_data.py:467-470)
    pop["cat"][:7] = 0
    canon = lambda s, i, j: sorted(zip(s.tolist(), i.tolist(), j.tolist()), key=lambda h: (-h[0], h[1], h[2]))

    # ---- intersection_vs_union
    lt, keep_l = ns["build_side"](hap["ids"], hap["nlev"], hap["plen"], hap["cat"], 0)
    rt, keep_r = ns["build_side"](pop["ids"], pop["nlev"], pop["plen"], pop["cat"], 1)
    got = canon(*ns["score_grid"](lt, rt, 0.55))
    ids = lambda c: [[[int(t[1:]) for t in level] for level in item] for item in synthetic.c5_level_token_lists(c)]
    want = native.levels(False, ids(hap), ids(pop), 0.55, hap["cat"], pop["cat"], mode, cap=1 << 16)
    assert got == want and len(want) > 5

    # ---- fuzzy_match
    sides = []
    for c in (hap, pop):
        codes, lengths, first, nlev = synthetic.c5_level_codes(c)
        sides.append(ns["build_fuzzy_side"](codes, lengths, first, nlev, c["cat"], len(synthetic.C5_ALPHABET)))
    got = canon(*ns["score_fuzzy_grid"](sides[0][0], sides[1][0], 0.6))
    cps = lambda c: [[[ord(ch) for ch in " ".join(level)] for level in item] for item in synthetic.c5_level_token_lists(c)]
    want = native.levels(True, cps(hap), cps(pop), 0.6, hap["cat"], pop["cat"], mode, cap=1 << 16)
    assert got == want and len(want) > 5


def test_analyse_golden(golden):
    """Row f4: ``Matcher._analyse`` (matcher.py:290-312).  The fixture holds what the REFERENCE's ``_analyse`` returned
    for results the reference's own ``compare`` produced (tests/golden/make_golden.py section 6: ``gec_``-prefixed
    variables on both sides, shared variable names, a cohort pair without hits); the package must return the same
    strings from the results its GPU path computes on the same cohorts."""
    from napkon_string_matching_amd.matcher import Matcher
    from napkon_string_matching_amd.types.comparable import ComparisonResults
    from napkon_string_matching_amd.types.questionnaire import Questionnaire

    case = golden("analyse.json")
    results = {}
    for a, b, thr in case["steps"]:
        left, right = Questionnaire(pd.DataFrame(case["cohorts"][a])), Questionnaire(pd.DataFrame(case["cohorts"][b]))
        comp = left.compare(right, None, None, left_name=a, right_name=b, cached=False,
                            **{**case["compare_kwargs"], "score_threshold": thr})
        results[f"{a} vs {b}"] = comp
        assert len(comp) == case["rows"][f"{a} vs {b}"]
    m = Matcher(None, {"matching": case["compare_kwargs"]})
    m.results = ComparisonResults(results)
    got = m._analyse()
    assert got == case["analysis"] and "pop vs hap" not in got  # an empty result is skipped (:299-300)
    assert list(got) == list(case["analysis"])  # same order as the results were stored
