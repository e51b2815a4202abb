"""Operands beyond the register-resident kernels (round-2 verdict: "a drop-in must not die on them"): items of more than
64 distinct tokens, strings of more than 512 code units, grids of more than 255 distinct code units -- RAW plugin calls
and the levels loop, through the package's API (which routes only the wide items through csrc/any_grids.hip) against the
oracle.  Bit-exact."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tuples(hits):
    return hits.as_tuples()


def _words(rng, n, lo=2, hi=9, alphabet="abcdefghijklmnopqrstuvwxyz"):
    return ["".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(n)]


def test_raw_jaccard_100_token_items():
    """intersection_vs_union on items of up to 120 distinct tokens (score_functions.py:10-13 has no size limit)."""
    from napkon_string_matching_amd.compare import score_functions as sf
    from oracle import native

    rng = random.Random(5)
    vocab = [f"t{k}" for k in range(400)]
    item = lambda big: rng.sample(vocab, rng.randint(65, 120) if big else rng.randint(1, 30))
    left = [item(k % 7 == 0) for k in range(90)]
    right = [item(k % 5 == 0) for k in range(150)]
    right[10] = list(left[0])           # a wide near-duplicate pair
    right[11] = left[7][:100] + ["zz"]
    left[3] = left[3] + left[3][:2]     # repeated tokens count once
    ids = {t: k for k, t in enumerate(vocab + ["zz"])}
    csr = lambda rows: native.csr([sorted({ids[t] for t in r}) for r in rows])
    for thr in (0.0, 0.2, 0.6):
        want = native.jaccard_raw(csr(left), csr(right), thr, cap=1 << 15)
        got = _tuples(sf.intersection_vs_union.raw_grid(left, right, thr))
        assert got == want and len(want) > 0
    assert sf.intersection_vs_union(left[0], right[10]) == 1.0
    with pytest.raises(ZeroDivisionError):
        sf.intersection_vs_union.raw_grid(left + [[]], right + [[]], 0.5)


def test_raw_fuzzy_1500_unit_strings_and_400_symbols():
    """fuzzy_match on strings of up to 1500 code units, and on a grid whose strings use 400 distinct code units
    (rapidfuzz has neither limit, score_functions.py:27)."""
    from napkon_string_matching_amd.compare import score_functions as sf
    from oracle import native

    rng = random.Random(6)
    greek = [chr(c) for c in range(0x3B1, 0x3B1 + 24)]
    cjk = [chr(c) for c in range(0x4E00, 0x4E00 + 380)]  # letters for str.isalnum: default_process keeps them
    sentence = lambda n_words, abc: " ".join(_words(rng, n_words, alphabet=abc))
    latin = "abcdefghijklmnopqrstuvwxyz"
    left = [sentence(rng.randint(3, 12), latin) for _ in range(40)]
    right = [sentence(rng.randint(3, 12), latin) for _ in range(70)]
    left[1] = sentence(230, latin)[:1500]                       # ~1500 code units
    left[2] = sentence(90, latin)[:600]
    right[3] = left[1][:700] + sentence(100, latin)[:650]
    right[4] = left[2][:-5] + "xyzzy"
    right[5] = "".join(rng.choice(cjk) for _ in range(300))     # 380 more symbols in the grid
    left[6] = right[5][:200] + "".join(rng.choice(greek) for _ in range(40))
    cps = lambda strings: native.csr([[ord(c) for c in sf.fuzzy_operand(s)] for s in strings])
    for thr in (0.0, 0.3, 0.7):
        want = native.indel_raw(cps(left), cps(right), thr, cap=1 << 13)
        got = _tuples(sf.fuzzy_match.raw_grid(left, right, thr))
        assert got == want and len(want) > 0
    assert sf.fuzzy_match(left[1], left[1]) == 1.0 and 0.5 < sf.fuzzy_match(left[1], right[3]) < 1.0


def test_levels_wide_items_through_gen_comparable():
    """The Matcher's loop (gen_comparable -> compare_terms) with a few wide items in otherwise ordinary cohorts: a Tokens
    list of 100 distinct tokens, and a Term whose joined levels exceed 512 code units -- against the oracle's
    restatement of gen_comparable, for both score functions."""
    import pandas as pd

    from napkon_string_matching_amd import synthetic
    from napkon_string_matching_amd.types.questionnaire import Questionnaire
    from oracle import compare as oc

    rng = random.Random(7)
    frames = {}
    for name, n, seed in (("hap", 60, 1), ("pop", 75, 2)):
        rows = synthetic.cohort_records(name, n, seed, vocab=60, max_entries=4, tokens_per_entry=2)
        frames[name] = rows
    long_entry = lambda k: " ".join(f"w{rng.randrange(500)}x{k}" for _ in range(35))
    frames["hap"][4]["Tokens"] = [long_entry(1), long_entry(2), long_entry(3)]          # ~100 distinct tokens, 3 levels
    frames["pop"][9]["Tokens"] = frames["hap"][4]["Tokens"][1:] + ["extra tokens here"]
    frames["pop"][12]["Tokens"] = [long_entry(4), long_entry(5)]
    frames["hap"][7]["Term"] = [" ".join(_words(rng, 70)), " ".join(_words(rng, 60)), "kurz"]   # joined levels > 512 units
    frames["pop"][20]["Term"] = [frames["hap"][7]["Term"][0], frames["hap"][7]["Term"][1], "kurz und gut"]
    hap, pop = pd.DataFrame(frames["hap"]), pd.DataFrame(frames["pop"])
    for func, column, thr in (("intersection_vs_union", "Tokens", 0.15), ("fuzzy_match", "Term", 0.45),
                              ("fuzzy_match", "Tokens", 0.4)):
        kw = dict(score_func=func, compare_column=column, score_threshold=thr, left_name="hap", right_name="pop",
                  filter_categories=True)
        got = Questionnaire(hap).gen_comparable(Questionnaire(pop), None, None, **kw).dataframe()
        want = oc.gen_comparable(hap, pop, {}, {}, **kw)
        assert list(got.index) == list(want.index), (func, column)
        assert list(got["MatchScore"]) == list(want["MatchScore"]), (func, column)
        assert len(want) > 3


def test_irregular_levels_through_gen_comparable():
    """Items the suffix-nested fast layouts cannot hold (round-3 verdict: the build raised NotImplementedError for the whole
    compare()):  (1) a CONTEXT-DEPENDENT tokenizer -- the reference re-tokenises every suffix of the compare value
    (types/comparable_data.py:283-299: level l = tokenize(" ".join(items[-(l + 1):]))), so a tokenizer that treats the
    START of its text specially (here: a sentence-initial capital is lower-cased, as truecasing tokenizers do) yields
    "ca" at the level where the entry "Ca. ..." comes first and "Ca." at the deeper ones -- level l is then not
    "level l - 1 plus more";  (2) compare_column
    "Variable" with names of 100 characters: one level per character (:283-285, :567-574), 100 levels.  Only those items
    leave the fast path; the result must be the oracle's gen_comparable, for both score functions."""
    import pandas as pd

    from napkon_string_matching_amd import synthetic
    from napkon_string_matching_amd.types import comparable_data as cd
    from napkon_string_matching_amd.types.questionnaire import Questionnaire
    from oracle import compare as oc

    def punkt_like(text: str):
        """Whitespace split; the first word of the text is true-cased and loses a trailing period."""
        words = text.split()
        if words and words[0][:1].isupper():
            words[0] = words[0].lower().rstrip(".")
        return words

    rng = random.Random(11)
    frames = {}
    for name, n, seed in (("hap", 70, 3), ("pop", 90, 4)):
        frames[name] = synthetic.cohort_records(name, n, seed, vocab=50, max_entries=4, tokens_per_entry=2, min_entries=2)
    # an entry "Ca. ..." in the middle of the value: the level at which it comes first tokenises it as "ca", the deeper
    # levels (where text precedes it) as "Ca."
    for name, rows in frames.items():
        for k in range(3, len(rows), 9):
            toks = rows[k]["Tokens"]
            toks[len(toks) // 2] = "Ca. " + toks[len(toks) // 2]
    frames["pop"][5]["Tokens"] = list(frames["hap"][3]["Tokens"])
    for name, rows in frames.items():  # (2) Variable names of up to 100 characters
        for k, row in enumerate(rows):
            row["Variable"] = "".join(rng.choice("abcdefgh_") for _ in range(100 if k % 4 == 0 else rng.randint(3, 40)))
    frames["pop"][8]["Variable"] = frames["hap"][0]["Variable"][:-3] + "xyz"
    frames["pop"][16]["Variable"] = frames["hap"][4]["Variable"]
    hap, pop = pd.DataFrame(frames["hap"]), pd.DataFrame(frames["pop"])
    tok = cd.Tokenizer(word_tokenize=punkt_like)
    old = cd.ComparableData.tokenizer
    cd.ComparableData.tokenizer = tok
    try:
        levels = cd.ComparableData.gen_comp_value(frames["hap"][3]["Tokens"])
        assert not all(set(a) <= set(b) for a, b in zip(levels, levels[1:])), "the tokenizer should break the nesting"
        for func, column, thr in (("intersection_vs_union", "Tokens", 0.15), ("fuzzy_match", "Tokens", 0.4),
                                  ("intersection_vs_union", "Variable", 0.55), ("fuzzy_match", "Variable", 0.6)):
            kw = dict(score_func=func, compare_column=column, score_threshold=thr, left_name="hap", right_name="pop",
                      filter_categories=column == "Tokens")
            got = Questionnaire(hap).gen_comparable(Questionnaire(pop), None, None, **kw).dataframe()
            want = oc.gen_comparable(hap, pop, {}, {}, tokenizer=dict(word_tokenize=punkt_like), **kw)
            assert list(got.index) == list(want.index), (func, column)
            assert list(got["MatchScore"]) == list(want["MatchScore"]), (func, column)
            assert len(want) > 3, (func, column, len(want))
    finally:
        cd.ComparableData.tokenizer = old


def test_any_grid_caps_fail_loudly():
    from napkon_string_matching_amd import wide

    import torch

    dev = torch.device("cuda:0")
    with pytest.raises(NotImplementedError):
        wide.indel_any_grid([["a" * 5000]], [["a"]], 0.5, device=dev)
    with pytest.raises(NotImplementedError):
        wide.indel_any_grid([["".join(chr(0x4E00 + k) for k in range(1100))]], [["a"]], 0.5, device=dev)
