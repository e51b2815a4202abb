"""The oracle's gen_comparable against THE REFERENCE ITSELF on random cohort frames (build container only:
needs /root/reference; the reference is imported with the inert rapidfuzz / nltk stand-ins of
tests/golden/make_golden.py, so only intersection_vs_union is exercised -- fuzzy_match stays unpinned).
The same generator drives tools/fuzz_api.py, which compares the GPU package with the oracle."""
import sys
from pathlib import Path

import pytest

REFERENCE = Path("/root/reference")
pytestmark = pytest.mark.skipif(not REFERENCE.exists(), reason="the reference checkout only exists in the build container")


def test_oracle_gen_comparable_matches_reference():
    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    import make_golden

    make_golden.install_stand_ins()
    from napkon_string_matching.types.mapping import Mapping
    from napkon_string_matching.types.questionnaire import Questionnaire

    from oracle import compare as oc
    from support import random_frames as rf

    outcomes = {"frames": 0, "rows": 0}
    for seed in range(1, 1501):
        left, right, wl, bl, kw, _kinds = rf.case(seed, score_funcs=("intersection_vs_union",), sizes=(0, 1, 4, 4, 12, 12, 25, 25))

        def run(fn):
            try:
                return fn(), None
            except Exception as exc:  # the exception TYPE is part of the contract
                return None, exc

        want, want_exc = run(lambda: Questionnaire(left.copy()).gen_comparable(
            Questionnaire(right.copy()), Mapping(data=wl), Mapping(data=bl), **kw).dataframe())
        got, got_exc = run(lambda: oc.gen_comparable(left.copy(), right.copy(), wl, bl, **kw))
        assert (want_exc is None) == (got_exc is None) and type(want_exc) is type(got_exc), (
            f"seed {seed}: reference {want_exc!r} / oracle {got_exc!r}")
        if want_exc is not None:
            outcomes[type(want_exc).__name__] = outcomes.get(type(want_exc).__name__, 0) + 1
            continue
        problem = rf.frames_differ(got, want, 0.0)
        assert problem is None, f"seed {seed}: {problem}"
        outcomes["frames"] += 1
        outcomes["rows"] += len(want)
    assert outcomes["frames"] > 400 and outcomes["rows"] > 4000 and outcomes.get("KeyError", 0) > 20, outcomes
    print(outcomes)


def test_oracle_compare_matches_reference(tmp_path):
    """``compare``: score at ``cache_threshold or score_threshold``, keep ``>= score_threshold``, order by
    score descending (the order of equal scores is the reference's quicksort's business: not compared)."""
    import random

    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    import make_golden

    make_golden.install_stand_ins()
    from napkon_string_matching.types.mapping import Mapping
    from napkon_string_matching.types.questionnaire import Questionnaire

    from oracle import compare as oc
    from support import random_frames as rf

    frames = rows = 0
    for seed in range(2001, 2401):
        left, right, wl, bl, kw, _kinds = rf.case(seed, score_funcs=("intersection_vs_union",), sizes=(1, 4, 12, 25))
        rng = random.Random(seed)
        kw["cache_threshold"] = rng.choice([None, 0.05, 0.3, 0.6])
        ref_kw = dict(kw, cached=False, cache_dir=tmp_path / f"cache{seed}")
        column = ref_kw.pop("compare_column")

        def run(fn):
            try:
                return fn(), None
            except Exception as exc:
                return None, exc

        want, want_exc = run(lambda: Questionnaire(left.copy()).compare(
            Questionnaire(right.copy()), Mapping(data=wl), Mapping(data=bl), column, **ref_kw).dataframe())
        got, got_exc = run(lambda: oc.compare(left.copy(), right.copy(), wl, bl, **kw))
        assert (want_exc is None) == (got_exc is None) and type(want_exc) is type(got_exc), (
            f"seed {seed}: reference {want_exc!r} / oracle {got_exc!r}")
        if want_exc is not None:
            continue
        scores = list(want["MatchScore"])
        assert all(a >= b for a, b in zip(scores, scores[1:])), f"seed {seed}: reference order"
        canon = want.iloc[sorted(range(len(want)), key=lambda k: (-scores[k], want.index[k]))]
        problem = rf.frames_differ(got, canon, 0.0)
        assert problem is None, f"seed {seed}: {problem}"
        frames += 1
        rows += len(want)
    assert frames > 100 and rows > 1000, (frames, rows)
