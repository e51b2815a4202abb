#!/usr/bin/env python3
"""Randomised differential test at the reference's interface: ComparableData.gen_comparable (this
package, GPU) against oracle.compare.gen_comparable (CPU restatement of comparable_data.py:133-246) on
random cohort frames -- NaN compare values, empty token lists, duplicate identifiers, list / scalar
categories, whitelist and blacklist mappings, both score functions, three compare columns.  Same
exception type or the same frame (pair labels, columns, cell values, scores) is required.

    python tools/fuzz_api.py [--seconds 120] [--seed 0]
"""
import argparse
import json
import random
import sys
import time
import traceback
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "napkon-string-matching_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()


    from napkon_string_matching_amd.types.questionnaire import Questionnaire
    from oracle import compare as oc

    sys.path.insert(0, str(ROOT / "tests"))
    from support import random_frames as rf

    t_end = time.time() + args.seconds
    rnd = args.seed
    stats = {"rounds": 0, "frames": 0, "raised": {}, "rows": 0}
    next_report = time.time() + 60.0
    while time.time() < t_end:
        rnd += 1
        if time.time() >= next_report:  # a long silent GPU job is taken to be hung
            print(json.dumps({"progress": stats, "round_seed": rnd}), flush=True)
            next_report += 60.0
        left, right, wl, bl, kw, kinds = rf.case(rnd)
        stats["rounds"] += 1

        def run(fn):
            try:
                return fn(), None
            except Exception as exc:  # the reference's own per-pair errors are part of the contract
                return None, exc

        if rnd % 4 == 1:  # inside item_memo(): pooled level encoding, shared vocabulary and category bits
            from napkon_string_matching_amd.types.comparable_data import ComparableData

            def pooled():
                ql, qr = Questionnaire(left.copy()), Questionnaire(right.copy())
                with ComparableData.item_memo():
                    first = ql.gen_comparable(qr, wl, bl, **kw).dataframe()
                    again = ql.gen_comparable(qr, wl, bl, **kw).dataframe()  # every item now comes from the pool
                assert rf.frames_differ(again, first, 0.0) is None
                return first

            want, want_exc = run(lambda: oc.gen_comparable(left.copy(), right.copy(), wl, bl, **kw))
            got, got_exc = run(pooled)
        elif rnd % 3 == 0:  # every third round through compare(): cache threshold, score filter, descending order
            ckw = dict(kw, cache_threshold=random.Random(rnd).choice([None, 0.05, 0.3, 0.6]))
            column = ckw.pop("compare_column")
            want, want_exc = run(lambda: oc.compare(left.copy(), right.copy(), wl, bl, column, **ckw))
            got, got_exc = run(lambda: Questionnaire(left.copy()).compare(
                Questionnaire(right.copy()), wl, bl, column, cached=False, **ckw).dataframe())
        else:
            want, want_exc = run(lambda: oc.gen_comparable(left.copy(), right.copy(), wl, bl, **kw))
            got, got_exc = run(lambda: Questionnaire(left.copy()).gen_comparable(
                Questionnaire(right.copy()), wl, bl, **kw).dataframe())
        problem = None
        if (want_exc is None) != (got_exc is None) or (want_exc is not None and type(want_exc) is not type(got_exc)):
            problem = f"oracle: {want_exc!r}  /  package: {got_exc!r}"
            if got_exc is not None:
                traceback.print_exception(type(got_exc), got_exc, got_exc.__traceback__)
        elif want_exc is not None:
            stats["raised"][type(want_exc).__name__] = stats["raised"].get(type(want_exc).__name__, 0) + 1
        else:
            stats["frames"] += 1
            stats["rows"] += len(want)
            problem = rf.frames_differ(got, want, 0.0 if kw["score_func"] == "intersection_vs_union" else 1e-6)
        if problem:
            print(json.dumps({"FAIL": problem, "round_seed": rnd, "kw": kw, "n_left": len(left), "n_right": len(right),
                              "categories": kinds}))
            sys.exit(1)
    print(json.dumps({"ok": True, **stats, "seconds": args.seconds, "last_seed": rnd}))


if __name__ == "__main__":
    main()
