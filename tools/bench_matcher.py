#!/usr/bin/env python3
"""End-to-end timing of the reference-shaped API (Matcher.match_questionnaires) on synthetic cohorts:
where does the time go between host preparation and the GPU grids?

    python tools/bench_matcher.py [--rows N] [--score-func intersection_vs_union|fuzzy_match]
"""
import argparse
import cProfile
import json
import pstats
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "napkon-string-matching_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=20_000)
    ap.add_argument("--score-func", default="intersection_vs_union")
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()

    import pandas as pd
    import torch

    from napkon_string_matching_amd import matching, synthetic
    from napkon_string_matching_amd.types.questionnaire import Questionnaire

    t0 = time.perf_counter()
    frames = {}
    for k, name in enumerate(("hap", "pop", "suep")):
        c = synthetic.c5_cohort(args.rows, 11 + k, plant_from=frames.get("_hap"))
        if name == "hap":
            frames["_hap"] = c
        tok = c["tok"]
        cats = synthetic.c5_category_lists(c)
        e, t = c["entries"], c["tokens_per_entry"]
        frames[name] = pd.DataFrame({
            "Identifier": [f"{name}#{i}" for i in range(args.rows)],
            "Variable": [f"{name}_v{i}" for i in range(args.rows)],
            "Sheet": [f"s{i % 7}" for i in range(args.rows)],
            "Category": cats,
            "Term": [[f"h{i % 5}", f"q {i}"] for i in range(args.rows)],
            "Tokens": [[" ".join(f"t{int(v)}" for v in row[x * t:(x + 1) * t]) for x in range(e)] for row in tok],
        })
    frames.pop("_hap")
    t_gen = time.perf_counter() - t0
    config = {"matching": {"score_threshold": 0.7, "cache_threshold": 0.5, "compare_column": "Tokens",
                           "score_func": args.score_func, "variable_score_threshold": 0.9,
                           "filter_categories": True},
              "steps": ["questionnaires"]}
    tables = {k: Questionnaire(v) for k, v in frames.items()}
    torch.cuda.synchronize()
    prof = cProfile.Profile() if args.profile else None
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    m = matching.match(config, write=False, questionnaires=tables)
    if prof:
        prof.disable()
    dt = time.perf_counter() - t0
    out = {"rows_per_cohort": args.rows, "score_func": args.score_func, "generate_s": round(t_gen, 2),
           "match_seconds": round(dt, 3), "pairs": 3 * args.rows ** 2, "pairs_per_s_end_to_end": 3 * args.rows ** 2 / dt,
           "hits": {k: len(v) for k, v in m.results.items()}}
    print(json.dumps(out))
    if prof:
        pstats.Stats(prof).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
