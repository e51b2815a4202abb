#!/usr/bin/env python3
"""A whole cohort of operands beyond the fast kernels (strings of 600..1500 code units; items of 80..200 tokens) through
the general kernels (csrc/any_grids.hip):  python tools/bench_wide.py [--rows 1500]  -> ms per grid at several thresholds."""
import argparse, json, random, sys, time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "napkon-string-matching_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1500)
    args = ap.parse_args()
    import torch

    from napkon_string_matching_amd.compare import score_functions as sf

    rng = random.Random(5)
    words = ["".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(3, 9))) for _ in range(3000)]
    text = lambda: " ".join(rng.choice(words) for _ in range(rng.randint(90, 230)))[:1500]
    left = [text() for _ in range(args.rows)]
    right = [text() for _ in range(args.rows)]
    for k in range(0, args.rows, 50):
        right[k] = left[rng.randrange(args.rows)][:-7] + "xyz"
    toks = lambda: [f"t{rng.randrange(5000)}" for _ in range(rng.randint(80, 200))]
    lset, rset = [toks() for _ in range(args.rows)], [toks() for _ in range(args.rows)]
    for k in range(0, args.rows, 50):
        rset[k] = list(lset[rng.randrange(args.rows)])[:-3]
    out = {"rows": args.rows}
    for name, fn, a, b in (("fuzzy_match", sf.fuzzy_match, left, right), ("intersection_vs_union", sf.intersection_vs_union, lset, rset)):
        for thr in (0.9, 0.7, 0.5, 0.2):
            fn.raw_grid(a[:64], b[:64], thr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hits = fn.raw_grid(a, b, thr)
            torch.cuda.synchronize()
            out[f"{name}@{thr}"] = {"ms": round((time.perf_counter() - t0) * 1e3, 1), "hits": len(hits)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
