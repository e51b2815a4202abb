#!/usr/bin/env python3
"""RAW fuzzy_match grid on Term-like strings (Zipf-distributed words, 40..250 code units): the shape
of the reference's default configuration (compare_column Term, score_func fuzzy_match,
config.yml:13-14), which needs the multi-word kernels (row stride 128 / 256).

    python tools/bench_terms.py [--rows N] [--min-words A] [--max-words B] [--threshold T] [--check M]
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "napkon-string-matching_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def term_strings(n, seed, vocab, lo, hi, plant_from=None, fraction=0.01):
    import numpy as np

    rng = np.random.default_rng(seed)
    letters = np.array(list("enisratdhulcgmobwfkzvpjyxq"))
    weight = 1.0 / np.arange(1, len(letters) + 1) ** 0.7
    weight /= weight.sum()
    vrng = np.random.default_rng(99)
    words = ["".join(vrng.choice(letters, size=int(vrng.integers(3, 13)), p=weight)) for _ in range(vocab)]
    zipf = 1.0 / np.arange(1, vocab + 1)
    zipf /= zipf.sum()
    counts = rng.integers(lo, hi + 1, size=n)
    picks = rng.choice(vocab, size=int(counts.sum()), p=zipf)
    out, at = [], 0
    for c in counts:
        out.append(" ".join(words[k] for k in picks[at:at + c])[:256].strip())
        at += c
    if plant_from is not None:
        for t in rng.choice(n, size=max(1, int(fraction * n)), replace=False):
            src = plant_from[int(rng.integers(0, len(plant_from)))].split(" ")
            if len(src) > 2:
                src[int(rng.integers(0, len(src)))] = words[int(rng.integers(0, vocab))]
            out[t] = " ".join(src)[:256].strip()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50_000)
    ap.add_argument("--min-words", type=int, default=6)
    ap.add_argument("--max-words", type=int, default=16)
    ap.add_argument("--vocab", type=int, default=5000)
    ap.add_argument("--threshold", type=float, default=0.8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--no-exhaustive", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    from napkon_string_matching_amd import grid, tables

    dev = torch.device("cuda:0")
    left = term_strings(args.rows, 1, args.vocab, args.min_words, args.max_words)
    right = term_strings(args.rows, 2, args.vocab, args.min_words, args.max_words, plant_from=left)
    t0 = time.perf_counter()
    lt, rt = tables.encode_strings(left, right, dev)
    torch.cuda.synchronize()
    t_enc = time.perf_counter() - t0
    lens = np.array([len(s) for s in left])
    out = {"rows": args.rows, "stride": lt.stride, "alphabet": lt.alphabet, "threshold": args.threshold,
           "len_mean": float(lens.mean()), "len_max": int(lens.max()), "encode_seconds": t_enc}

    def timed(prune):
        hits = grid.indel_raw_grid(lt, rt, args.threshold, prune=prune)  # warm-up, sizes the hit buffer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hits = grid.indel_raw_grid(lt, rt, args.threshold, prune=prune, capacity=max(1024, 2 * len(hits.score)))
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps, hits

    ms, hits = timed(True)
    out["pruned_ms"] = ms * 1e3
    out["pruned_pairs_per_s"] = args.rows * args.rows / ms
    out["hits"] = len(hits.score)
    if not args.no_exhaustive:
        ms2, hits2 = timed(False)
        out["exhaustive_ms"] = ms2 * 1e3
        out["exhaustive_pairs_per_s"] = args.rows * args.rows / ms2
        out["prune_equals_exhaustive"] = hits.as_tuples() == hits2.as_tuples()
    if args.check:
        from oracle import score_functions as osf

        m = args.check
        a, b = tables.encode_strings(left[:m], right[:m], dev)
        got = {(i, j): s for s, i, j in grid.indel_raw_grid(a, b, args.threshold).as_tuples()}
        want = {}
        for i in range(m):
            for j in range(m):
                s = osf.fuzzy_match(left[i], right[j])
                if s >= args.threshold:
                    want[i, j] = s
        out["check"] = {"pairs": m * m, "identical": set(got) == set(want) and all(abs(got[k] - want[k]) <= 1e-6 for k in want)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
