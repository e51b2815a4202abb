#!/usr/bin/env python3
"""Randomised differential test of every grid kernel against the C oracle (oracle/c/nsm_oracle.c), for a
time budget:  python tools/fuzz_parity.py [--seconds 300] [--seed 0]

Each round draws a kernel family, table shapes, vocabulary / alphabet sizes, category layout, partition
on/off and a threshold, and requires the hit list (score, i, j) to be IDENTICAL to the oracle's.  Exits
non-zero with the failing round's seed.  Test infrastructure: not part of the product path.
"""
import argparse
import os
import json
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "napkon-string-matching_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--family", default=None, help="only this kernel family (jaccard_raw, indel_raw, jaccard_levels, indel_levels, indel_split, wide)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from napkon_string_matching_amd import _lib, grid, tables
    from oracle import native

    dev = torch.device("cuda:0")
    thresholds = [0.0, 1e-9, 0.05, 0.1, 0.25, 1 / 3, 0.4, 0.5, 0.6, 2 / 3, 0.7, 0.75, 0.8, 0.9, 0.95, 1.0, 1.2]
    counts = {}
    t_end = time.time() + args.seconds
    rnd = args.seed
    total_hits = 0

    def check(got, want, what):
        nonlocal total_hits
        got_t = got.as_tuples()
        total_hits += len(want)
        if got_t != want:
            extra = sorted(set(got_t) - set(want))[:5]
            missing = sorted(set(want) - set(got_t))[:5]
            print(json.dumps({"FAIL": what, "round_seed": rnd, "got": len(got_t), "want": len(want),
                              "extra": extra, "missing": missing}))
            sys.exit(1)

    def rand_sets(rng, n, kmax, vocab, allow_empty):
        rows = []
        for _ in range(n):
            k = rng.randint(0 if allow_empty else 1, kmax)
            rows.append(rng.sample(range(vocab), min(k, vocab)))
        return rows

    def dup_some(rng, left, right, frac, mutate):
        for k in range(len(right)):
            if rng.random() < frac and left:
                right[k] = mutate(list(left[rng.randrange(len(left))]))

    def nested_item(rng, vocab, max_levels, max_new):
        base, out = [], []
        for _ in range(rng.randint(1, max_levels)):
            for v in rng.sample(range(vocab), min(vocab, rng.randint(0 if base else 1, max_new))):
                if v not in base:
                    base.append(v)
            out.append(list(base))
        return out

    def rand_string(rng, alphabet, lo, hi):
        return "".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))).strip()

    next_report = time.time() + 60.0
    while time.time() < t_end:
        rnd += 1
        if time.time() >= next_report:  # a long silent GPU job is taken to be hung
            print(json.dumps({"progress": counts, "round_seed": rnd}), flush=True)
            next_report += 60.0
        rng = random.Random(rnd)
        # posting entries of the global inverted index (include/nsm_hip.h: post_format): every format gets its share
        fmt_rng = random.Random(rnd * 7919 + 1)
        tables.COMPACT_POSTINGS = fmt_rng.random() < 0.8
        tables.RAW_POST_FORMAT = fmt_rng.choice([1, 2, 2])
        family = rng.choice(["jaccard_raw", "indel_raw", "jaccard_levels", "indel_levels", "indel_levels", "indel_split", "wide"])
        if args.family:
            family = args.family
        counts[family] = counts.get(family, 0) + 1
        thr = rng.choice(thresholds)
        n, m = rng.randint(1, 400), rng.randint(1, 600)
        if family == "wide":
            # operands beyond the fast kernels through the plugin faces (only the wide items leave the fast path)
            from napkon_string_matching_amd.compare import score_functions as sf

            n, m = rng.randint(1, 60), rng.randint(1, 90)
            if rng.random() < 0.5:
                vocab = rng.choice([150, 3000])
                size = lambda: rng.randint(65, 140) if rng.random() < 0.15 else rng.randint(1, 40)
                left = [rng.sample(range(vocab), min(vocab, size())) for _ in range(n)]
                right = [rng.sample(range(vocab), min(vocab, size())) for _ in range(m)]
                dup_some(rng, left, right, 0.1, lambda r: list(dict.fromkeys(r[: max(1, len(r) - rng.randint(0, 3))] + [rng.randrange(vocab)])))
                want = native.jaccard_raw(native.csr([sorted(set(r)) for r in left]), native.csr([sorted(set(r)) for r in right]), thr,
                                          cap=1 << 16)
                names = lambda rows: [[f"t{v}" for v in r] for r in rows]
                check(sf.intersection_vs_union.raw_grid(names(left), names(right), thr), want, f"wide jaccard_raw vocab={vocab} thr={thr} {n}x{m}")
            else:
                alphabet = rng.choice(["abcdefgh ", "".join(chr(0x4E00 + k) for k in range(rng.choice([40, 300, 600])))])
                length = lambda: rng.randint(513, 1400) if rng.random() < 0.1 else rng.randint(0, 80)
                left = [rand_string(rng, alphabet, 0, length()) for _ in range(n)]
                right = [rand_string(rng, alphabet, 0, length()) for _ in range(m)]
                dup_some(rng, left, right, 0.1, lambda s_: ("".join(s_)[:-1] + rng.choice(alphabet)).strip())
                cp = lambda ss: native.csr([[ord(c) for c in sf.fuzzy_operand(s_)] for s_ in ss])
                want = native.indel_raw(cp(left), cp(right), thr, cap=1 << 16)
                check(sf.fuzzy_match.raw_grid(left, right, thr), want, f"wide indel_raw |alphabet|={len(alphabet)} thr={thr} {n}x{m}")
            continue
        if family == "jaccard_raw":
            width = rng.choice([16, 16, 32, 64])
            # the inverted-index kernel (low thresholds) on request too, and with several chunks of left rows per block
            # the right table's global inverted index (prefix filter) forced / chosen by the library, the per-tile LDS index
            # (W <= 32), or no index at all
            index = rng.choice([None, None, True, True, False] + (["tile"] if width < 64 else [])) if thr > 0 else None
            if index and rng.random() < 0.4:
                n = rng.randint(1500, 7000)
            kmax = rng.randint(1, width)
            vocab = rng.choice([kmax + 1, 3 * kmax, 50 * kmax, 100_000])
            left = rand_sets(rng, n, kmax, vocab, allow_empty=False)
            right = rand_sets(rng, m, kmax, vocab, allow_empty=rng.random() < 0.3)

            def mutate(r):
                if len(r) > 1 and rng.random() < 0.5:
                    r[rng.randrange(len(r))] = rng.randrange(vocab)
                return list(dict.fromkeys(r))

            dup_some(rng, left, right, 0.1, mutate)
            pad = lambda rr: np.array([r + [-1] * (width - len(r)) for r in rr], dtype=np.int32).reshape(len(rr), width)
            lt = tables.SetTable.from_padded(pad(left), "left", dev, width=width)
            rt = tables.SetTable.from_padded(pad(right), "right", dev, width=width)
            want = native.jaccard_raw(native.csr(left), native.csr(right), thr, cap=1 << 19)
            prune = rng.random() < 0.7
            check(grid.jaccard_raw_grid(lt, rt, thr, prune=prune, capacity=rng.choice([64, 4096, 1 << 16]), index=index), want,
                  f"jaccard_raw W={width} kmax={kmax} vocab={vocab} thr={thr} prune={prune} index={index} {n}x{m}")
        elif family == "indel_raw":
            hi = rng.choice([8, 30, 64, 64, 100, 128, 200, 256, 400, 512])
            alphabet = rng.choice(["ab", "abcdefgh ", "abcdefghijklmnopqrstuvwxyz0123456789 ", "".join(chr(0x100 + k) for k in range(150))])
            n, m = min(n, 200), min(m, 300)
            left = [rand_string(rng, alphabet, 0 if rng.random() < 0.2 else 1, hi) for _ in range(n)]
            right = [rand_string(rng, alphabet, 0 if rng.random() < 0.2 else 1, hi) for _ in range(m)]

            def mutate(s):
                s = "".join(s)
                if s and rng.random() < 0.7:
                    k = rng.randrange(len(s))
                    s = s[:k] + rng.choice(alphabet) + s[k + rng.randint(0, 1):]
                return s.strip()[:hi]

            dup_some(rng, left, right, 0.1, mutate)
            lt, rt = tables.encode_strings(left, right, dev)
            cp = lambda ss: native.csr([[ord(c) for c in s] for s in ss])
            want = native.indel_raw(cp(left), cp(right), thr, cap=1 << 18)
            prune = rng.random() < 0.7
            two_stage = rng.random() < 0.7  # (64-unit tables: the 16-bucket first stage of the histogram filter, or not)
            check(grid.indel_raw_grid(lt, rt, thr, prune=prune, two_stage=two_stage), want,
                  f"indel_raw hi={hi} |alphabet|={len(alphabet)} thr={thr} prune={prune} two_stage={two_stage} {n}x{m}")
        else:
            ncat = rng.choice([0, 3, 6, 40, 64])
            mode = _lib.CAT_NONE if ncat == 0 else rng.choice([_lib.CAT_INTERSECT, _lib.CAT_INTERSECT_OR_BOTH_EMPTY])
            partition = rng.random() < 0.7

            def cats(k):
                out = np.zeros(k, dtype=np.uint64)
                for q in range(k):
                    for _ in range(rng.choice([0, 1, 1, 2, 3])):
                        out[q] |= np.uint64(1) << np.uint64(rng.randrange(max(1, ncat)))
                return out

            lcat, rcat = cats(n), cats(m)
            partition = partition and tables.partition_allowed(mode, lcat, rcat)
            if family == "jaccard_levels":
                vocab = rng.choice([12, 60, 400, 20_000])
                max_levels, max_new = rng.choice([1, 2, 4, 9, 20]), rng.choice([1, 2, 3, 8])
                left = [nested_item(rng, vocab, max_levels, max_new) for _ in range(n)]
                right = [nested_item(rng, vocab, max_levels, max_new) for _ in range(m)]
                dup_some(rng, left, right, 0.1, lambda it: [list(lv) for lv in (it[:-1] if len(it) > 1 and rng.random() < 0.5 else it)])
                biggest = max(len(it[-1]) for it in left + right)
                if biggest > 64:
                    continue
                width = tables.pick_width(biggest)
                vocabulary = tables.Vocabulary()
                lt = tables.SetTable.from_levels(left, "left", dev, vocabulary, width=width, categories=lcat, category_mode=mode,
                                                 partition=partition)
                rt = tables.SetTable.from_levels(right, "right", dev, vocabulary, width=width, categories=rcat, category_mode=mode,
                                                 partition=partition)
                want = native.levels(False, left, right, thr, lcat, rcat, mode, cap=1 << 19)
                # the inverted-index kernel forced / forbidden / chosen by the library (W <= 32, positive thresholds)
                index = rng.choice([None, True, True, False] + (["tile"] if width <= 32 else [])) if thr > 0 else None
                check(grid.jaccard_levels_grid(lt, rt, thr, category_mode=mode, index=index), want,
                      f"jaccard_levels vocab={vocab} levels<={max_levels} new<={max_new} W={width} thr={thr} mode={mode} "
                      f"partition={partition} ncat={ncat} index={index} {n}x{m}")
            else:
                n, m = min(n, 120), min(m, 200)
                lcat, rcat = lcat[:n], rcat[:m]
                hi = rng.choice([10, 40, 64, 64, 120, 250, 500])
                queue_cap = None
                if family == "indel_split":
                    # one-word strings at thresholds where the split path runs (scan -> survivor queue -> finish kernel),
                    # sometimes with a queue so small that it overflows (gated fused fallback)
                    hi = rng.choice([12, 30, 40, 64])
                    thr = rng.choice([0.7, 0.7, 0.75, 0.8, 0.9, 1.0])
                    queue_cap = rng.choice([None, None, 1024, 512 + 16 * 64, 512 + 16 * 2000])  # workspace bytes
                alphabet = rng.choice(["abc ", "abcdefghij klm", "abcdefghijklmnopqrstuvwxyz0123456789 "])
                max_levels = rng.choice([1, 2, 4, 4, 7])
                item = lambda: [rand_string(rng, alphabet, 0, hi) for _ in range(rng.randint(1, max_levels))]
                left, right = [item() for _ in range(n)], [item() for _ in range(m)]

                def mutate(it):
                    it = list(it)
                    k = rng.randrange(len(it))
                    it[k] = (it[k][:-1] + rng.choice(alphabet)).strip()
                    return it

                dup_some(rng, left, right, 0.1, mutate)
                li, ls, ri, rs = tables.encode_level_strings(left, right, dev, lcat, rcat, mode, partition=partition)
                cps = lambda items: [[[ord(c) for c in s] for s in it] for it in items]
                want = native.levels(True, cps(left), cps(right), thr, lcat, rcat, mode, cap=1 << 18)
                # multi-word strings: the shared-tile kernel (default) or the round-2 park kernel
                # (one-word strings: the split path at thresholds >= 0.7, else -- and with park -- the fused park kernel)
                park = rng.random() < 0.25 and family != "indel_split"
                prune = rng.random() < 0.8 or family == "indel_split"
                check(grid.indel_levels_grid(li, ls, ri, rs, thr, category_mode=mode, park=park, prune=prune, workspace=queue_cap), want,
                      f"{family} hi={hi} |alphabet|={len(alphabet)} levels<={max_levels} thr={thr} mode={mode} "
                      f"partition={partition} ncat={ncat} park={park} prune={prune} stride={ls.stride} queue_cap={queue_cap} {n}x{m}")
    print(json.dumps({"ok": True, "rounds": counts, "oracle_hits_compared": total_hits, "seconds": args.seconds,
                      "first_seed": args.seed + 1, "last_seed": rnd}))


if __name__ == "__main__":
    main()
