#!/usr/bin/env python3
"""Levels-mode (Matcher) kernels on BASELINE configs[4]-shaped cohorts, single GPU.

    python tools/bench_levels.py [--rows N] [--steps K] [--check M]

Three cohorts (hap / pop / suep) of N items, 4 entries x 2 words each, 1-2 of 32 categories,
filter_categories on, both score functions back to back, device threshold = max(cache_threshold 0.5, score_threshold 0.7)
(reference config.yml:11-12).  Prints one JSON line; `--check M` also replays an M x M corner of
every grid through the oracle (must be identical).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--check", type=int, default=0)
    ap.add_argument("--threshold", type=float, default=0.7)
    ap.add_argument("--vocab", type=int, default=20_000)
    ap.add_argument("--jaccard-flags", type=int, default=1, help="flags of nsm_jaccard_levels_grid: 1 | 4 (force index) | 8 (no index)")
    ap.add_argument("--indel-flags", type=int, default=1, help="flags of nsm_indel_levels_grid: 1 | 16 (the fused park kernel)")
    ap.add_argument("--workspace-mb", type=int, default=-1, help="split-path workspace handed to nsm_indel_levels_grid (-1 = what the library asks for)")
    ap.add_argument("--words", action="store_true", help="word-like tokens (synthetic.word_vocabulary) instead of t<digits>: the c5w corpus")
    ap.add_argument("--tile-stats", action="store_true", help="variant build with -DNSM_TILE_STATS: print the shared-tile kernel's work counters")
    ap.add_argument("--scan-stats", action="store_true", help="variant build with -DNSM_SCAN_STATS: print the scan's work counters")
    ap.add_argument("--tokens-per-entry", type=int, default=2,
                    help="words per entry; 6 makes the level strings 40..170 code units (multi-word Indel kernels)")
    args = ap.parse_args()

    import torch

    from napkon_string_matching_amd import _lib, grid, synthetic, tables
    from napkon_string_matching_amd.compare import score_functions as sf

    dev = torch.device("cuda:0")
    lib = _lib.load()
    names = ["hap", "pop", "suep"]
    cohorts = {}
    lex = synthetic.word_vocabulary(args.vocab) if args.words else None
    for k, nm in enumerate(names):  # pop and suep carry 1 % near-duplicates of hap items
        cohorts[nm] = synthetic.c5_cohort(args.rows, 11 + k, vocab=args.vocab, plant_from=cohorts.get("hap"),
                                          tokens_per_entry=args.tokens_per_entry, lex=lex)
    pairs = [("hap", "pop"), ("hap", "suep"), ("pop", "suep")]

    width = tables.pick_width(4 * args.tokens_per_entry)
    t0 = time.perf_counter()
    set_tables = {}
    for nm, c in cohorts.items():
        for side in ("left", "right"):
            set_tables[nm, side] = tables.SetTable.from_nested_arrays(
                c["ids"], c["plen"], c["nlev"], side, dev, categories=c["cat"], width=width,
                category_mode=_lib.CAT_INTERSECT_OR_BOTH_EMPTY)
    t_sets = time.perf_counter() - t0
    t0 = time.perf_counter()
    level_strings = {nm: [[sf.fuzzy_operand(lv) for lv in it] for it in synthetic.c5_level_token_lists(c)]
                     for nm, c in cohorts.items()}
    str_tables = {}
    for a, b in pairs:
        str_tables[a, b] = tables.encode_level_strings(level_strings[a], level_strings[b], dev,
                                                       cohorts[a]["cat"], cohorts[b]["cat"],
                                                       _lib.CAT_INTERSECT_OR_BOTH_EMPTY)
    t_strs = time.perf_counter() - t0

    buf = grid.HitBuffer(1 << 16, dev)
    buf.scratch = torch.empty_like(buf.records)
    stream = torch.cuda.current_stream(dev).cuda_stream
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY

    def run_jaccard(a, b):
        buf.count.zero_()
        _lib.check(lib.nsm_jaccard_levels_grid(set_tables[a, "left"].struct(), set_tables[b, "right"].struct(),
                                               args.threshold, set_tables[a, "left"].category_mode, args.jaccard_flags, buf.records.data_ptr(), buf.capacity,
                                               buf.count.data_ptr(), stream), "jaccard_levels")
        lib.nsm_sort_hits(buf.records.data_ptr(), buf.scratch.data_ptr(), buf.capacity, buf.count.data_ptr(), 0, args.rows, stream)

    ws_bytes = max(int(lib.nsm_indel_levels_workspace_bytes(t[0].struct(), t[1].struct(), t[2].struct(), t[3].struct(),
                                                            args.threshold, args.indel_flags, 0.0)) for t in str_tables.values())
    if args.workspace_mb >= 0 and ws_bytes:
        ws_bytes = args.workspace_mb << 20
    ws = grid.split_workspace(ws_bytes, dev) if ws_bytes else None  # the split path's survivor queue: caller-owned (ABI 4)

    def run_indel(a, b):
        li, ls, ri, rs = str_tables[a, b]
        buf.count.zero_()
        _lib.check(lib.nsm_indel_levels_grid(li.struct(), ls.struct(), ri.struct(), rs.struct(), args.threshold,
                                             li.category_mode, args.indel_flags, buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(),
                                             ws.data_ptr() if ws is not None else 0, ws.numel() * 8 if ws is not None else 0,
                                             0.0, stream), "indel_levels")
        lib.nsm_sort_hits(buf.records.data_ptr(), buf.scratch.data_ptr(), buf.capacity, buf.count.data_ptr(), 0, args.rows, stream)

    out = {"rows_per_cohort": args.rows, "threshold": args.threshold, "pairs_per_grid": args.rows ** 2,
           "tokens_per_entry": args.tokens_per_entry, "string_stride": str_tables[pairs[0]][1].stride,
           "encode_seconds": {"sets": round(t_sets, 2), "level_strings": round(t_strs, 2)}}
    for label, fn in (("intersection_vs_union", run_jaccard), ("fuzzy_match", run_indel)):
        fn(*pairs[0])
        torch.cuda.synchronize()
        hits = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for a, b in pairs:
                fn(a, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        for a, b in pairs:
            fn(a, b)
            hits.append(int(buf.count.item()))
        out[label] = {"ms_per_3_grids": dt * 1e3, "pairs_per_s": 3 * args.rows ** 2 / dt, "hits": hits}
        if label == "fuzzy_match" and ws is not None:
            ctl = ws[:64].cpu().numpy()
            out[label]["split_overflowed"] = bool(int(ctl[1]) & 0xFFFFFFFF)
            out[label]["split_round_survivors_last_grid"] = [int(v) for v in ctl[2:64] if v > 0]

    if args.scan_stats:
        import ctypes

        st = (ctypes.c_ulonglong * 8)()
        lib.nsm_debug_scan_stats(st)  # reset
        run_indel(*pairs[0])
        lib.nsm_debug_scan_stats(st)
        out["scan_stats_first_grid"] = dict(zip(
            ["rows_visited", "pairs_in_category", "pairs_alive_after_H", "rows_scored", "pairs_alive_after_step1",
             "two_row_passes", "one_row_passes"], [int(v) for v in st]))

    if args.tile_stats:
        import ctypes

        st = (ctypes.c_ulonglong * 16)()
        lib.nsm_debug_tile_stats(st)  # reset
        run_indel(*pairs[0])
        lib.nsm_debug_tile_stats(st)
        out["tile_stats_first_grid"] = dict(zip(
            ["batches", "step1 two-row passes", "step1 two-row iterations", "step1 one-row passes", "step1 one-row iterations",
             "later two-row passes", "later two-row iterations", "later one-row passes", "later one-row iterations",
             "dense calls", "dense LCS passes", "dense iterations", "parked pairs", "table builds"], [int(v) for v in st]))

    if args.check:
        from oracle import native

        m = args.check
        for a, b in pairs:
            la = synthetic.c5_level_token_lists(cohorts[a], slice(0, m))
            lb = synthetic.c5_level_token_lists(cohorts[b], slice(0, m))
            vocab = {}
            ids = lambda items: [[[vocab.setdefault(t, len(vocab)) for t in lv] for lv in it] for it in items]
            want = native.levels(False, ids(la), ids(lb), args.threshold, cohorts[a]["cat"][:m], cohorts[b]["cat"][:m], 2)
            sub_l = tables.SetTable.from_nested_arrays(cohorts[a]["ids"][:m], cohorts[a]["plen"][:m], cohorts[a]["nlev"][:m],
                                                       "left", dev, categories=cohorts[a]["cat"][:m], width=width,
                                                       category_mode=mode)
            sub_r = tables.SetTable.from_nested_arrays(cohorts[b]["ids"][:m], cohorts[b]["plen"][:m], cohorts[b]["nlev"][:m],
                                                       "right", dev, categories=cohorts[b]["cat"][:m], width=width,
                                                       category_mode=mode)
            got = grid.jaccard_levels_grid(sub_l, sub_r, args.threshold, category_mode=mode).as_tuples()
            assert got == want, (a, b, len(got), len(want))
            sl = [[sf.fuzzy_operand(lv) for lv in it] for it in la]
            sr = [[sf.fuzzy_operand(lv) for lv in it] for it in lb]
            cps = lambda items: [[[ord(ch) for ch in s] for s in it] for it in items]
            want = native.levels(True, cps(sl), cps(sr), args.threshold, cohorts[a]["cat"][:m], cohorts[b]["cat"][:m], 2)
            li, ls, ri, rs = tables.encode_level_strings(sl, sr, dev, cohorts[a]["cat"][:m], cohorts[b]["cat"][:m], mode)
            got = grid.indel_levels_grid(li, ls, ri, rs, args.threshold, category_mode=mode).as_tuples()
            assert [(i, j) for _, i, j in got] == [(i, j) for _, i, j in want]
            assert all(abs(x[0] - y[0]) <= 1e-6 for x, y in zip(got, want))
        out["check"] = f"{m}x{m} corner of every grid identical to the oracle"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
