"""RAW Jaccard: global inverted index vs per-tile index vs signature kernel, kernel time only (HIP events), over vocabulary
sizes, id distributions and thresholds -- the data behind the launcher's choice (csrc/jaccard_raw_impl.hpp:
NSM_GLOBAL_INDEX_RATIO).  python tools/sweep_global.py [--rows N] [--width 16|32] [--zipf]   -> one JSON line per case"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "napkon-string-matching_amd"))


def prefix_class(width, thr):
    """(longest prefix, cls_end) as csrc/jaccard_raw_global.hip computes them."""
    kmin = {}
    for s in range(1, 2 * width + 1):
        kmin[s] = next((k for k in range(0, s // 2 + 1) if k / (s - k) >= thr), None)
    longest = 0
    for a in range(1, width + 1):
        needs = [kmin[a + b] for b in range(1, width + 1) if kmin[a + b] is not None and 1 <= kmin[a + b] <= min(a, b)]
        if needs:
            longest = max(longest, a - min(needs) + 1)
    return longest, (1 if longest <= 1 else 2 if longest <= 2 else 3 if longest <= 4 else 4 if longest <= 8 else 5)


def zipf_sets(n, seed, vocab, width, mean):
    """Sets whose ids follow a Zipf law; id 0 = the RAREST token (the numbering the header recommends)."""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    size = np.clip(rng.poisson(mean, n), 1, width)
    out = np.full((n, width), -1, dtype=np.int32)
    draws = rng.choice(vocab, size=(n, 2 * width), p=p)
    for r in range(n):
        u = list(dict.fromkeys(draws[r].tolist()))[: size[r]]
        out[r, : len(u)] = [vocab - 1 - v for v in u]  # frequent token -> large id
    return out


def main():
    import torch

    from napkon_string_matching_amd import _lib, grid, synthetic, tables

    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50_000)
    ap.add_argument("--width", type=int, default=16)
    ap.add_argument("--zipf", action="store_true")
    ap.add_argument("--vocabs", type=int, nargs="*", default=[1 << 17, 4096, 500, 60])
    ap.add_argument("--thresholds", type=float, nargs="*", default=[0.1, 0.2, 0.3, 0.5, 0.6, 0.8, 0.9])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    stream = torch.cuda.current_stream(dev).cuda_stream
    for vocab in args.vocabs:
        if args.zipf:
            left = zipf_sets(args.rows, 1, vocab, args.width, args.width / 2)
            right = zipf_sets(args.rows, 2, vocab, args.width, args.width / 2)
        else:
            kw = dict(id_range=vocab, width=args.width, mean=args.width / 2)
            left = synthetic.token_sets(args.rows, 1234, **kw)
            right = synthetic.plant_near_duplicate_sets(left, synthetic.token_sets(args.rows, 5678, **kw), 5679, id_range=vocab)
        lt = tables.SetTable.from_padded(left, "left", dev, width=args.width)
        rt = tables.SetTable.from_padded(right, "right", dev, width=args.width)
        ls, rs = lt.struct(), rt.struct()
        buf = grid.HitBuffer(1 << 24, dev)
        for thr in args.thresholds:
            longest, cls_end = prefix_class(args.width, thr)
            row = {"vocab": vocab, "zipf": args.zipf, "rows": args.rows, "width": args.width, "threshold": thr, "longest_prefix": longest,
                   "visited_estimate": rt.post_sq[cls_end - 1] * lt.n / max(1, rt.n), "pairs": lt.n * rt.n}
            modes = [("global", _lib.FLAG_PRUNE | _lib.FLAG_INDEX), ("sig", _lib.FLAG_PRUNE | _lib.FLAG_NO_INDEX), ("auto", _lib.FLAG_PRUNE)]
            if args.width <= 32:
                modes.insert(1, ("tile", _lib.FLAG_PRUNE | _lib.FLAG_INDEX | _lib.FLAG_TILE_INDEX))
            for name, flags in modes:
                def run():
                    buf.count.zero_()
                    _lib.check(lib.nsm_jaccard_raw_grid(ls, rs, thr, flags, buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(),
                                                        stream), name)
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 5
                e0.record()
                for _ in range(reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                row[name + "_ms"] = round(e0.elapsed_time(e1) / reps, 4)
                hits = int(buf.count.item())
                row.setdefault("hits", hits)
                assert hits == row["hits"], (row, name, hits)
            row["pairs_per_visited"] = round(row["pairs"] / max(1.0, row["visited_estimate"]), 1)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
