"""C2-shaped grid (50k x 50k, Poisson(8) ids of 2^17 or of --id-range): inverted-index kernel vs signature-prune kernel
over a range of thresholds, and what the library picks by itself.  python tools/sweep_index.py [--id-range N] [--width 32]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "napkon-string-matching_amd"))


def main():
    import torch

    from napkon_string_matching_amd import grid, synthetic, tables

    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=50_000)
    ap.add_argument("--id-range", type=int, default=0)
    ap.add_argument("--width", type=int, default=16, help="table width W (16 or 32); sets hold Poisson(W / 2) ids")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    kw = {"id_range": args.id_range} if args.id_range else {}
    sets = dict(kw, width=args.width, mean=args.width / 2)
    left = synthetic.token_sets(args.rows, 1234, **sets)
    right = synthetic.plant_near_duplicate_sets(left, synthetic.token_sets(args.rows, 5678, **sets), 5679, **kw)
    lt = tables.SetTable.from_padded(left, "left", dev, width=args.width)
    rt = tables.SetTable.from_padded(right, "right", dev, width=args.width)
    out = []
    for thr in (0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.35, 0.4, 0.45, 0.5, 0.6, 0.8):
        row = {"threshold": thr}
        for name, index in (("index", True), ("matrix", False), ("auto", None)):
            ms = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = grid.jaccard_raw_grid(lt, rt, thr, capacity=1 << 22, index=index)
                torch.cuda.synchronize()
                ms.append((time.perf_counter() - t0) * 1e3)
            row[name + "_ms"] = round(min(ms[1:]), 3)
            row.setdefault("hits", len(res))
            assert len(res) == row["hits"], (thr, name, len(res), row["hits"])
        out.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
