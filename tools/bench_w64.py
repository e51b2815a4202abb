#!/usr/bin/env python3
"""RAW Jaccard on wide sets (W = 64: 33..64 distinct tokens per item), pruned vs exhaustive:
    python tools/bench_w64.py [--rows N] [--threshold T]"""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "napkon-string-matching_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=20000); ap.add_argument("--threshold", type=float, default=0.5)
args = ap.parse_args()
import torch
from napkon_string_matching_amd import grid, synthetic, tables
dev = torch.device("cuda:0")
left = synthetic.token_sets(args.rows, 1, mean=44.0, width=64)
right = synthetic.plant_near_duplicate_sets(left, synthetic.token_sets(args.rows, 2, mean=44.0, width=64), 3)
lt, rt = tables.SetTable.from_padded(left, "left", dev, width=64), tables.SetTable.from_padded(right, "right", dev, width=64)
out = {"rows": args.rows, "threshold": args.threshold}
for name, kw in (("pruned", dict(prune=True)), ("exhaustive", dict(prune=False))):
    h = grid.jaccard_raw_grid(lt, rt, args.threshold, **kw); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): h = grid.jaccard_raw_grid(lt, rt, args.threshold, **kw)
    torch.cuda.synchronize(); out[name + "_ms"] = (time.perf_counter() - t0) / 3 * 1e3; out[name + "_hits"] = len(h.score)
print(json.dumps(out))
