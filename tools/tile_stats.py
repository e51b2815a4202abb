#!/usr/bin/env python3
"""Work counters of the shared-tile fuzzy levels kernel on the term workload (variant build with -DNSM_TILE_STATS):
    tools/build_variant.sh stats indel_levels.hip "-DNSM_TILE_STATS"
    NSM_HIP_LIBRARY=napkon-string-matching_amd/csrc/variants/libnsm_stats.so python tools/tile_stats.py [--threshold 0.5]
"""
import argparse
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "napkon-string-matching_amd"))

ap = argparse.ArgumentParser()
ap.add_argument("--threshold", type=float, default=0.5)
ap.add_argument("--rows", type=int, default=20000)
args = ap.parse_args()

import torch  # noqa: E402

import bench  # noqa: E402
from napkon_string_matching_amd import _lib, grid  # noqa: E402

dev = torch.device("cuda:0")
comm = bench.Comm(argparse.Namespace(dist_backend="nccl", allow_gloo=False), 0, 1, dev)
work = bench.Workload("term", comm, args.rows, dev, 0, args.threshold)
lib = _lib.load()
buf = grid.HitBuffer(1 << 24, dev)
stream = torch.cuda.current_stream(dev).cuda_stream
out = (ctypes.c_ulonglong * 16)()
lib.nsm_debug_tile_stats(out)  # reset
work.launch(buf, stream, True)
lib.nsm_debug_tile_stats(out)
names = ["batches", "step1 two-row passes", "step1 two-row iterations", "step1 one-row passes", "step1 one-row iterations",
         "later two-row passes", "later two-row iterations", "later one-row passes", "later one-row iterations",
         "dense calls", "dense LCS passes", "dense iterations", "parked pairs", "table builds"]
for k, nm in enumerate(names):
    print(f"{nm:28s} {out[k]:>14d}")
print("hits", int(buf.count.item()))
