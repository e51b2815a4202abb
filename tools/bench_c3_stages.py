#!/usr/bin/env python3
"""configs[2] through both RAW fuzzy pruning kernels on one box: two-stage filter (16 + 32 histogram buckets) against the
32-bucket kernel (NSM_FLAG_ONE_STAGE):  python tools/bench_c3_stages.py  -> kernel ms each (HIP events, 20 launches)."""
import json, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "napkon-string-matching_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch

from napkon_string_matching_amd import _lib, grid, synthetic, tables

(lc, ll), (rc, rl) = synthetic.c3_corpus()
dev = torch.device("cuda:0")
a = len(synthetic.STRING_ALPHABET)
lt, rt = tables.StrTable.from_codes(lc, ll, a, dev), tables.StrTable.from_codes(rc, rl, a, dev)
lib = _lib.load()
buf = grid.HitBuffer(1 << 16, dev)
out = {}
for name, flags in (("two_stage", _lib.FLAG_PRUNE), ("one_stage", _lib.FLAG_PRUNE | _lib.FLAG_ONE_STAGE)):
    def run():
        buf.count.zero_()
        _lib.check(lib.nsm_indel_raw_grid(lt.struct(), rt.struct(), 0.8, flags, buf.records.data_ptr(), buf.capacity,
                                          buf.count.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), name)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    out[name] = {"ms": round(e0.elapsed_time(e1) / 20, 3), "hits": int(buf.count.item())}
print(json.dumps(out))
