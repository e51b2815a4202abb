#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: mean counter values and mean duration."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        seen = set()
        for r in rows:
            name = r["Kernel_Name"].split("(")[0][-60:]
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (r["Dispatch_Id"], name)
            if key not in seen:
                seen.add(key)
                dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in agg.items():
            if "nsm" not in k:
                continue
            print(f"{k}  n={len(dur[k])} mean_us={sum(dur[k]) / len(dur[k]) / 1e3:.1f}")
            for c, vals in sorted(v.items()):
                print(f"    {c:28s} {sum(vals) / len(vals):.4g}")
