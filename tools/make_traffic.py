#!/usr/bin/env python3
"""profiles/traffic_<workload>.json (read by bench.py for roofline.traffic) from the per-kernel PMC
summary tools/pmc_summary.py wrote:  tools/make_traffic.py c2 profiles/r01_c2_hbm_pmc.txt 'jaccard_raw_kernel<16, true>'"""
import json
import re
import sys
from pathlib import Path

workload, summary, kernel = sys.argv[1:4]
vals = {}
name = None
for line in Path(summary).read_text().splitlines():
    if not line.startswith(" "):
        name = line.split("  n=")[0]
    else:
        m = re.match(r"\s+(\w+)\s+([0-9.e+-]+)", line)
        if m and name is not None and kernel in name:
            vals[m.group(1)] = float(m.group(2))
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
out = {
    "workload": workload,
    "kernel": kernel,
    "FETCH_SIZE_KiB": fetch,
    "WRITE_SIZE_KiB": write,
    "correction": "FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 reports half the bytes of wide coalesced reads; "
                  "the scalar-load share is uncalibrated, so this is an upper bound)",
    "hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --workload {workload} "
              f"--steps 3 --warmup 1`, {summary} (tools/refresh_profiles.sh)",
}
Path(f"profiles/traffic_{workload}.json").write_text(json.dumps(out, indent=1))
print(out["hbm_bytes_per_launch"])
