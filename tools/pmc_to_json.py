#!/usr/bin/env python3
"""rocprofv3 --pmc CSV output -> per-kernel counter means.

    tools/pmc_to_json.py <workload> <stamp> <outdir> <pass dir>...

Writes <outdir>/<workload>_sq_pmc.txt (human-readable, every nsm kernel) and <outdir>/pmc_<workload>.json
(read by bench.py: counter means per launch of every nsm kernel, the VALU-issue fraction and the HBM bytes,
stamped with the git head the caller passed and the sha256 of csrc/ so that a stale profile is visible).

Derived figures (MI355X_MICROARCH.md, sections "rocprofv3 PMC slots", "HBM", "DVFS give-back"):
  cycles            = GRBM_GUI_ACTIVE / 8              (rocprofv3 sums the 8 XCDs)
  valu_busy_frac    = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * cycles)       (SQ_ACTIVE_INST_* count quad-cycles)
  valu_issue_frac   = SQ_INSTS_VALU * 2 / (1024 * cycles)                   (a wave64 VALU op occupies a SIMD-32 for >= 2 cycles)
  hbm_bytes         = (2 * FETCH_SIZE + WRITE_SIZE) * 1024                  (gfx950: FETCH_SIZE reports half of wide reads; KiB units)
"""
import collections
import csv
import glob
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def csrc_hash() -> str:
    h = hashlib.sha256()
    for f in sorted((ROOT / "napkon-string-matching_amd" / "csrc").glob("*.h*")):
        if f.suffix in (".hip", ".hpp"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def short(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")  # (else every such kernel collapses to "nsm::")
    return name.split("(")[0].strip()


def main():
    workload, stamp, outdir = sys.argv[1:4]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in sys.argv[4:]:
        for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                name = short(r["Kernel_Name"])
                if "nsm::" not in name:
                    continue
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                key = (f, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    kernels = {}
    lines = [f"# workload {workload}, stamp {stamp}, csrc {csrc_hash()}; per-launch means over the profiled launches",
             "# (launches under --pmc run serialised and slower than untraced ones: use durations here only with these counters)"]
    for name in sorted(agg, key=lambda k: -sum(dur[k])):
        c = {k: sum(v) / len(v) for k, v in agg[name].items()}
        n_launch = max(len(v) for v in agg[name].values())
        mean_us = sum(dur[name]) / len(dur[name]) / 1e3
        entry = {"launches_profiled": n_launch, "mean_us_under_pmc": mean_us, "counters": c}
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc > 0:
            entry["cycles"] = cyc
            entry["clock_GHz"] = cyc / (mean_us * 1e3)
            if "SQ_ACTIVE_INST_VALU" in c:
                entry["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
            if "SQ_INSTS_VALU" in c:
                entry["valu_issue_frac"] = c["SQ_INSTS_VALU"] * 2 / (1024 * cyc)
            if "SQ_INSTS_SALU" in c:
                entry["salu_issue_frac"] = c["SQ_INSTS_SALU"] / (256 * cyc)  # one scalar unit per CU
            if "SQ_LDS_IDX_ACTIVE" in c:
                entry["lds_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            entry["hbm_bytes_per_launch"] = (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024
        kernels[name] = entry
        lines.append(f"{name}  launches={n_launch} mean_us={mean_us:.1f}")
        for k in sorted(c):
            lines.append(f"    {k:26s} {c[k]:.5g}")
        for k in ("cycles", "clock_GHz", "valu_busy_frac", "valu_issue_frac", "salu_issue_frac", "lds_busy_frac",
                  "hbm_bytes_per_launch"):
            if k in entry:
                lines.append(f"    -> {k:23s} {entry[k]:.5g}")
    Path(outdir, f"{workload}_sq_pmc.txt").write_text("\n".join(lines) + "\n")
    Path(outdir, f"pmc_{workload}.json").write_text(json.dumps(
        {"workload": workload, "git_head": stamp, "csrc_sha256_16": csrc_hash(),
         "method": "rocprofv3 --pmc <set> --kernel-trace, separate passes (tools/pmc_collect.sh)", "kernels": kernels},
        indent=1))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
