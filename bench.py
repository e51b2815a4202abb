#!/usr/bin/env python3
"""Headline benchmark: pair-comparisons/sec of the all-pairs match loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload all|c2|c2low|c3|c4|c5|c5w|term] [--no-cpu-baseline]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(one process per GPU, RCCL).  One "step" = one pass of the hot path over one synthetic batch whose
operand tables are already resident in HBM: zero the hit counter, score the whole N x M pair grid
on the GPU (threshold test included), order the hits canonically and -- for N > 1 -- all-gather
the (score, i, j) hits of every rank.  Left rows are block-sharded over the ranks, the right side is
replicated; per-GPU work is fixed as N grows ("weak").

The headline workload is BASELINE.json configs[2] ("c3": 200k x 200k strings of 16..64 code units, fuzzy_match,
threshold 0.8) -- the LARGEST configuration BASELINE.json quotes on one GPU (4e10 pairs per step; configs[1], "c2":
50k x 50k token-id sets, mean 8 ids, intersection_vs_union, threshold 0.5, is 2.5e9).  `--workload c4` is configs[3]
(1M x 1M token-id sets, threshold 0.8, the 1M left rows divided over the ranks: "strong"); `--workload c5` is
configs[4] (three 500k-item cohorts, 4 levels, 32 categories, filter_categories, both score functions back to back
through the levels-mode kernels, left rows of every cohort pair divided over the ranks: "strong") at config.yml's
score threshold 0.7; `--workload c5w` is the same shape on WORD-LIKE text (20 000 distinct words of 3..7 letters
instead of "t<digits>") at the threshold the reference's flow hands to the grid when a cache threshold is set, 0.5
(types/comparable_data.py:102-108, config.yml:11-12), with the 0.7 run reported beside it.

The JSON line carries, besides the driver's contract fields:
  roofline      dominant kernel against the bound it can actually approach, the VALU issue rate (this path
                keeps its operands on chip: measured HBM traffic is < 2 % of peak, and the "algorithmic bytes"
                of SURVEY.md 8d -- 128 B per pair -- exceed what any kernel moves by two orders of magnitude,
                so a fraction of HBM peak built on them is > 1 and says nothing).
                  achieved = VALU wave-instructions per launch (SQ_INSTS_VALU, profiles/pmc_<workload>.json,
                             collected by tools/pmc_collect.sh) / the kernel's duration measured here with HIP
                             events on the launch stream
                  peak     = 1024 SIMDs x 2.4 GHz / 2 cycles (a wave64 VALU op occupies a SIMD-32 for >= 2 cycles)
                  frac     = achieved / peak  (<= 1 by construction; half-rate VOP3 ops cap it near 0.5)
                beside it: frac_pmc (the same ratio from the profile's own cycle count), valu_busy_quadcycles
                (SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles: counts a quad-cycle per instruction, can exceed 1),
                traffic (HBM bytes per launch, FETCH_SIZE x 2 + WRITE_SIZE) with hbm_frac, the algorithmic and
                compulsory byte counts, and the profile's stamp (git head, hash of csrc/ -- a stale profile shows).
  exhaustive    the same grid with the exact prune disabled (every pair's full comparison evaluated)
  cpu_baseline  the oracle's restatement of the reference's loop, 1 core, bounded sample
  c2 c2low c4 c5 c5w term   (default `--workload all`) every other BASELINE config, configs[4] on word-like text
                and the reference's default configuration, timed in the SAME invocation after the headline, each
                under the same contract (W warm-up steps, K timed steps between fences; c5 / c5w: min(K, 5) / min(K, 3)
                steps of six 500k x 500k grids) with its own value / ms_per_step / steps / kernel_ms / roofline /
                exhaustive / cpu_baseline.  N = 1 runs all six; N > 1 runs the configs BASELINE.json quotes on 8
                GPUs (c4, c5 and c5w: left rows divided over the ranks, "strong").  `--workload <one>` times that
                workload alone.
  config.sub    the sub-records' numbers once more, as plain scalars under the `config` object (a parser that keeps
                `config` and drops unknown top-level keys still carries them): per workload ms_per_step, value,
                kernel_ms, roofline_frac, exhaustive_ms_per_step, cpu_baseline_value, steps.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

# ROCr reads HSA_* at hsa_init: this must be in the environment before torch touches the GPU (the host driver
# only supports dmabuf IPC; without it RCCL's P2P setup fails with hipIpcGetMemHandle: invalid argument)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "napkon-string-matching_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2  # G wave64-instructions/s: 1024 SIMD-32 x 2.4 GHz, >= 2 cycles per instruction
BYTES_PER_PAIR = 128  # two 64-byte operand rows


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("all", "c2", "c2low", "c3", "c4", "c5", "c5w", "term"), default="all",
                    help="all = the c3 headline line with every other workload attached as a sub-record")
    ap.add_argument("--rows", type=int, default=0, help="override rows per side per GPU (debug)")
    ap.add_argument("--right-rows", type=int, default=0, help="override the right side's rows (debug)")
    ap.add_argument("--threshold", type=float, default=None, help="override the workload's threshold (debug)")
    ap.add_argument("--id-range", type=int, default=0, help="c2 / c4: draw token ids from [0, N) instead of 2^17 (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--post-format", type=int, choices=(0, 1, 2), default=None,
                    help="posting entries of the RAW tables' global index: 0 = 64 bits, 1 = 32 bits, 2 = 32 bits + signature "
                         "fold (the default where the rows fit); A/B runs of the global-index kernel")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the exhaustive (prune off) run and c5w's second run at the score threshold (profiling passes)")
    ap.add_argument("--graph", type=int, default=0, help="1: replay each step as one captured hipGraph")
    ap.add_argument("--serial-sort", action="store_true",
                    help="N = 1: order the hits on the launch stream instead of a second stream (no overlap with the next step's grid)")
    ap.add_argument("--capacity", type=int, default=1 << 13)
    ap.add_argument("--c5-stride", type=int, default=64, choices=(64, 128),
                    help="c5 / c5w: row stride of the level-string tables (128 = the shared-tile kernel, A/B runs)")
    ap.add_argument("--split-workspace-mb", type=int, default=-1,
                    help="c5 / c5w: MiB of split-path workspace handed to nsm_indel_levels_grid (-1 = what the library asks "
                         "for, 0 = none: the single-kernel path; A/B runs)")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo = CPU-staged rehearsal of the N > 1 path (e.g. 2 ranks sharing one GPU)")
    ap.add_argument("--allow-gloo", action="store_true",
                    help="if RCCL cannot be brought up on every rank, run the exchange staged through host memory "
                         "instead of exiting non-zero (the JSON's config.exchange says which one ran)")
    ap.add_argument("--no-c4", action="store_true", help="(kept for old command lines) the headline alone, no sub-records")
    return ap.parse_args()


class Workload:
    """Operand tables in HBM + the launch closure of one workload."""

    def __init__(self, name, comm, rows, device, right_rows=0, threshold=None, id_range=0):
        import numpy as np
        import torch

        from napkon_string_matching_amd import _lib, synthetic, tables

        rank, world = comm.rank, comm.world
        self.comm = comm
        self.name = name
        self.lib = _lib.load()
        self.flag_prune = _lib.FLAG_PRUNE
        self.scaling = "weak"
        lo_global = None
        if name == "c4":  # configs[3]: the C2 generator at 1M x 1M, threshold 0.8, left rows divided over the ranks
            from napkon_string_matching_amd import distributed

            total = rows or 1_000_000
            right_rows = right_rows or total
            lo_global, hi_global = distributed.shard_bounds(total, rank, world)
            rows = hi_global - lo_global
            threshold = 0.8 if threshold is None else threshold
            self.scaling = "strong"
        if name == "c2low":  # configs[1]'s grid at the API's default score_threshold (types/comparable_data.py:74)
            threshold = 0.1 if threshold is None else threshold
        if name in ("c2", "c2low", "c4"):
            n = rows or 50_000
            m = right_rows or rows or 50_000
            self.threshold = 0.5 if threshold is None else threshold
            kw = {"id_range": id_range} if id_range else {}
            right = synthetic.token_sets(m, 5678, **kw)
            if name == "c4":
                # ONE global left corpus (seeded: every rank generates it and keeps its block), so that the hits of
                # the ranks together are the hits of the 1-rank run
                whole = synthetic.token_sets(total, 1234, **kw)
                left = np.ascontiguousarray(whole[lo_global:lo_global + n])
                right_planted = synthetic.plant_near_duplicate_sets(whole, right, 5679, **kw) if rank == 0 else None
                del whole
            else:
                # weak scaling: global left corpus = world * n rows; this rank scores rows [rank*n, (rank+1)*n)
                left = synthetic.token_sets(n, 1234 + 1000 * rank, **kw)
                right_planted = synthetic.plant_near_duplicate_sets(left, right, 5679, **kw) if rank == 0 else None
            self.host = (left, right, right_planted)
            self.left_np = left
            right = comm.broadcast_numpy(right_planted, right)
            self.right_np = right
            orig = np.arange(n, dtype=np.int32) + (rank * n if lo_global is None else lo_global)
            t_enc = time.perf_counter()
            self.left = tables.SetTable.from_padded(left, "left", device, orig=orig)
            self.right = tables.SetTable.from_padded(right, "right", device)
            torch.cuda.synchronize(device)
            self.encode_h2d_seconds = time.perf_counter() - t_enc
            self.launch_fn = self.lib.nsm_jaccard_raw_grid
            # the right table carries a global inverted index (tables.SetTable.from_padded): on these vocabularies the
            # library generates candidates from it at every threshold (csrc/jaccard_raw_global.hip)
            self.kernel = self.kernel_match = "jaccard_raw_global_kernel<16"  # (<16, 2>: entries with the signature fold)
            self.kernel_match_exhaustive = "jaccard_raw_kernel<16, false>"
            self.dtype = "int32"
            self.label = (f"{name.upper()}: {n}x{m} token-id sets/GPU (Poisson(8) ids, W=16), intersection_vs_union RAW, "
                          f"threshold {self.threshold}")
        elif name == "term":
            # the reference's DEFAULT configuration (config.yml:11-14,22): compare_column Term, score_func
            # fuzzy_match, gen_comparable at cache_threshold 0.5, no category filter -> levels mode over joined
            # level strings of 60..240 code units (multi-word LCS).  The right cohort is the same on every rank.
            from napkon_string_matching_amd.compare import score_functions as sf

            n = rows or 20_000
            m = right_rows or rows or 20_000
            self.threshold = 0.5 if threshold is None else threshold
            left_items = synthetic.term_cohort(n, 1234 + 1000 * rank)
            right_items = synthetic.term_cohort(m, 5678, plant_from=synthetic.term_cohort(n, 1234))
            self.left_np = [[sf.fuzzy_operand(lv) for lv in it] for it in synthetic.term_levels(left_items)]
            self.right_np = [[sf.fuzzy_operand(lv) for lv in it] for it in synthetic.term_levels(right_items)]
            t_enc = time.perf_counter()
            self.term_tables = tables.encode_level_strings(self.left_np, self.right_np, device, left_offset=rank * n)
            torch.cuda.synchronize(device)
            self.encode_h2d_seconds = time.perf_counter() - t_enc
            li, ls, ri, rs = self.term_tables
            self.left, self.right = ls, rs
            self.term_structs = (li.struct(), ls.struct(), ri.struct(), rs.struct())
            self.kernel = f"indel_levels_tile_kernel<{ls.stride // 64}>"
            self.kernel_match = f"indel_levels_tile_kernel<{ls.stride // 64}>"
            self.kernel_match_exhaustive = f"indel_levels_kernel<{ls.stride // 64}>"
            self.dtype = "u64"
            lens = [len(s) for it in self.left_np for s in it[1:]]
            self.label = (f"TERM: {n}x{m} Term-shaped items/GPU (3-5 entries -> 3-5 suffix-nested levels, joined level strings "
                          f"mean {sum(lens) / max(1, len(lens)):.0f} / max {max(lens)} code units, stride {ls.stride}), "
                          f"compare_terms x fuzzy_match, no categories, threshold {self.threshold}")
        else:
            n = rows or 200_000
            m = right_rows or rows or 200_000
            self.threshold = 0.8 if threshold is None else threshold
            left = synthetic.strings(n, 1234 + 1000 * rank)
            right = synthetic.strings(m, 5678)
            if rank == 0:
                rp = synthetic.plant_near_duplicate_strings(left, right, 5679)
                packed = np.concatenate([rp[0], rp[1].astype(np.int32).view(np.uint8).reshape(m, 4)], axis=1)
            else:
                packed = None
            packed = comm.broadcast_numpy(packed, np.zeros((m, 68), np.uint8))
            rc, rl = packed[:, :64].copy(), packed[:, 64:].copy().view(np.int32).reshape(m)
            self.left_np, self.right_np = left, (rc, rl)
            orig = np.arange(n, dtype=np.int32) + rank * n
            alpha = len(synthetic.STRING_ALPHABET)
            t_enc = time.perf_counter()
            self.left = tables.StrTable.from_codes(left[0], left[1], alpha, device, orig=orig)
            self.right = tables.StrTable.from_codes(rc, rl, alpha, device)
            torch.cuda.synchronize(device)
            self.encode_h2d_seconds = time.perf_counter() - t_enc
            self.launch_fn = self.lib.nsm_indel_raw_grid
            self.kernel = "indel_raw_coarse_kernel (two-stage histogram filter)"
            self.kernel_match, self.kernel_match_exhaustive = "indel_raw_coarse_kernel", "indel_raw_kernel<false>"
            self.dtype = "u64"
            self.label = f"C3: {n}x{m} strings/GPU (len U[16,64], 37 symbols), fuzzy_match RAW, threshold {self.threshold}"
        self.n, self.m = n, m
        self.ls, self.rs = self.left.struct(), self.right.struct()
        self.bytes_per_pair = BYTES_PER_PAIR
        if name == "term":  # both items' level strings (mean levels x row stride)
            li, ls, ri, rs = self.term_tables
            self.bytes_per_pair = int(round((ls.n / max(1, li.n) + rs.n / max(1, ri.n)) * ls.stride))

    def launch(self, buf, stream, prune=True):
        from napkon_string_matching_amd import _lib

        flags = self.flag_prune if prune else 0
        if self.name == "term":  # prune off = every step scored wave-wide (no parking, no histogram bound)
            st = self.term_structs
            flags = self.flag_prune if prune else _lib.FLAG_WAVE_WIDE
            _lib.check(
                self.lib.nsm_indel_levels_grid(st[0], st[1], st[2], st[3], float(self.threshold), _lib.CAT_NONE, flags,
                                               buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(), 0, 0, 0.0, stream),
                self.kernel,
            )
            return
        _lib.check(
            self.launch_fn(self.ls, self.rs, float(self.threshold), flags, buf.records.data_ptr(), buf.capacity,
                           buf.count.data_ptr(), stream),
            self.kernel,
        )


class Comm:
    """Process groups of an N > 1 run.  The default group is gloo (control plane: barriers, agreement, host-side
    numbers); the device-resident exchange runs on an RCCL group that is brought up and probed on EVERY rank, and
    the ranks agree on the outcome over gloo before anyone uses it -- a rank-local fallback would leave the ranks
    split over two backends and the next collective hanging."""

    def __init__(self, args, rank, world, device):
        import torch

        self.rank, self.world, self.device = rank, world, device
        self.dist = None
        self.dev_group = None
        self.exchange = None
        self.rccl_ranks = None
        if world == 1:
            return
        from datetime import timedelta

        import torch.distributed as dist

        self.dist = dist
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=600))
        if args.dist_backend == "nccl":
            ok, err = 1, ""
            try:  # one tiny collective up front so that a broken transport shows here, not mid-run
                group = dist.new_group(backend="nccl", timeout=timedelta(seconds=180))
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe, group=group)
                torch.cuda.synchronize(device)
                self.rccl_ranks = int(probe.item())
                if self.rccl_ranks != world:
                    raise RuntimeError(f"RCCL all-reduce saw {self.rccl_ranks} ranks, expected {world}")
                self.dev_group = group
            except Exception as exc:  # noqa: BLE001 -- reported below, after the ranks have agreed
                ok, err = 0, f"{type(exc).__name__}: {exc}"
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # gloo: every rank learns whether ALL ranks are up
            if int(flag.item()) == 0:
                self.dev_group = None
                if err:
                    print(f"[bench] rank {rank}: RCCL unavailable ({err})", file=sys.stderr)
                if not args.allow_gloo:
                    dist.barrier()
                    dist.destroy_process_group()
                    raise SystemExit("bench.py: RCCL could not be brought up on every rank; pass --allow-gloo to "
                                     "measure the host-staged exchange instead (that is NOT an RCCL number)")
                if rank == 0:
                    print("[bench] --allow-gloo: the hit exchange is staged through host memory", file=sys.stderr)
        self.exchange = "rccl all-gather" if self.dev_group is not None else "gloo, staged through host memory"

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_seconds(self, dt):
        if self.dist is None:
            return dt
        import torch

        t = torch.tensor([dt], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, v):
        """Sum of a host integer over the ranks (gloo control plane)."""
        if self.dist is None:
            return int(v)
        import torch

        t = torch.tensor([int(v)], dtype=torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def broadcast_numpy(self, array, like):
        """Rank 0's array on every rank (the replicated right side)."""
        import torch

        if self.dist is None:
            return array
        t = torch.from_numpy(array if array is not None else like.copy())
        if self.dev_group is not None:
            t = t.to(self.device)
            self.dist.broadcast(t, src=0, group=self.dev_group)
            return t.cpu().numpy()
        self.dist.broadcast(t, src=0)
        return t.numpy()

    def gather_hits(self, buf, out, async_op=False):
        """all-gatherv of the (score, i, j) hits.  RCCL has no native gatherv; the hit buffer carries its own
        counter in a trailing record (grid.HitBuffer), so ONE all-gather of the max-padded storage moves records
        and counts together.  Returns the work handle when ``async_op``."""
        from napkon_string_matching_amd import distributed

        # the product's exchange (ComparableData.gen_comparable under a process group) is this same function
        _, work = distributed.all_gather_storage(buf.storage, out=out, group=self.dev_group, async_op=async_op)
        return work

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def csrc_hash():
    """sha256 (16 hex digits) of the kernel sources: profiles are stamped with it (tools/pmc_to_json.py)."""
    h = hashlib.sha256()
    for f in sorted((ROOT / "napkon-string-matching_amd" / "csrc").glob("*.h*")):
        if f.suffix in (".hip", ".hpp"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def valu_roofline(profile_name, kernel_match, kernel_label, kernel_ms, launches_per_ms_sample, algorithmic_bytes,
                  compulsory_bytes, default_shape=True):
    """The roofline object (see the module docstring).  ``kernel_ms`` = duration of ``launches_per_ms_sample``
    launches of the dominant kernel, measured by the caller with HIP events."""
    roof = {
        "bound": "valu_issue",
        "kernel": kernel_label,
        "achieved": None,
        "peak": VALU_PEAK_GINST,
        "unit": "G wave64 VALU instructions/s",
        "frac": None,
        "traffic": None,
        "kernel_ms": kernel_ms,
        "algorithmic_bytes_per_launch": algorithmic_bytes,
        "compulsory_hbm_bytes_per_launch": compulsory_bytes,
        "effective_GBps_of_algorithmic_bytes": algorithmic_bytes * launches_per_ms_sample / (kernel_ms * 1e-3) / 1e9,
    }
    pfile = ROOT / "profiles" / f"pmc_{profile_name}.json"
    if not default_shape:  # --rows / --right-rows / --threshold: the profile's instruction counts are another grid's
        roof["profile"] = "not applicable: the workload's shape was overridden on the command line"
        return roof
    if not pfile.exists():
        roof["profile"] = f"{pfile.name} missing: run tools/pmc_collect.sh on the GPU box"
        return roof
    prof = json.loads(pfile.read_text())
    if isinstance(kernel_match, tuple):
        # a grid call that is several launches (the split path of the one-word fuzzy levels grid: scan and finish kernel,
        # once per round): kernel_match = (kernels..., the kernel launched ONCE per call); the counters of one call are the
        # kernels' per-launch means times their launches, over the number of calls profiled
        *names, once = kernel_match
        per_call = next((v for k, v in prof["kernels"].items() if once in k), None)
        parts = [v for n in names for k, v in prof["kernels"].items() if n in k]
        if per_call is None or len(parts) != len(names):
            roof["profile"] = f"{pfile.name} has no kernels matching {kernel_match!r}"
            return roof
        calls = per_call["launches_profiled"]
        keys = set.intersection(*(set(v["counters"]) for v in parts))
        counters = {k: sum(v["counters"][k] * v["launches_profiled"] for v in parts) / calls for k in keys}
        entry = {"counters": counters, "launches_per_call": [v["launches_profiled"] / calls for v in parts],
                 "mean_us_under_pmc": sum(v["mean_us_under_pmc"] * v["launches_profiled"] for v in parts) / calls}
        cyc = counters.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc > 0 and "SQ_INSTS_VALU" in counters:
            entry["valu_issue_frac"] = counters["SQ_INSTS_VALU"] * 2 / (1024 * cyc)
        if cyc > 0 and "SQ_ACTIVE_INST_VALU" in counters:
            entry["valu_busy_frac"] = counters["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
        if cyc > 0 and "SQ_INSTS_SALU" in counters:
            entry["salu_issue_frac"] = counters["SQ_INSTS_SALU"] / (256 * cyc)
        if cyc > 0 and "SQ_LDS_IDX_ACTIVE" in counters:
            entry["lds_busy_frac"] = counters["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
        if all("hbm_bytes_per_launch" in v for v in parts):
            entry["hbm_bytes_per_launch"] = sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in parts) / calls
        roof["launches_per_grid_call"] = dict(zip(names, entry["launches_per_call"]))
    else:
        entry = next((v for k, v in prof["kernels"].items() if kernel_match in k), None)
    if entry is None:
        roof["profile"] = f"{pfile.name} has no kernel matching {kernel_match!r}"
        return roof
    c = entry["counters"]
    insts = c["SQ_INSTS_VALU"] * launches_per_ms_sample
    roof["achieved"] = insts / (kernel_ms * 1e-3) / 1e9
    roof["frac"] = roof["achieved"] / VALU_PEAK_GINST
    roof["frac_pmc"] = entry.get("valu_issue_frac")
    roof["valu_busy_quadcycles"] = entry.get("valu_busy_frac")
    roof["salu_issue_frac"] = entry.get("salu_issue_frac")
    roof["lds_busy_frac"] = entry.get("lds_busy_frac")
    roof["valu_instructions_per_launch"] = c["SQ_INSTS_VALU"]
    if "hbm_bytes_per_launch" in entry:
        roof["traffic"] = entry["hbm_bytes_per_launch"] * launches_per_ms_sample
        roof["hbm_frac"] = roof["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
    roof["profile"] = {
        "file": f"profiles/{pfile.name}",
        "git_head": prof.get("git_head"),
        "csrc_sha256_16": prof.get("csrc_sha256_16"),
        "matches_source": prof.get("csrc_sha256_16") == csrc_hash(),
        "kernel_us_under_pmc": entry.get("mean_us_under_pmc"),
    }
    # the kernel sources changed after the counters were collected: the instruction count may describe a kernel
    # that no longer exists (tests/test_gpu_bench.py requires a fresh profile for the default run)
    roof["stale"] = not roof["profile"]["matches_source"]
    return roof


def global_index_bytes(work):
    """Algorithmic bytes of one launch of jaccard_raw_global_kernel on this workload: what the prefix-filter probe has to
    touch whatever the implementation -- every left row (64 B of ids + its size), two offsets per probed id, every posting
    entry of the probed lists' useful classes (4 B each, 8 B when the table has too many rows for the compact entry), and the signature words of the candidates the entry-only tests
    keep (not counted: data dependent, a few per cent of the entries).  Prefix lengths and posting classes as
    csrc/jaccard_raw_global.hip computes them (the same double arithmetic for kmin)."""
    import numpy as np

    width, thr = work.left.width, work.threshold
    kmin = {}
    for s in range(1, 2 * width + 1):
        kmin[s] = next((k for k in range(0, s // 2 + 1) if k / (s - k) >= thr), None)
    prefix = np.zeros(width + 1, dtype=np.int64)
    for a in range(1, width + 1):
        needs = [kmin[a + b] for b in range(1, width + 1) if kmin[a + b] is not None and 1 <= kmin[a + b] <= min(a, b)]
        if needs:
            prefix[a] = a - min(needs) + 1
    longest = int(prefix.max())
    cls_end = 1 if longest <= 1 else 2 if longest <= 2 else 3 if longest <= 4 else 4 if longest <= 8 else 5
    ids = np.asarray(work.left_np, dtype=np.int64)
    big = np.iinfo(np.int64).max
    ids = np.sort(np.where(ids >= 0, ids, big), axis=1)  # ascending, padding last: the table's row order
    cnt = (ids != big).sum(axis=1)
    probe = np.arange(ids.shape[1])[None, :] < prefix[np.minimum(cnt, width)][:, None]
    toks = ids[probe]
    toks = toks[toks < work.right.vocab]
    ps = work.right.post_start.cpu().numpy().astype(np.int64)
    visited = int((ps[5 * toks + cls_end] - ps[5 * toks]).sum())
    entry_bytes = 4 if work.right.post_format == 1 else 8  # (nsm_hip.h, post: RAW tables carry 64-bit entries, format 0 or 2)
    return {"posting_entries_visited": visited, "probes": int(probe.sum()), "posting_entry_bytes": entry_bytes,
            "bytes": visited * entry_bytes + ids.shape[0] * (ids.shape[1] * 4 + 4) + int(probe.sum()) * 8}


def hbm_roofline(profile_name, kernel_match, kernel_label, kernel_ms, algorithmic, n_hits, default_shape=True):
    """Roofline of an inverted-index kernel: no operand is reused on chip (every probe walks its own posting list), so the
    byte model of SURVEY.md 8d applies as written -- achieved = ALGORITHMIC bytes per launch / the kernel's duration (HIP
    events), peak = 8 TB/s HBM3E, traffic = the bytes the counters saw (FETCH_SIZE x 2 + WRITE_SIZE, separate passes)."""
    alg_bytes = algorithmic["bytes"] + n_hits * 16
    roof = {
        "bound": "hbm",
        "kernel": kernel_label,
        "achieved": alg_bytes / (kernel_ms * 1e-3) / 1e9,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": alg_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "traffic": None,
        "kernel_ms": kernel_ms,
        "algorithmic_bytes_per_launch": alg_bytes,
        "posting_entries_visited": algorithmic["posting_entries_visited"],
        "probes": algorithmic["probes"],
        "note": "latency- and request-bound gather kernel: the posting lists (and, for candidates, 8-byte signature words) are "
                "fetched in 64-byte sectors, so the counters see several times the algorithmic bytes",
    }
    pfile = ROOT / "profiles" / f"pmc_{profile_name}.json"
    if not default_shape:
        roof["profile"] = "not applicable: the workload's shape was overridden on the command line"
        return roof
    if not pfile.exists():
        roof["profile"] = f"{pfile.name} missing: run tools/pmc_collect.sh on the GPU box"
        return roof
    prof = json.loads(pfile.read_text())
    entry = next((v for k, v in prof["kernels"].items() if kernel_match in k), None)
    if entry is None:
        roof["profile"] = f"{pfile.name} has no kernel matching {kernel_match!r}"
        return roof
    if "hbm_bytes_per_launch" in entry:
        roof["traffic"] = entry["hbm_bytes_per_launch"]
        roof["hbm_frac"] = roof["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        roof["traffic_over_algorithmic"] = roof["traffic"] / alg_bytes
    roof["valu_issue_frac_pmc"] = entry.get("valu_issue_frac")
    roof["salu_issue_frac"] = entry.get("salu_issue_frac")
    roof["lds_busy_frac"] = entry.get("lds_busy_frac")
    roof["profile"] = {
        "file": f"profiles/{pfile.name}", "git_head": prof.get("git_head"), "csrc_sha256_16": prof.get("csrc_sha256_16"),
        "matches_source": prof.get("csrc_sha256_16") == csrc_hash(), "kernel_us_under_pmc": entry.get("mean_us_under_pmc"),
    }
    roof["stale"] = not roof["profile"]["matches_source"]
    return roof


def cpu_baseline(work, budget_pairs):
    """The oracle's Python restatement of the reference loop on a bounded sample, one core."""
    import numpy as np

    from napkon_string_matching_amd import synthetic
    from oracle import compare as oc

    if work.name in ("c2", "c2low", "c4"):
        side = int(budget_pairs ** 0.5)
        left = synthetic.decode_sets(work.left_np[:side])
        right = synthetic.decode_sets(work.right_np[:side])
        t0 = time.perf_counter()
        hits = oc.raw_grid_hits(left, right, "intersection_vs_union", work.threshold)
        dt = time.perf_counter() - t0
        sample = f"{side}x{side} sub-grid of the same corpus, set-based intersection_vs_union per pair (Python)"
    elif work.name == "term":
        from oracle import native

        side = int(budget_pairs ** 0.5)
        cps = lambda items: [[[ord(c) for c in s] for s in it] for it in items]
        left, right = cps(work.left_np[:side]), cps(work.right_np[:side])
        t0 = time.perf_counter()
        hits = native.levels(True, left, right, work.threshold, None, None, 0, cap=side * side)
        dt = time.perf_counter() - t0
        sample = (f"{side}x{side} sub-grid, compare_terms x Indel ratio per pair by the C oracle (O(nm) LCS DP per level "
                  "pair, scalar C, NOT rapidfuzz, NOT the reference's Python loop)")
    else:
        side = int(budget_pairs ** 0.5)
        left = synthetic.decode_strings(work.left_np[0][:side], work.left_np[1][:side])
        right = synthetic.decode_strings(work.right_np[0][:side], work.right_np[1][:side])
        t0 = time.perf_counter()
        hits = oc.raw_grid_hits(left, right, "fuzzy_match", work.threshold)
        dt = time.perf_counter() - t0
        sample = f"{side}x{side} sub-grid, pure-Python LCS Indel ratio per pair (NOT rapidfuzz)"
    del hits, np
    return {
        "value": side * side / dt,
        "unit": "pair-comparisons/s",
        "cores": 1,
        "kind": "port",
        "sample": sample,
        "seconds": round(dt, 2),
        "host_cpus": os.cpu_count(),
    }


def run_c5(args, comm, device, steps=None, warmup=None, cpu=True, sub=False, words=False):
    """BASELINE configs[4]: three hap / pop / suep shaped cohorts, levels mode (compare_terms), categories
    filtered, intersection_vs_union then fuzzy_match for every cohort pair.  One step = the six grids.
    The left rows of every cohort pair are divided over the ranks (total work fixed: "strong").
    ``words`` = the "c5w" workload: the same shape over 20 000 distinct word-like tokens, at the CACHE threshold 0.5."""
    import numpy as np
    import torch

    from napkon_string_matching_amd import _lib, distributed, grid, synthetic, tables
    rank, world = comm.rank, comm.world
    steps = args.steps if steps is None else steps
    warmup = max(1, args.warmup if warmup is None else warmup)
    lib = _lib.load()
    rows = args.rows or 500_000
    # Thresholds (config.yml:11-12: cache_threshold 0.5, score_threshold 0.7).  The reference's compare() hands gen_comparable
    # the CACHE threshold when one is set (types/comparable_data.py:102-108), so a Matcher run on config.yml scores the
    # grid at 0.5 and filters to 0.7 afterwards.  c5's "t<digits>" level strings share a 12-symbol alphabet: unrelated
    # items score ~0.6 and a 500k x 500k grid at 0.5 would report ~1e10 hits -- an artefact of the digit corpus, so c5 is
    # quoted at the SCORE threshold 0.7 and says so; c5w (word-like text, where unrelated items score ~0.35) runs at 0.5,
    # with its 0.7 run beside it.  As a sub-record of the default run a --threshold meant for the headline is ignored.
    name = "c5w" if words else "c5"
    default_threshold = 0.5 if words else 0.7
    threshold = default_threshold if (args.threshold is None or sub) else args.threshold
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY
    names = ("hap", "pop", "suep")
    lex = synthetic.word_vocabulary(20_000) if words else None
    cohorts = {}
    for k, nm in enumerate(names):  # the same on every rank (seeded); pop / suep carry 1 % near-duplicates of hap
        cohorts[nm] = synthetic.c5_cohort(rows, 11 + k, plant_from=cohorts.get("hap"), lex=lex)
    alphabet = synthetic.c5_alphabet(cohorts["hap"])
    pairs = [("hap", "pop"), ("hap", "suep"), ("pop", "suep")]
    lo, hi = distributed.shard_bounds(rows, rank, world)
    # fuzzy_match operands (join_sorted + default_process of every level), as dense code units: generated
    # vectorised -- the per-item Python strings of round 1 took longer than everything else together
    t0 = time.perf_counter()
    level_codes = {nm: synthetic.c5_level_codes(c) for nm, c in cohorts.items()}
    if args.c5_stride != 64:  # (A/B: the same strings in 128-unit rows -> the multi-word kernels)
        level_codes = {nm: (np.pad(v[0], ((0, 0), (0, args.c5_stride - 64))),) + tuple(v[1:]) for nm, v in level_codes.items()}
    t_generate = time.perf_counter() - t0
    t0 = time.perf_counter()
    grids = []
    for a, b in pairs:
        ca, cb = cohorts[a], cohorts[b]
        lt = tables.SetTable.from_nested_arrays(ca["ids"][lo:hi], ca["plen"][lo:hi], ca["nlev"][lo:hi], "left", device,
                                                categories=ca["cat"][lo:hi], width=16, category_mode=mode,
                                                orig=np.arange(lo, hi, dtype=np.int32))
        rt = tables.SetTable.from_nested_arrays(cb["ids"], cb["plen"], cb["nlev"], "right", device,
                                                categories=cb["cat"], width=16, category_mode=mode)
        codes_a, len_a, first_a, nlev_a = level_codes[a]
        e = ca["entries"]
        left_levels = (codes_a[lo * e: hi * e], len_a[lo * e: hi * e], first_a[lo:hi] - lo * e, nlev_a[lo:hi])
        li, ls, ri, rs = tables.encode_level_codes(left_levels, level_codes[b], len(alphabet), device,
                                                   ca["cat"][lo:hi], cb["cat"], mode, left_offset=lo)
        grids.append(("jaccard", (lt, rt), (lt.struct(), rt.struct()), lt.category_mode))
        grids.append(("indel", (li, ls, ri, rs), (li.struct(), ls.struct(), ri.struct(), rs.struct()), li.category_mode))
    torch.cuda.synchronize(device)
    t_encode = time.perf_counter() - t0

    stream = torch.cuda.current_stream(device).cuda_stream
    jaccard_flags = _lib.FLAG_PRUNE | _lib.FLAG_INDEX
    state = {"threshold": float(threshold), "ws": None}

    def size_workspace():
        # What the product's host does per grid (grid.indel_levels_grid): MEASURE on a sample of the left rows how many pairs
        # outlive step 1 (a scan-only launch, ~1 ms at 500k x 500k), then the split path -- its queue sized to the measurement
        # -- where few do, the shared-tile kernel where many do.  Done once here (the tables do not change between steps);
        # the probe's time is reported beside the step (config.probe_ms_per_step).
        # The split path's survivor queue is the CALLER's (ABI 4): one torch tensor, sized to the largest fuzzy grid and
        # shared by the three (they run one after the other on the launch stream).
        for timed_pass in (False, True):  # (the first pass pays one-off costs: lazy module loads, the library's side stream)
            state["route"] = []
            torch.cuda.synchronize(device)
            t_probe = time.perf_counter()
            for g in grids:
                if g[0] != "indel":
                    continue
                li, ls, ri, rs = g[1]
                extra, expected, measured = (grid.route_one_word(li, ls, ri, rs, state["threshold"], g[3])
                                             if ls.stride == 64 and state["threshold"] > 0 else (0, 0.0, None))
                state["route"].append({"flags": 1 | extra, "expected": expected, "measured": measured})
                state.setdefault("route_of", {})[id(g)] = state["route"][-1]
            torch.cuda.synchronize(device)
            state["probe_ms"] = (time.perf_counter() - t_probe) * 1e3
        indel = [g for g in grids if g[0] == "indel"]
        want = max(int(lib.nsm_indel_levels_workspace_bytes(g[2][0], g[2][1], g[2][2], g[2][3], state["threshold"], r["flags"],
                                                            r["expected"])) for g, r in zip(indel, state["route"]))
        if args.split_workspace_mb >= 0 and want > 0:
            want = args.split_workspace_mb << 20
        state["ws"] = grid.split_workspace(want, device) if want > 0 else None

    def launch(g, b):
        kind, _keep, st, cat_mode = g
        thr = state["threshold"]
        if kind == "jaccard":
            # (the product's host path forces the inverted-index kernel when the vocabulary has >= 8192 tokens --
            # types/comparable_data.py:_levels_grid; the synthetic cohorts draw from 20 000 words)
            rc = lib.nsm_jaccard_levels_grid(st[0], st[1], thr, cat_mode, jaccard_flags, b.records.data_ptr(),
                                             b.capacity, b.count.data_ptr(), stream)
        else:
            ws = state["ws"]
            r = state["route_of"][id(g)]
            rc = lib.nsm_indel_levels_grid(st[0], st[1], st[2], st[3], thr, cat_mode, r["flags"], b.records.data_ptr(),
                                           b.capacity, b.count.data_ptr(), ws.data_ptr() if ws is not None else 0,
                                           ws.numel() * 8 if ws is not None else 0, r["expected"], stream)
        _lib.check(rc, kind + "_levels_grid")

    # hit buffers sized from a counting pass (the counter keeps counting past the capacity: include/nsm_hip.h)
    def probe_capacity():
        b = grid.HitBuffer(1, device)
        most = 0
        for g in grids:
            b.count.zero_()
            launch(g, b)
            most = max(most, int(b.count.item()))
        cap = max(args.capacity, 1 << 16)
        while cap < most + most // 8:
            cap <<= 1
        return cap

    def measure(n_steps, n_warmup):
        size_workspace()
        capacity = probe_capacity()
        bufs = [grid.HitBuffer(capacity, device) for _ in range(2)]
        for b in bufs:
            b.scratch = torch.empty_like(b.records)
        gathered = [torch.empty((world,) + tuple(b.storage.shape), dtype=b.storage.dtype, device=device) for b in bufs]
        pending = [None, None]
        turn = [0]
        counts = []

        def step(record=False):
            for g in grids:
                k = turn[0] & 1
                turn[0] += 1
                b = bufs[k]
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None
                b.count.zero_()
                launch(g, b)
                _lib.check(lib.nsm_sort_hits(b.records.data_ptr(), b.scratch.data_ptr(), b.capacity, b.count.data_ptr(), 0, rows,
                                             stream), "nsm_sort_hits")
                if world > 1:
                    pending[k] = comm.gather_hits(b, gathered[k], async_op=True)
                if record:
                    counts.append(int(b.count.item()))

        def fence():
            for k in (0, 1):
                if pending[k] is not None:
                    pending[k].wait()
                    pending[k] = None
            comm.barrier()
            torch.cuda.synchronize(device)

        for _ in range(n_warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        fence()
        dt = comm.max_seconds(time.perf_counter() - t0)
        step(record=True)
        fence()
        if max(counts) > capacity:
            raise SystemExit(f"hit buffer overflow ({max(counts)} > {capacity}); raise --capacity")

        # dominant kernel: the three fuzzy grids of a step, HIP events on the launch stream
        def kernel_ms(kind, reps):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b = bufs[0]
            torch.cuda.synchronize(device)
            ev0.record()
            for _ in range(reps):
                for g in grids:
                    if g[0] == kind:
                        launch(g, b)
            ev1.record()
            torch.cuda.synchronize(device)
            return ev0.elapsed_time(ev1) / reps

        reps = max(2, min(n_steps, 5))
        out = {"dt": dt, "counts": counts, "ms_indel": kernel_ms("indel", reps), "ms_jac": kernel_ms("jaccard", reps),
               "capacity": capacity, "overflow_word": int(state["ws"][1].item()) & 0xFFFFFFFF if state["ws"] is not None else None}
        del bufs, gathered
        return out

    m = measure(steps, warmup)
    dt, counts, ms_indel, ms_jac = m["dt"], m["counts"], m["ms_indel"], m["ms_jac"]
    split_used = state["ws"] is not None
    pairs_per_step = 2 * len(pairs) * rows * rows  # both score functions over every cohort pair
    local_pairs = len(pairs) * (hi - lo) * rows    # pairs one fuzzy pass of this rank scores
    bytes_per_pair = 2 * 4 * 64                     # both items' level storage: 4 level strings of 64 B each
    str_bytes = sum(t.nbytes() for g in grids if g[0] == "indel" for t in (g[1][1], g[1][3]))
    tiled = (not split_used) and (threshold < 0.55 or any(r["flags"] & _lib.FLAG_TILE for r in state["route"]))
    if split_used:
        kernel_match = ("indel_levels_park_kernel<1, true>", "indel_levels_finish_kernel", "split_begin_kernel")
        kernel_label = ("nsm_indel_levels_grid, split path: indel_levels_park_kernel<1, true> (scan) + "
                        "indel_levels_finish_kernel per round (3 grid calls per step)")
    elif tiled:  # (many pairs outlive step 1: measured by the probe; csrc/indel_levels.hip takes it by itself below 0.55)
        kernel_match = "indel_levels_tile_kernel<1>"
        kernel_label = "nsm_indel_levels_grid, shared-tile kernel: indel_levels_tile_kernel<1> (3 grid calls per step)"
    else:
        kernel_match = "indel_levels_park_kernel<1, false>"
        kernel_label = "nsm_indel_levels_grid, single-kernel path: indel_levels_park_kernel<1> (3 grid calls per step)"
    text = ("20 000 distinct word-like tokens of 3..7 letters (level strings mean 30 / max 63 code units)" if words
            else '"t<digits>" tokens of 20 000 (12-symbol alphabet)')
    result = {
        "metric": "pair-comparisons/sec (whole node), N x M all-pairs",
        "value": pairs_per_step * steps / dt,
        "unit": "pair-comparisons/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int32 (intersection_vs_union) + u64 (fuzzy_match), f64 scores",
        "data": "synthetic",
        "config": {
            "workload": f"{name.upper()}: 3 cohorts x {rows} items (4 levels, ~8 tokens, 1-2 of 32 categories; {text}), "
                        f"compare_terms levels mode, filter_categories, intersection_vs_union then fuzzy_match, threshold {threshold}",
            "mode": "MATCHER",
            "threshold": threshold,
            "threshold_note": ("the cache threshold: what the reference's compare() passes to gen_comparable when one is set "
                               "(types/comparable_data.py:102-108, config.yml:11)" if threshold == 0.5 else
                               "config.yml:12's score threshold; the reference's compare() would hand the grid the cache "
                               "threshold 0.5 -- see c5w for that regime on word-like text" if threshold == 0.7 else "override"),
            "pairs_per_step": pairs_per_step,
            "hits_per_grid_this_rank": counts,
            "hits_per_grid_all_ranks": [comm.sum_int(c) for c in counts],
            "hit_capacity": m["capacity"],
            "sharding": f"left rows of every cohort pair block-sharded over {world} rank(s), right replicated, hits all-gathered",
            "exchange": comm.exchange,
            "rccl_ranks_seen": comm.rccl_ranks,
            "encode_and_h2d_seconds_once": round(t_encode, 2),
            "level_string_generation_seconds_once": round(t_generate, 2),
            "fuzzy_grids_ms_per_step": ms_indel,
            "jaccard_grids_ms_per_step": ms_jac,
            "fuzzy_path": ("split (scan -> survivor queue in a caller-owned workspace -> finish)" if split_used else
                           "shared-tile kernel" if tiled else "single kernel (scan + park + dense finish)"),
            "fuzzy_route_measured": [r["measured"] for r in state["route"]],
            "probe_ms_per_step": state.get("probe_ms"),
            "split_workspace_bytes": int(state["ws"].numel() * 8) if split_used else 0,
            "split_queue_overflowed": bool(m["overflow_word"]) if split_used else None,
        },
        "roofline": valu_roofline(name, kernel_match, kernel_label,
                                  ms_indel, len(pairs), local_pairs // len(pairs) * bytes_per_pair, str_bytes // len(pairs),
                                  default_shape=not args.rows and (args.threshold is None or sub) and world == 1),
    }
    if words and threshold == default_threshold and not args.no_extras:
        # ... and the same grids at config.yml's score threshold (what the result keeps after the cache filter)
        state["threshold"] = 0.7
        m7 = measure(max(1, min(steps, 2)), 1)
        result["at_score_threshold"] = {
            "threshold": 0.7, "steps": max(1, min(steps, 2)), "ms_per_step": m7["dt"] / max(1, min(steps, 2)) * 1e3,
            "fuzzy_grids_ms_per_step": m7["ms_indel"], "jaccard_grids_ms_per_step": m7["ms_jac"],
            "hits_per_grid_this_rank": m7["counts"],
            "fuzzy_path": "split" if state["ws"] is not None else "single kernel (scan + park + dense finish)",
            "split_queue_overflowed": bool(m7["overflow_word"]) if state["ws"] is not None else None,
        }
        state["threshold"] = float(threshold)
    if rank == 0 and world == 1 and cpu and not args.no_cpu_baseline:
        from oracle import compare as oc
        from oracle import score_functions as osf

        side = 300
        la = synthetic.c5_level_token_lists(cohorts["hap"], slice(0, side))
        lb = synthetic.c5_level_token_lists(cohorts["pop"], slice(0, side))
        ca, cb = cohorts["hap"]["cat"][:side], cohorts["pop"]["cat"][:side]
        t0 = time.perf_counter()
        kept = 0
        for func in (osf.intersection_vs_union, osf.fuzzy_match):
            for i in range(side):
                for j in range(side):
                    if (int(ca[i]) & int(cb[j])) or (not ca[i] and not cb[j]):  # categories_matching first (:213-221)
                        kept += oc.compare_terms(la[i], lb[j], func) >= threshold
        dt_cpu = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": 2 * side * side / dt_cpu, "unit": "pair-comparisons/s", "cores": 1, "kind": "port",
            "sample": f"{side}x{side} hap x pop corner, category predicate then compare_terms per pair, both score functions "
                      "(Python; fuzzy_match = pure-Python LCS, NOT rapidfuzz)",
            "seconds": round(dt_cpu, 2), "host_cpus": os.cpu_count(),
        }
    return result


def run_raw(args, comm, device, name, rows=0, right_rows=0, threshold=None, steps=None, warmup=None, extras=True,
            cpu_scale=1.0):
    """One RAW-mode / term workload under the contract: W warm-up steps, K timed steps between
    barrier + synchronize fences, MAX over the ranks.  Returns the JSON object."""
    import torch

    from napkon_string_matching_amd import _lib, grid

    rank, world = comm.rank, comm.world
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    capacity = args.capacity
    if name in ("c4", "c2low") and capacity == 1 << 13:
        capacity = 1 << 16  # ~12k hits at 1M x 1M; ~53k at 50k x 50k and threshold 0.1
    if name == "term" and capacity == 1 << 13:
        capacity = 1 << 24  # ~2.6 % of the pairs reach the cache threshold 0.5
    work = Workload(name, comm, rows, device, right_rows, threshold, getattr(args, "id_range", 0) if name in ("c2", "c2low", "c4") else 0)
    # the committed profile describes the workload's default per-GPU grid (c4 divides its rows over the ranks)
    default_shape = (not rows and not right_rows and threshold is None and (name != "c4" or world == 1)
                     and not getattr(args, "id_range", 0))
    # two hit buffers: the all-gather of step k overlaps the grid kernel of step k+1 (RCCL runs on
    # its own stream; the buffer is only reused after its gather has completed)
    bufs = [grid.HitBuffer(capacity, device) for _ in range(2)]
    for b in bufs:
        b.scratch = torch.empty_like(b.records)
    gathered = [torch.empty((world,) + tuple(b.storage.shape), dtype=b.storage.dtype, device=device) for b in bufs]
    pending = [None, None]
    buf = bufs[0]
    stream = torch.cuda.current_stream(device).cuda_stream
    lib = work.lib
    step_no = [0]
    id_limit = max(comm.sum_int(work.n), work.m)  # every reported row id is below it (nsm_sort_hits: fewer key bits)

    # The three dependent launches of a step (zero the counter, grid kernel, hit ordering) are captured
    # once per hit buffer and mode into a hipGraph and replayed: one submission per step.
    graphs = {}

    def step_graph(k, prune):
        key = (k, prune)
        if key not in graphs:
            b = bufs[k]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                s = torch.cuda.current_stream(device).cuda_stream
                b.count.zero_()
                work.launch(b, s, prune)
                lib.nsm_sort_hits(b.records.data_ptr(), b.scratch.data_ptr(), b.capacity, b.count.data_ptr(), 0, id_limit, s)
            graphs[key] = g
        return graphs[key]

    # N = 1: the hit ordering of step k runs on a second stream and overlaps the grid kernel of step k + 1 (the two steps
    # use different hit buffers; a buffer is only reused after its ordering has finished) -- the same pipelining the
    # N > 1 path applies to the all-gather.  Every step's full work still happens inside the timed region.
    side = torch.cuda.Stream(device) if (world == 1 and not args.graph and not args.serial_sort) else None
    grid_done = [torch.cuda.Event() for _ in range(2)]
    ordered = [torch.cuda.Event() for _ in range(2)]
    used = [False, False]

    def step(prune=True):
        k = step_no[0] & 1
        step_no[0] += 1
        b = bufs[k]
        if pending[k] is not None:
            pending[k].wait()
            pending[k] = None
        if args.graph:
            step_graph(k, prune).replay()
        elif side is not None:
            main = torch.cuda.current_stream(device)
            if used[k]:
                main.wait_event(ordered[k])  # the buffer's previous ordering has finished
            b.count.zero_()
            work.launch(b, stream, prune)
            grid_done[k].record(main)
            side.wait_event(grid_done[k])
            _lib.check(lib.nsm_sort_hits(b.records.data_ptr(), b.scratch.data_ptr(), b.capacity, b.count.data_ptr(), 0, id_limit,
                                         side.cuda_stream), "nsm_sort_hits")
            ordered[k].record(side)
            used[k] = True
        else:
            b.count.zero_()
            work.launch(b, stream, prune)
            _lib.check(lib.nsm_sort_hits(b.records.data_ptr(), b.scratch.data_ptr(), b.capacity, b.count.data_ptr(), 0, id_limit,
                                         stream), "nsm_sort_hits")
        if world > 1:
            pending[k] = comm.gather_hits(b, gathered[k], async_op=True)

    def fence():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        comm.barrier()
        torch.cuda.synchronize(device)

    def timed(n_steps, prune):
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(prune)
        fence()
        return comm.max_seconds(time.perf_counter() - t0)

    for _ in range(warmup):
        step(True)
    dt = timed(steps, True)
    n_hits = int(buf.count.item())
    if n_hits > buf.capacity:
        raise SystemExit(f"hit buffer overflow ({n_hits} > {buf.capacity}); raise --capacity")
    pairs_per_step = comm.sum_int(work.n) * work.m  # (the last shard of a strong-scaled grid may be shorter)
    value = pairs_per_step * steps / dt

    # ---- per-kernel duration of the dominant kernel, HIP events on the launch stream
    def kernel_ms(prune, reps):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        buf.count.zero_()
        work.launch(buf, stream, prune)
        torch.cuda.synchronize(device)
        ev0.record()
        for _ in range(reps):
            work.launch(buf, stream, prune)
        ev1.record()
        torch.cuda.synchronize(device)
        return ev0.elapsed_time(ev1) / reps

    k_ms = kernel_ms(True, max(3, steps))
    alg_bytes = work.n * work.m * work.bytes_per_pair
    compulsory = work.left.nbytes() + work.right.nbytes() + n_hits * 16
    result = {
        "metric": "pair-comparisons/sec (whole node), N x M all-pairs",
        "value": value,
        "unit": "pair-comparisons/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True,
        "scaling": work.scaling,
        "vs_baseline": None,
        "dtype": work.dtype,
        "data": "synthetic",
        "config": {
            "workload": work.label,
            "score_func": "fuzzy_match" if work.name in ("c3", "term") else "intersection_vs_union",
            "mode": "MATCHER" if work.name == "term" else "RAW",
            "threshold": work.threshold,
            "pairs_per_step": pairs_per_step,
            "hits_per_rank": n_hits,
            "hits_all_ranks": comm.sum_int(n_hits),
            "sharding": f"left rows block-sharded over {world} rank(s), right replicated, hits all-gathered",
            "exchange": comm.exchange,
            "rccl_ranks_seen": comm.rccl_ranks,
            "exact_prune": True,
            "hit_ordering": ("second stream, overlapped with the next step's grid kernel (two hit buffers)" if side is not None
                             else "launch stream"),
            "encode_and_h2d_seconds_once": round(work.encode_h2d_seconds, 4),
        },
        "roofline": (hbm_roofline(work.name, work.kernel_match, work.kernel + " (candidates from the right table's global "
                                  "inverted index)", k_ms, global_index_bytes(work), n_hits, default_shape=default_shape)
                     if work.name in ("c2", "c2low", "c4") else
                     valu_roofline(work.name, work.kernel_match, work.kernel + " (exact prune on)", k_ms, 1, alg_bytes,
                                   compulsory, default_shape=default_shape)),
    }
    if extras and not args.no_extras:
        # exhaustive variant (prune off): fewer repetitions, it is the slow one
        ex_steps = max(2, min(steps, 5))
        step(False)
        dt_ex = timed(ex_steps, False)
        k_ms_ex = kernel_ms(False, ex_steps)
        ex_roof = valu_roofline(work.name, work.kernel_match_exhaustive, work.kernel, k_ms_ex, 1, alg_bytes, compulsory,
                                default_shape=default_shape)
        result["exhaustive"] = {
            "note": "same grid, exact prune disabled: every pair's full comparison is evaluated",
            "value": pairs_per_step * ex_steps / dt_ex,
            "ms_per_step": dt_ex / ex_steps * 1e3,
            "kernel_ms": k_ms_ex,
            "valu_issue_frac": ex_roof.get("frac"),
            "lds_busy_frac": ex_roof.get("lds_busy_frac"),
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        budget = {"c3": 80_000, "term": 360_000}.get(work.name, 20_000_000)
        result["cpu_baseline"] = cpu_baseline(work, int(budget * cpu_scale))
    return result


def main():
    args = parse_args()
    import torch

    if args.post_format is not None:
        from napkon_string_matching_amd import tables as _tables

        _tables.COMPACT_POSTINGS = args.post_format != 0
        _tables.RAW_POST_FORMAT = args.post_format or 2

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the match loop has no CPU fallback")
    local_rank = local_rank % torch.cuda.device_count()  # rehearsal: several ranks may share a GPU
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    comm = Comm(args, rank, world, device)
    assert comm.world == args.gpus

    def release():
        import gc

        gc.collect()
        torch.cuda.empty_cache()

    if args.workload in ("c5", "c5w"):
        result = run_c5(args, comm, device, words=args.workload == "c5w")
    elif args.workload != "all":
        result = run_raw(args, comm, device, args.workload, args.rows, args.right_rows, args.threshold)
    else:
        # the headline: BASELINE configs[2], the largest configuration quoted on one GPU ...
        result = run_raw(args, comm, device, "c3", args.rows, args.right_rows, args.threshold)
        # ... and every other workload under the same clock, attached as sub-records.  N > 1: the configs
        # BASELINE.json quotes on 8 GPUs (c4, c5; c5w = c5 on word-like text), left rows divided over the ranks.
        names = () if args.no_c4 else (("c2", "c2low", "c4", "c5", "c5w", "term") if world == 1 else ("c4", "c5", "c5w"))
        t_all = time.perf_counter()
        mirror = {}
        for name in names:
            release()
            t_sub = time.perf_counter()
            if name in ("c5", "c5w"):
                sub = run_c5(args, comm, device, steps=max(1, min(args.steps, 5 if name == "c5" else 3)), warmup=1, sub=True,
                             words=name == "c5w")
            else:
                # c2low / c4 share c2's CPU restatement (same function, same corpus generator): a short sample
                sub = run_raw(args, comm, device, name, args.rows, 0 if name == "c4" else args.right_rows, None,
                              extras=world == 1, cpu_scale=0.25 if name in ("c2low", "c4") else 1.0)
            rec = {k: sub[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "scaling", "dtype", "config",
                                       "roofline", "exhaustive", "cpu_baseline", "at_score_threshold") if k in sub}
            rec["workload"] = sub["config"]["workload"]
            rec["kernel_ms"] = sub["roofline"]["kernel_ms"]
            rec["wall_seconds_incl_setup"] = round(time.perf_counter() - t_sub, 1)
            result[name] = rec
            # the same numbers as plain scalars under `config` (see the module docstring: config.sub)
            mirror[name] = {
                "ms_per_step": rec["ms_per_step"], "value": rec["value"], "steps": rec["steps"], "kernel_ms": rec["kernel_ms"],
                "roofline_frac": rec["roofline"].get("frac"),
                "roofline_kernel": rec["roofline"].get("kernel"),
                "exhaustive_ms_per_step": rec.get("exhaustive", {}).get("ms_per_step"),
                "cpu_baseline_value": rec.get("cpu_baseline", {}).get("value"),
                "threshold": sub["config"].get("threshold"),
                "hits": sub["config"].get("hits_all_ranks", sub["config"].get("hits_per_grid_all_ranks")),
            }
            if "at_score_threshold" in rec:
                mirror[name]["ms_per_step_at_0.7"] = rec["at_score_threshold"]["ms_per_step"]
        if names:
            result["sub_records_wall_seconds"] = round(time.perf_counter() - t_all, 1)
            result["config"]["sub"] = mirror
    if world > 1 and comm.dev_group is not None:
        assert comm.rccl_ranks == world, f"RCCL saw {comm.rccl_ranks} ranks of {world}"
    if rank == 0:
        print(json.dumps(result))
    comm.close()


if __name__ == "__main__":
    main()
