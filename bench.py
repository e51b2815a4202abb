#!/usr/bin/env python3
"""Headline benchmark: pair-comparisons/sec of the all-pairs match loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--no-cpu-baseline]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(one process per GPU, RCCL).  One "step" = one pass of the hot path over one synthetic batch whose
operand tables are already resident in HBM: zero the hit counter, score the whole N x M pair grid
on the GPU (threshold test included), order the hits canonically and -- for N > 1 -- all-gather
the (score, i, j) hits of every rank.  Left rows are block-sharded over the ranks, the right side is
replicated; per-GPU work is fixed as N grows ("weak").

Workload at N = 1 is BASELINE.json configs[1] ("c2": 50k x 50k token-id sets, mean 8 ids,
intersection_vs_union, threshold 0.5); `--workload c3` runs configs[2] (200k x 200k strings,
fuzzy_match, threshold 0.8).  `--workload c4` is configs[3] (1M x 1M token-id sets, threshold 0.8, the
1M left rows divided over the ranks: "strong"); `--workload c5` is configs[4] (three 500k-item
cohorts, 4 levels, 32 categories, filter_categories, both score functions back to back through the
levels-mode kernels, left rows of every cohort pair divided over the ranks: "strong").

The JSON line carries, besides the driver's contract fields:
  roofline      dominant kernel; achieved = algorithmic bytes (128 B per pair) per
                launch / average kernel duration measured with HIP events on the launch stream
  exhaustive    the same grid with the exact prune disabled (every pair's matrix evaluated)
  limiter       the pruned kernel against its real bound, the VALU issue rate of the per-row filter
  cpu_baseline  the oracle's restatement of the reference's Python loop, 1 core, bounded sample
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "napkon-string-matching_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_PAIR = 128  # two 64-byte operand rows


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("c2", "c3", "c4", "c5", "term"), default="c2")
    ap.add_argument("--rows", type=int, default=0, help="override rows per side per GPU (debug)")
    ap.add_argument("--right-rows", type=int, default=0, help="override the right side's rows (debug)")
    ap.add_argument("--threshold", type=float, default=None, help="override the workload's threshold (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", type=int, default=0, help="1: replay each step as one captured hipGraph")
    ap.add_argument("--capacity", type=int, default=1 << 13)
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="gloo = CPU-staged rehearsal of the N > 1 path (e.g. 2 ranks sharing one GPU)")
    return ap.parse_args()


class Workload:
    """Operand tables in HBM + the launch closure of one workload."""

    def __init__(self, name, rank, world, rows, device, right_rows=0, threshold=None):
        import numpy as np
        import torch

        from napkon_string_matching_amd import _lib, synthetic, tables

        self.name = name
        self.lib = _lib.load()
        self.flag_prune = _lib.FLAG_PRUNE
        self.scaling = "weak"
        if name == "c4":  # configs[3]: the C2 generator at 1M x 1M, threshold 0.8, left rows divided over the ranks
            total = rows or 1_000_000
            right_rows = right_rows or total
            rows = -(-total // world)
            threshold = 0.8 if threshold is None else threshold
            self.scaling = "strong"
        if name in ("c2", "c4"):
            n = rows or 50_000
            m = right_rows or rows or 50_000
            self.threshold = 0.5 if threshold is None else threshold
            # global left corpus = world * n rows; this rank scores rows [rank*n, (rank+1)*n)
            left = synthetic.token_sets(n, 1234 + 1000 * rank)
            right = synthetic.token_sets(m, 5678)
            if rank == 0:
                right_planted = synthetic.plant_near_duplicate_sets(left, right, 5679)
            else:
                right_planted = None
            self.host = (left, right, right_planted)
            self.left_np = left
            right = self._share_right(right_planted, right, device, world)
            self.right_np = right
            orig = np.arange(n, dtype=np.int32) + rank * n
            t_enc = time.perf_counter()
            self.left = tables.SetTable.from_padded(left, "left", device, orig=orig)
            self.right = tables.SetTable.from_padded(right, "right", device)
            torch.cuda.synchronize(device)
            self.encode_h2d_seconds = time.perf_counter() - t_enc
            self.launch_fn = self.lib.nsm_jaccard_raw_grid
            self.kernel = "jaccard_raw_kernel<16>"
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
            self.kernel = f"indel_levels_kernel<{ls.stride // 64}>"
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
            packed = self._share_right(packed, np.zeros((m, 68), np.uint8), device, world)
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
            self.kernel = "indel_raw_kernel"
            self.dtype = "u64"
            self.label = f"C3: {n}x{m} strings/GPU (len U[16,64], 37 symbols), fuzzy_match RAW, threshold {self.threshold}"
        self.n, self.m = n, m
        self.ls, self.rs = self.left.struct(), self.right.struct()
        self.bytes_per_pair = BYTES_PER_PAIR
        if name == "term":  # both items' level strings (mean levels x row stride)
            li, ls, ri, rs = self.term_tables
            self.bytes_per_pair = int(round((ls.n / max(1, li.n) + rs.n / max(1, ri.n)) * ls.stride))

    @staticmethod
    def _share_right(planted, like, device, world):
        """The right side is replicated: rank 0's array is broadcast over RCCL."""
        import torch

        if world == 1:
            return planted
        import torch.distributed as dist

        t = torch.from_numpy(planted if planted is not None else like.copy())
        if dist.get_backend() == "nccl":
            t = t.to(device)
        dist.broadcast(t, src=0)
        return t.cpu().numpy()

    def launch(self, buf, stream, prune=True):
        from napkon_string_matching_amd import _lib

        flags = self.flag_prune if prune else 0
        if self.name == "term":  # prune off = every step scored wave-wide (no parking, no histogram bound)
            st = self.term_structs
            flags = self.flag_prune if prune else _lib.FLAG_WAVE_WIDE
            _lib.check(
                self.lib.nsm_indel_levels_grid(st[0], st[1], st[2], st[3], float(self.threshold), _lib.CAT_NONE, flags,
                                               buf.records.data_ptr(), buf.capacity, buf.count.data_ptr(), stream),
                self.kernel,
            )
            return
        _lib.check(
            self.launch_fn(self.ls, self.rs, float(self.threshold), flags, buf.records.data_ptr(), buf.capacity,
                           buf.count.data_ptr(), stream),
            self.kernel,
        )


def gather_hits(buf, out, world, device, async_op=False):
    """all-gatherv of the (score, i, j) hits.  RCCL has no native gatherv; the hit buffer carries its
    own counter in a trailing record (grid.HitBuffer), so ONE all-gather of the max-padded storage
    moves records and counts together.  Returns the work handle when ``async_op``."""
    import torch
    import torch.distributed as dist

    if dist.get_backend() != "nccl":  # gloo rehearsal: stage through host memory
        host = torch.empty((world * buf.storage.shape[0], 2), dtype=buf.storage.dtype)
        dist.all_gather_into_tensor(host, buf.storage.cpu())
        out.copy_(host.view_as(out))
        return None
    return dist.all_gather_into_tensor(out.view(-1, 2), buf.storage, async_op=async_op)


def cpu_baseline(work, budget_pairs):
    """The oracle's Python restatement of the reference loop on a bounded sample, one core."""
    import numpy as np

    from napkon_string_matching_amd import synthetic
    from oracle import compare as oc

    if work.name in ("c2", "c4"):
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


def run_c5(args, rank, world, device, dist):
    """BASELINE configs[4]: three hap / pop / suep shaped cohorts, levels mode (compare_terms), categories
    filtered, intersection_vs_union then fuzzy_match for every cohort pair.  One step = the six grids.
    The left rows of every cohort pair are divided over the ranks (total work fixed: "strong")."""
    import numpy as np
    import torch

    from napkon_string_matching_amd import _lib, distributed, grid, synthetic, tables
    from napkon_string_matching_amd.compare import score_functions as sf

    lib = _lib.load()
    rows = args.rows or 500_000
    threshold = 0.7 if args.threshold is None else args.threshold  # max(cache 0.5, score 0.7), config.yml:11-12
    mode = _lib.CAT_INTERSECT_OR_BOTH_EMPTY
    names = ("hap", "pop", "suep")
    cohorts = {}
    for k, nm in enumerate(names):  # the same on every rank (seeded); pop / suep carry 1 % near-duplicates of hap
        cohorts[nm] = synthetic.c5_cohort(rows, 11 + k, plant_from=cohorts.get("hap"))
    pairs = [("hap", "pop"), ("hap", "suep"), ("pop", "suep")]
    lo, hi = distributed.shard_bounds(rows, rank, world)
    t0 = time.perf_counter()
    level_strings = {nm: [[sf.fuzzy_operand(lv) for lv in it] for it in synthetic.c5_level_token_lists(c)]
                     for nm, c in cohorts.items()}
    grids = []
    for a, b in pairs:
        ca, cb = cohorts[a], cohorts[b]
        lt = tables.SetTable.from_nested_arrays(ca["ids"][lo:hi], ca["plen"][lo:hi], ca["nlev"][lo:hi], "left", device,
                                                categories=ca["cat"][lo:hi], width=16, category_mode=mode,
                                                orig=np.arange(lo, hi, dtype=np.int32))
        rt = tables.SetTable.from_nested_arrays(cb["ids"], cb["plen"], cb["nlev"], "right", device,
                                                categories=cb["cat"], width=16, category_mode=mode)
        li, ls, ri, rs = tables.encode_level_strings(level_strings[a][lo:hi], level_strings[b], device, ca["cat"][lo:hi],
                                                     cb["cat"], mode, left_offset=lo)
        grids.append(("jaccard", (lt, rt), (lt.struct(), rt.struct()), lt.category_mode))
        grids.append(("indel", (li, ls, ri, rs), (li.struct(), ls.struct(), ri.struct(), rs.struct()), li.category_mode))
    torch.cuda.synchronize(device)
    t_encode = time.perf_counter() - t0

    capacity = max(args.capacity, 1 << 16)
    bufs = [grid.HitBuffer(capacity, device) for _ in range(2)]
    for b in bufs:
        b.scratch = torch.empty_like(b.records)
    gathered = [torch.empty((world,) + tuple(b.storage.shape), dtype=b.storage.dtype, device=device) for b in bufs]
    pending = [None, None]
    stream = torch.cuda.current_stream(device).cuda_stream
    turn = [0]
    counts = []

    def launch(g, b):
        kind, _keep, st, cat_mode = g
        if kind == "jaccard":
            rc = lib.nsm_jaccard_levels_grid(st[0], st[1], float(threshold), cat_mode, 1, b.records.data_ptr(), b.capacity,
                                             b.count.data_ptr(), stream)
        else:
            rc = lib.nsm_indel_levels_grid(st[0], st[1], st[2], st[3], float(threshold), cat_mode, 1, b.records.data_ptr(),
                                           b.capacity, b.count.data_ptr(), stream)
        _lib.check(rc, kind + "_levels_grid")

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
            lib.nsm_sort_hits(b.records.data_ptr(), b.scratch.data_ptr(), b.capacity, b.count.data_ptr(), stream)
            if world > 1:
                pending[k] = gather_hits(b, gathered[k], world, device, async_op=True)
            if record:
                counts.append(int(b.count.item()))

    def fence():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(max(1, args.warmup)):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    step(record=True)
    fence()
    if max(counts) > capacity:
        raise SystemExit(f"hit buffer overflow ({max(counts)} > {capacity}); raise --capacity")

    # dominant kernel: indel_levels_kernel (the three fuzzy grids of a step), HIP events on the launch stream
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

    reps = max(2, min(args.steps, 5))
    ms_indel, ms_jac = kernel_ms("indel", reps), kernel_ms("jaccard", reps)
    pairs_per_step = 2 * len(pairs) * rows * rows  # both score functions over every cohort pair
    local_pairs = len(pairs) * (hi - lo) * rows    # pairs one fuzzy pass of this rank scores
    bytes_per_pair = 2 * 4 * 64                     # both items' level storage: 4 level strings of 64 B each
    achieved = local_pairs * bytes_per_pair / (ms_indel * 1e-3) / 1e9
    result = {
        "metric": "pair-comparisons/sec (whole node), N x M all-pairs",
        "value": pairs_per_step * args.steps / dt,
        "unit": "pair-comparisons/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": max(1, args.warmup),
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int32 (intersection_vs_union) + u64 (fuzzy_match), f64 scores",
        "data": "synthetic",
        "config": {
            "workload": f"C5: 3 cohorts x {rows} items (4 levels, ~8 tokens, 1-2 of 32 categories), compare_terms levels mode, "
                        f"filter_categories, intersection_vs_union then fuzzy_match, threshold {threshold}",
            "mode": "MATCHER",
            "threshold": threshold,
            "pairs_per_step": pairs_per_step,
            "hits_per_grid_this_rank": counts,
            "sharding": f"left rows of every cohort pair block-sharded over {world} rank(s), right replicated, hits all-gathered",
            "encode_and_h2d_seconds_once": round(t_encode, 2),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "indel_levels_kernel<1> (3 launches per step)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "kernel_ms": ms_indel,
            "algorithmic_bytes_per_pair": bytes_per_pair,
            "jaccard_levels_kernel_ms": ms_jac,
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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
    if rank == 0:
        print(json.dumps(result))


def main():
    args = parse_args()
    import torch

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
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
        if args.dist_backend == "nccl":
            from datetime import timedelta

            try:  # RCCL; one tiny collective up front so that a broken transport shows here, not mid-run
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device,
                                        timeout=timedelta(seconds=300))
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe)
                torch.cuda.synchronize(device)
                assert int(probe.item()) == world
            except Exception as exc:  # keep the run alive: the same exchange staged through host memory
                print(f"[bench] RCCL unavailable ({type(exc).__name__}: {exc}); falling back to the gloo-staged exchange",
                      file=sys.stderr)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                args.dist_backend = "gloo"
                dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from napkon_string_matching_amd import grid

    if args.workload == "c4" and args.capacity == 1 << 13:
        args.capacity = 1 << 16  # ~12k hits at 1M x 1M
    if args.workload == "term" and args.capacity == 1 << 13:
        args.capacity = 1 << 24  # ~2.3 % of the pairs reach the cache threshold 0.5
    if args.workload == "c5":
        run_c5(args, rank, world, device, dist)
        if world > 1:
            dist.destroy_process_group()
        return
    work = Workload(args.workload, rank, world, args.rows, device, args.right_rows, args.threshold)
    # two hit buffers: the all-gather of step k overlaps the grid kernel of step k+1 (RCCL runs on
    # its own stream; the buffer is only reused after its gather has completed)
    bufs = [grid.HitBuffer(args.capacity, device) for _ in range(2)]
    for b in bufs:
        b.scratch = torch.empty_like(b.records)
    gathered = [torch.empty((world,) + tuple(b.storage.shape), dtype=b.storage.dtype, device=device) for b in bufs]
    pending = [None, None]
    buf = bufs[0]
    stream = torch.cuda.current_stream(device).cuda_stream
    lib = work.lib
    step_no = [0]

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
                lib.nsm_sort_hits(b.records.data_ptr(), b.scratch.data_ptr(), b.capacity, b.count.data_ptr(), s)
            graphs[key] = g
        return graphs[key]

    def step(prune=True):
        k = step_no[0] & 1
        step_no[0] += 1
        b = bufs[k]
        if pending[k] is not None:
            pending[k].wait()
            pending[k] = None
        if args.graph:
            step_graph(k, prune).replay()
        else:
            b.count.zero_()
            work.launch(b, stream, prune)
            lib.nsm_sort_hits(b.records.data_ptr(), b.scratch.data_ptr(), b.capacity, b.count.data_ptr(), stream)
        if world > 1:
            pending[k] = gather_hits(b, gathered[k], world, device, async_op=True)

    def drain():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None

    def fence():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed(steps, prune):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(prune)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    for _ in range(args.warmup):
        step(True)
    dt = timed(args.steps, True)
    n_hits = int(buf.count.item())
    if n_hits > buf.capacity:
        raise SystemExit(f"hit buffer overflow ({n_hits} > {buf.capacity}); raise --capacity")
    pairs_per_step = work.n * work.m * world
    value = pairs_per_step * args.steps / dt

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

    k_ms = kernel_ms(True, max(3, args.steps))
    # exhaustive variant (prune off): fewer repetitions, it is the slow one
    ex_steps = max(2, min(args.steps, 5))
    step(False)
    dt_ex = timed(ex_steps, False)
    k_ms_ex = kernel_ms(False, ex_steps)

    def roof(ms):
        achieved = work.n * work.m * work.bytes_per_pair / (ms * 1e-3) / 1e9
        return achieved

    traffic = None
    tfile = ROOT / "profiles" / f"traffic_{work.name}.json"  # measured by tools/refresh_profiles.sh
    if tfile.exists():
        try:
            traffic = json.loads(tfile.read_text()).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "pair-comparisons/sec (whole node), N x M all-pairs",
        "value": value,
        "unit": "pair-comparisons/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
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
            "sharding": f"left rows block-sharded over {world} rank(s), right replicated, hits all-gathered",
            "exchange": None if world == 1 else ("rccl all-gather" if args.dist_backend == "nccl" else "gloo, staged through host memory"),
            "exact_prune": True,
            "encode_and_h2d_seconds_once": round(work.encode_h2d_seconds, 4),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": work.kernel + " (exact prune on)",
            "achieved": roof(k_ms),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": roof(k_ms) / HBM_PEAK_GBPS,
            "traffic": traffic,
            "hbm_measured_GBps": None if traffic is None else traffic / (k_ms * 1e-3) / 1e9,
            "kernel_ms": k_ms,
            "algorithmic_bytes_per_launch": work.n * work.m * work.bytes_per_pair,
            "compulsory_hbm_bytes_per_launch": work.left.nbytes() + work.right.nbytes() + n_hits * 16,
        },
        "exhaustive": {
            "note": "same grid, exact prune disabled: every pair's full comparison is evaluated",
            "value": pairs_per_step * ex_steps / dt_ex,
            "ms_per_step": dt_ex / ex_steps * 1e3,
            "kernel_ms": k_ms_ex,
            "achieved_GBps": roof(k_ms_ex),
            "frac": roof(k_ms_ex) / HBM_PEAK_GBPS,
        },
    }
    # The honest limiter of the pruned kernel is the VALU issue rate of its per-row filter, not HBM (DESIGN.md
    # 4.0/4.1/4.3): filter instructions per (wavefront, left row) x their measured issue cycles
    # (profiles/r01_valu_issue_rates_gfx950.txt) against the kernel's measured duration.
    filter_cycles = 8 * 4.3 + 2.3 if work.name == "c3" else 4 * 4.3 + 2.3  # 8 v_sad_u8 + v_or | 2 v_and + 2 v_bcnt + v_and
    wave_rows = -(-work.m // 64) * work.n
    ideal_ms = wave_rows * filter_cycles / (256 * 4 * 2.4e9) * 1e3
    result["limiter"] = {
        "bound": "valu_issue",
        "note": "filter cycles per (wavefront, left row) x all wave-rows / (1024 SIMDs x 2.4 GHz); size-class skips "
                "make the true work smaller, survivors and loop overhead make it larger",
        "filter_cycles_per_wave_row": filter_cycles,
        "ideal_kernel_ms": ideal_ms,
        "frac": ideal_ms / k_ms,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(work, {"c3": 80_000, "term": 360_000}.get(work.name, 20_000_000))
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
