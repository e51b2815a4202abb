"""bench.py's contract on the GPU box: one JSON line with the driver's fields, the roofline and
cpu_baseline objects; the N = 2 path (two ranks sharing the GPU, gloo-staged exchange) launched the way
the driver launches N > 1."""
import json
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def _line(cmd):
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_roofline(roof, need_frac):
    assert "traffic" in roof and roof["kernel_ms"] > 0 and roof["algorithmic_bytes_per_launch"] > 0
    if roof["bound"] == "hbm":  # the inverted-index kernels: SURVEY 8d's byte model (no on-chip operand reuse)
        assert roof["peak"] == pytest.approx(8000.0) and roof["unit"] == "GB/s"
        assert 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
        if need_frac:
            assert roof["profile"]["file"].startswith("profiles/pmc_") and roof["traffic"] > 0
            assert 0.0 < roof["hbm_frac"] <= 1.0
        return
    assert roof["bound"] == "valu_issue" and roof["peak"] == pytest.approx(1228.8)
    if need_frac:
        assert roof["profile"]["file"].startswith("profiles/pmc_")
    if roof["frac"] is not None:  # a fraction of a bound: never above 1
        assert 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
        assert 0.0 < roof["frac_pmc"] <= 1.0
        assert roof["hbm_frac"] is None or 0.0 <= roof["hbm_frac"] <= 1.0
    else:
        assert not need_frac, roof


@pytest.mark.parametrize("workload", ["c2", "c2low", "c3", "c4", "c5", "c5w", "term"])
def test_bench_line(workload):
    rows = {"c2": "8000", "c2low": "8000", "c3": "8000", "c4": "60000", "c5": "6000", "c5w": "6000", "term": "3000"}[workload]
    d = _line([sys.executable, "bench.py", "--workload", workload, "--rows", rows, "--steps", "2", "--warmup", "1"])
    assert all(k in d for k in CONTRACT), sorted(d)
    assert not any(k in d for k in ("c2", "c2low", "c3", "c4", "c5", "c5w", "term"))  # one workload alone: no sub-records
    assert "sub" not in d["config"]
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["unit"] == "pair-comparisons/s" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"] and "limiter" not in d
    _check_roofline(d["roofline"], need_frac=False)  # reduced grid: the committed profile does not apply
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and cpu["sample"]
    assert d["scaling"] == ("weak" if workload in ("c2", "c2low", "c3", "term") else "strong")
    if workload == "c5w":  # the cache threshold 0.5 is what the reference's flow hands to the grid; 0.7 rides along
        assert d["config"]["threshold"] == 0.5 and d["at_score_threshold"]["threshold"] == 0.7
        assert d["at_score_threshold"]["ms_per_step"] > 0 and d["config"]["fuzzy_path"] == "shared-tile kernel"
    if workload == "c5":
        assert d["config"]["threshold"] == 0.7 and d["config"]["fuzzy_path"].startswith("split")
        assert d["config"]["split_queue_overflowed"] is False and d["config"]["split_workspace_bytes"] > 0


def test_bench_default_roofline_is_a_fraction():
    """The default invocation's roofline is a fraction of a bound the kernel can approach: in (0, 1], backed by
    the committed counter profile (round-1 verdict: 91.6 of HBM peak is not a roofline).  The headline is the largest
    single-GPU config (C3); every other BASELINE config, configs[4] on word-like text and the reference's default
    configuration ride along as sub-records under the same contract, and their numbers are mirrored as plain scalars
    into config.sub (round-3 verdict: the driver's parser keeps `config`, not unknown top-level keys)."""
    d = _line([sys.executable, "bench.py", "--steps", "5", "--warmup", "2"])
    _check_roofline(d["roofline"], need_frac=True)
    assert d["roofline"]["stale"] == (not d["roofline"]["profile"]["matches_source"])
    assert d["exhaustive"]["valu_issue_frac"] is None or 0.0 < d["exhaustive"]["valu_issue_frac"] <= 1.0
    assert "C3" in d["config"]["workload"] and d["cpu_baseline"]["value"] > 0 and d["dtype"] == "u64"
    mirror = d["config"]["sub"]
    assert sorted(mirror) == ["c2", "c2low", "c4", "c5", "c5w", "term"]
    for name, steps in (("c2", 5), ("c2low", 5), ("c4", 5), ("c5", 5), ("c5w", 3), ("term", 5)):
        sub = d[name]
        assert sub["steps"] == steps and sub["value"] > 0 and sub["ms_per_step"] > 0 and sub["kernel_ms"] > 0, name
        assert sub["unit"] == "pair-comparisons/s" and name.upper() in sub["workload"].upper()
        _check_roofline(sub["roofline"], need_frac=True)
        assert sub["cpu_baseline"]["kind"] == "port" and sub["cpu_baseline"]["value"] > 0
        if name not in ("c5", "c5w"):
            assert sub["exhaustive"]["ms_per_step"] >= 0.9 * sub["ms_per_step"]
        # the timed region fits the wall time the record reports for itself
        assert sub["steps"] * sub["ms_per_step"] * 1e-3 < sub["wall_seconds_incl_setup"]
        m = mirror[name]
        assert m["ms_per_step"] == sub["ms_per_step"] and m["value"] == sub["value"] and m["kernel_ms"] == sub["kernel_ms"]
        assert m["roofline_frac"] == sub["roofline"]["frac"] and m["cpu_baseline_value"] == sub["cpu_baseline"]["value"]
        assert m["steps"] == steps
        if "exhaustive" in sub:
            assert m["exhaustive_ms_per_step"] == sub["exhaustive"]["ms_per_step"]
    assert d["c5w"]["config"]["threshold"] == 0.5 and d["c5w"]["at_score_threshold"]["threshold"] == 0.7
    assert mirror["c5w"]["ms_per_step_at_0.7"] == d["c5w"]["at_score_threshold"]["ms_per_step"]
    assert d["c5"]["config"]["split_queue_overflowed"] is False
    assert d["sub_records_wall_seconds"] < 480


def test_bench_rccl_failure_is_loud():
    """Two ranks on ONE GPU cannot form an RCCL communicator: bench.py must agree on that across the ranks and
    exit non-zero (round-1 advice: a silent, rank-local gloo fallback), and run staged when --allow-gloo is given."""
    def launch(extra):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        return subprocess.run(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "4000", *extra],
            cwd=ROOT, capture_output=True, text=True, timeout=900)

    out = launch([])
    if out.returncode == 0:  # a box where two ranks may share a device: then the exchange must really be RCCL
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert line["config"]["exchange"] == "rccl all-gather"
        return
    assert "RCCL could not be brought up on every rank" in out.stderr
    out = launch(["--allow-gloo"])
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["config"]["exchange"].startswith("gloo") and line["n_gpus"] == 2


def test_bench_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--rows", "6000", "--dist-backend", "gloo"])
    assert d["n_gpus"] == 2 and d["config"]["pairs_per_step"] == 2 * 6000 * 6000 and "cpu_baseline" not in d
    # N > 1 carries the configs BASELINE.json quotes on 8 GPUs (c5w = c5 on word-like text), left rows divided over the ranks
    assert set(("c4", "c5", "c5w")) <= set(d) and "c2" not in d and "term" not in d and "C3" in d["config"]["workload"]
    assert sorted(d["config"]["sub"]) == ["c4", "c5", "c5w"]
    for name in ("c4", "c5", "c5w"):
        sub = d[name]
        assert sub["scaling"] == "strong" and sub["value"] > 0 and sub["config"]["exchange"].startswith("gloo")
        assert "rccl_ranks_seen" in sub["config"] and "cpu_baseline" not in sub
    # the ranks' hit counts add up to what ONE rank finds on the same (seeded) grid
    one = _line([sys.executable, "bench.py", "--workload", "c4", "--rows", "6000", "--steps", "1", "--warmup", "1",
                 "--no-cpu-baseline"])
    assert d["c4"]["config"]["hits_all_ranks"] == one["config"]["hits_per_rank"] > 0
    one = _line([sys.executable, "bench.py", "--workload", "c5", "--rows", "6000", "--steps", "1", "--warmup", "1",
                 "--no-cpu-baseline"])
    assert d["c5"]["config"]["hits_per_grid_all_ranks"] == one["config"]["hits_per_grid_this_rank"]
    one = _line([sys.executable, "bench.py", "--workload", "c5w", "--rows", "6000", "--steps", "1", "--warmup", "1",
                 "--no-cpu-baseline"])
    assert d["c5w"]["config"]["hits_per_grid_all_ranks"] == one["config"]["hits_per_grid_this_rank"]
