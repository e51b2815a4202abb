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
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["c2", "c3", "c4", "c5"])
def test_bench_line(workload):
    rows = {"c2": "8000", "c3": "8000", "c4": "60000", "c5": "6000"}[workload]
    d = _line([sys.executable, "bench.py", "--workload", workload, "--rows", rows, "--steps", "2", "--warmup", "1"])
    assert all(k in d for k in CONTRACT), sorted(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["unit"] == "pair-comparisons/s" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and "traffic" in roof
    cpu = d["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and cpu["sample"]
    assert d["scaling"] == ("weak" if workload in ("c2", "c3") else "strong")


def test_bench_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--rows", "6000", "--dist-backend", "gloo"])
    assert d["n_gpus"] == 2 and d["config"]["pairs_per_step"] == 2 * 6000 * 6000 and "cpu_baseline" not in d
