"""bench.py's contract (one JSON line with the fields the driver reads) and a rehearsal of its
multi-rank control flow on a single GPU (gloo, ranks sharing cuda:0)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _last_json(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields(device):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "3",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3 and d["unit"] == "frames/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["tracked_ok"] == d["frames_total"] == 12
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0.05 < r["frac"] < 1.0
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])


def test_two_rank_control_flow_rehearsal(device):
    env = dict(os.environ, PXT_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)  # rank 0 only prints
    assert d["n_gpus"] == 2 and d["frames_total"] == 20 and d["tracked_ok"] == 20
    assert abs(d["value"] - 20 / (10 * d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    assert d["cpu_baseline"]["value"] is None  # timed on rank 0 at N = 1 only
