"""bench.py contract on a real MI355X: one JSON line with the fields the driver reads, and the RCCL leg
(process group, weight-arena broadcast, barriers, max-reduce) exercised with a single rank under
torch.distributed.run — the multi-GPU launch the driver uses, minus the GPUs this box does not have."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"]


def run(cmd, env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_process():
    d = run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--ctx", "96", "--cpu-seconds", "2"], {})
    for k in REQUIRED + ["cpu_baseline"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["prefill"]["prompt_tokens"] == 95 and d["prefill"]["prompt_tokens_per_s"] > 0


def test_bench_rccl_leg_with_one_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", "bench.py", "--gpus", "1", "--steps", "1", "--warmup", "1", "--ctx", "64", "--no-cpu-baseline"]
    d = run(cmd, {"ZGPT2_FORCE_DIST": "1"})
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["weight_broadcast_ms"] is not None and d["weight_broadcast_ms"] >= 0
    # one rank: the reference IS the job — efficiency near 1 (two separately timed 64-step generations of one prompt)
    assert d["scaling_reference"]["n_gpus"] == 1 and d["scaling_reference"]["value"] > 0
    assert 0.7 < d["scaling_efficiency"] < 1.4, d["scaling_efficiency"]
    assert d["per_rank_tokens_per_s"]["min"] == d["per_rank_tokens_per_s"]["max"] > 0
    assert d["weight_broadcast_GBps"] is None or d["weight_broadcast_GBps"] > 0
