"""bench.py's launch contract without a GPU: `--gpus N` with no launcher must start N ranks itself (a fresh
torch.distributed.run child before any GPU call) and report n_gpus == N; a rank count that disagrees with --gpus is an
error, never a silent single-GPU run.  --dry-run rehearses rendezvous, sharding, the weight broadcast and the max-reduce
over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_2_launches_two_ranks_by_itself():
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], cwd=ROOT,
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["scaling"] == "weak"
    assert d["weight_broadcast_ms"] is not None and d["weight_broadcast_ms"] >= 0
    assert d["config"]["global_prompts"] == 16 and d["config"]["prompts_per_gpu"] == 8  # BASELINE configs[2]: 8 per GPU
    # the N > 1 line is self-contained: the one-GPU figure with the same per-GPU work, the efficiency it implies, the spread
    ref = d["scaling_reference"]
    assert ref["n_gpus"] == 1 and ref["value"] > 0
    assert abs(d["scaling_efficiency"] - d["value"] / (2 * ref["value"])) < 1e-3
    pr = d["per_rank_tokens_per_s"]
    assert 0 < pr["min"] <= pr["max"]
    assert d["value"] <= 2 * pr["max"] * 1.001  # the job is as slow as its slowest rank
    assert d["weight_broadcast_GBps"] is not None and d["weight_broadcast_GBps"] > 0


def test_rank_count_must_match_gpus():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--dry-run"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE=1" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
