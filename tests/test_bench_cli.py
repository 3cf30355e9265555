"""bench.py's launcher and sharding logic without a GPU: `--gpus 2` must start two ranks itself (gloo, a CPU stand-in for the
solver: RYDIFF_BENCH_STANDIN=1), deal the c4 parameter sets over them, gather the results and print ONE JSON line that says
n_gpus == 2.  The numbers of a stand-in run mean nothing; the plumbing is what is tested."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(*argv, env_extra=None):
    env = dict(os.environ, RYDIFF_BENCH_STANDIN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_spawns_the_ranks_and_reports_them():
    r = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "10", "--chunk", "3", "--time-steps", "8")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2
    assert out["config"]["workload"].startswith("c4")          # the multi-rank default workload
    assert out["scaling"] == "strong"
    assert out["config"]["trajectories_total"] == 10 and out["config"]["trajectories_this_rank"] == 5
    assert out["config"]["gathered_parameter_sets"] == 10       # all_gather of the per-set gradients reached every rank
    assert out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert "standin" in out["data"]


def test_single_rank_default_is_the_headline_workload():
    r = _run("--steps", "1", "--warmup", "0", "--time-steps", "8")
    assert r.returncode == 0, r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"]["workload"].startswith("c3") and out["scaling"] == "weak"


def test_rank_count_must_match_the_flag():
    r = _run("--gpus", "4", env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must match" in (r.stderr + r.stdout)
