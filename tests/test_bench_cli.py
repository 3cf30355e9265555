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


def _load_bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_state_sharded_leg_failure_only_costs_its_field():
    """The secondary c5 leg of the default line runs in a child process group.  Without a GPU the child ends
    with an error: the function must hand back an `error` field (rc and stderr tail) and nothing else."""
    import argparse

    bench = _load_bench()
    env_backup = os.environ.pop("RYDIFF_BENCH_STANDIN", None)
    try:
        out = bench.c5_leg_in_child_group(argparse.Namespace(variant=0), 1)
    finally:
        if env_backup is not None:
            os.environ["RYDIFF_BENCH_STANDIN"] = env_backup
    assert set(out) == {"error"} and "rc" in out["error"]


def test_state_sharded_leg_is_killed_as_a_group_on_timeout(tmp_path, monkeypatch):
    """A leg that hangs (here: a stand-in 'interpreter' that starts a grandchild and sleeps) is killed with its whole process
    group after the time limit; the caller gets an `error` field and carries on — no os._exit from a GPU-initialised rank."""
    import argparse
    import time

    bench = _load_bench()
    pidfile = tmp_path / "grandchild.pid"
    fake = tmp_path / "hang.sh"
    fake.write_text(f"#!/bin/bash\nsleep 300 &\necho $! > {pidfile}\nsleep 300\n")
    fake.chmod(0o755)
    monkeypatch.setattr(bench.sys, "executable", str(fake))
    monkeypatch.setattr(bench, "C5_LEG_TIMEOUT_S", 2.0)
    t0 = time.time()
    out = bench.c5_leg_in_child_group(argparse.Namespace(variant=0), 2)
    assert set(out) == {"error"} and "killed" in out["error"]
    assert time.time() - t0 < 30
    grandchild = int(pidfile.read_text())
    time.sleep(0.5)
    alive = Path(f"/proc/{grandchild}").exists() and "Z" not in Path(f"/proc/{grandchild}/stat").read_text().split()[2]
    assert not alive
