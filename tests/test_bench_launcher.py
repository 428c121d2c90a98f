"""CPU: `python bench.py --gpus N` starts its own ranks (no torchrun): the plan it would execute, a real two-rank rendezvous over gloo,
and a failing rank taking the others down with its exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    e.update(kw)
    return e


def test_launch_plan():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "7", "--warmup", "2", "--launch-dry-run"], capture_output=True, text=True, env=_env(), timeout=120)
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    ranks = plan["ranks"]
    assert len(ranks) == 4
    ports = {x["env"]["MASTER_PORT"] for x in ranks}
    assert len(ports) == 1
    for k, x in enumerate(ranks):
        e = x["env"]
        assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["MASTER_ADDR"]) == (str(k), str(k), "4", "127.0.0.1")
        assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert x["cmd"][1] == BENCH and "--launch-dry-run" not in x["cmd"] and x["cmd"][2:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def test_launcher_starts_ranks_that_find_each_other():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=_env(RS_BENCH_LAUNCH_SELFTEST="1"), timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0's line only
    assert json.loads(lines[0]) == {"selftest": 3, "world": 2, "gpus": 2}


def test_launcher_hands_on_a_failing_rank():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=_env(RS_BENCH_LAUNCH_SELFTEST="fail1"), timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "rank 1 exited with 3" in r.stderr


def test_launcher_ends_ranks_blocked_in_a_collective():
    """Round 6: a rank that dies AFTER the group has formed, while the others are blocked inside the next collective (the failure mode
    of a multi-GPU run: an out-of-memory or a fault on one rank mid-step): the launcher notices, stops the blocked ranks by PID —
    SIGTERM, SIGKILL after 5 s — and returns the dead rank's exit code.  No hang: the whole thing ends well inside the timeout."""
    import time
    t = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, env=_env(RS_BENCH_LAUNCH_SELFTEST="die1"), timeout=240)
    assert r.returncode == 5, (r.returncode, r.stderr[-600:])
    assert "rank 1 exited with 5" in r.stderr
    assert time.time() - t < 120
    assert not [l for l in r.stdout.strip().splitlines() if l.startswith("{")]      # no result line from a run that failed
