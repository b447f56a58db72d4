"""bench.py's world handling: `--gpus N` can neither print a line for a smaller world nor need a launcher.
CPU only: the refusal, the launcher's command line, and the relaunch itself with `--rendezvous-only` (gloo ranks that count each other).
Reference shape for N > 1: one job per distro fanned out by a cron, units/crons.go:303-332."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _env(**over):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(over)
    return env


def test_launch_plan_decisions():
    assert bench.launch_plan(1, None, 1) == ("run", None)
    assert bench.launch_plan(1, 1, 1) == ("run", None)
    assert bench.launch_plan(8, 8, 8) == ("run", None)                     # under the driver's torchrun
    assert bench.launch_plan(4, None, 8, single_process=True) == ("run", None)
    what, msg = bench.launch_plan(8, None, 1)
    assert what == "refuse" and "needs 8 devices" in msg
    what, msg = bench.launch_plan(8, 1, 8)                                  # WORLD_SIZE=1 exported by something else
    assert what == "refuse" and "WORLD_SIZE" in msg
    what, msg = bench.launch_plan(2, 4, 8)
    assert what == "refuse"
    what, argv = bench.launch_plan(4, None, 8)
    assert what == "launch" and "torch.distributed.run" in argv and argv[argv.index("--nproc-per-node") + 1] == "4"
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[-1].endswith("bench.py")


def test_env_world_reads_empty_as_unset(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "")
    assert bench.env_world() is None
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert bench.env_world() == 4
    monkeypatch.delenv("WORLD_SIZE")
    assert bench.env_world() is None


def test_gpus_2_without_devices_exits_non_zero():
    """`WORLD_SIZE= python bench.py --gpus 2` on a box without two devices: no JSON line, a message that says what is missing."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("this box has two devices")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=_env(WORLD_SIZE=""), timeout=300)
    assert r.returncode != 0
    assert "needs 2 devices" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def test_gpus_2_relaunches_itself_and_both_ranks_answer():
    """The launcher path end to end on the CPU: `python bench.py --gpus 2 --rendezvous-only` re-executes under torch.distributed.run
    and rank 0 prints how many ranks its all-reduce saw."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    obj = json.loads(lines[0])
    assert obj["n_gpus"] == 2 and obj["ranks_seen"] == 2
    assert "re-executing under torch.distributed.run" in r.stderr


def test_deadline_fires_once_and_a_claimed_line_is_not_printed_twice():
    import time
    d, fired = bench.Deadline(), []
    d.arm(0.05, lambda: fired.append("a"))
    time.sleep(0.3)
    assert fired == ["a"] and not d.claim()          # the timer took the line: the main thread must not print another
    d2, fired2 = bench.Deadline(), []
    d2.arm(0.05, lambda: fired2.append("a"))
    d2.cancel()
    time.sleep(0.2)
    assert fired2 == [] and d2.claim() and not d2.claim()
    d3, fired3 = bench.Deadline(), []
    d3.arm(5.0, lambda: fired3.append("first"))      # re-armed for the next part of the run: only the later timer is alive
    d3.arm(0.05, lambda: fired3.append("second"))
    time.sleep(0.3)
    assert fired3 == ["second"]
    d4, fired4 = bench.Deadline(), []
    assert d4.claim()                                # the line went out before the time was up: the timer has nothing to do
    d4.arm(0.05, lambda: fired4.append("late"))
    time.sleep(0.2)
    assert fired4 == []


def test_a_collective_that_never_returns_costs_the_deadline_not_the_line():
    """Two gloo ranks, rank 1 stays away from the second all-reduce: torch.distributed would sit there (and PyTorch's watchdog would end
    the process without a line); the run's Deadline prints the unmeasured line from rank 0 and ends every rank with code 3."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--stall-rank", "1", "--tick-deadline-s", "3"],
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    obj = json.loads(lines[0])
    assert obj["value"] is None and obj["n_gpus"] == 2 and "did not reach its line" in obj["error"]
    assert obj["config"]["multi"]["rccl_ranks_seen"] == 2
