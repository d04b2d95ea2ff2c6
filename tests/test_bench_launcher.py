"""`python bench.py --gpus N` must start its own ranks: the launcher is exercised on CPU through
--launch-check (every rank joins a gloo group on the host; no GPU work), and a failing rank must fail the whole command."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env)


def test_launcher_starts_n_ranks_and_relays_one_json_line():
    r = _run("--gpus", "3", "--launch-check")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0]) == {"launch_check": True, "n_gpus": 3, "max_rank": 2}


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a host without GPUs: there every rank fails at device setup")
def test_launcher_fails_when_a_rank_fails():
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "exited with code" in r.stderr
    assert not r.stdout.strip()
