"""bench.py --gpus N: a request for more GPUs than the machine has, or for a rank count that
differs from the launcher's, must fail loudly instead of timing fewer GPUs; with two GPUs present
the two-rank run (RCCL between two processes) must produce a complete line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _device_count():
    import torch
    return torch.cuda.device_count()        # does not initialise the GPU


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=600)


def test_more_gpus_than_present_is_refused():
    n = _device_count() + 1
    r = _run(["--gpus", str(max(n, 2)), "--steps", "1", "--warmup", "0", "--scale", "0.01", "--no-cpu-baseline"])
    assert r.returncode == 2 and "not falling back" in r.stderr and r.stdout.strip() == ""


def test_rank_count_must_match_the_launcher():
    r = _run(["--gpus", "2", "--steps", "1"], WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_two_rank_run_over_rccl():
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.05", "--cpu-scale", "0.005"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0
    assert line["multi_gpu"]["collectives_per_solve"] > 0
    one = json.loads(_run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--scale", "0.05", "--no-cpu-baseline"]).stdout.strip().splitlines()[-1])
    assert abs(line["final_cost"] - one["final_cost"]) <= 1e-9 * one["final_cost"]
