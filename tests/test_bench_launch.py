"""bench.py's launch contract (CPU): `python bench.py --gpus N` starts N ranks itself or fails loudly -- it never prints a
1-GPU line for an N-GPU request (round-2 VERDICT, missing 1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return e


def test_more_ranks_than_devices_exits_non_zero():
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices are present")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode != 0
    assert "HIP device(s) visible" in (out.stdout + out.stderr) and "{" not in out.stdout      # a clear message, no JSON line


def test_world_size_must_match_the_request():
    e = _env()
    e.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode != 0 and "WORLD_SIZE = 1" in (out.stdout + out.stderr)


def test_self_launch_spawns_one_rank_per_requested_gpu():
    """the same relaunch path the GPU run takes (torch.distributed.run on 127.0.0.1), with gloo and no device work"""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-spawn"], capture_output=True, text=True, timeout=300, env=_env())
    assert out.returncode == 0, out.stdout + out.stderr
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["selftest_spawn"] and r["n_gpus"] == 2 and r["ranks_in_all_reduce"] == 2
    # the per-rank fields an N-rank line carries, gathered over the group by the function bench_icp uses
    assert r["rccl_ranks"] == 2
    assert r["kernel_ms_per_step_per_rank"] == pytest.approx([0.1, 0.2])
    assert r["allreduce_us_per_iteration"] == pytest.approx(20.0) and r["allreduce_us_per_iteration_per_rank"] == pytest.approx([10.0, 20.0])
    assert r["host_enqueue_us_per_iteration_per_rank"] == pytest.approx([1.0, 2.0])
