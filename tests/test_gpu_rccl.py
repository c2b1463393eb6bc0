"""The RCCL call path on one GPU: a one-rank process group whose collectives are forced to run."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.gpu
def test_collectives_run_through_rccl_at_world_size_one():
    """Flat gradient all-reduce, parameter broadcast, sharded standardisation, the float64 / int64 all-to-all of
    the partial top-k lists, the gathers of merged blocks, ragged rows and embeddings: each through RCCL in a
    fresh child process, each equal to the single-process result bit for bit."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py"), str(_free_port())],
                         env=_env(), capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rccl world1 ok" in out.stdout


@pytest.mark.gpu
def test_bench_takes_the_distributed_branches_on_one_gpu():
    """`bench.py --gpus 1` with MAGIC_BENCH_FORCE_DIST=1: RCCL process group of one rank, gradient exchange inside
    the timed step, sharded KNN classes -- the code the multi-GPU launch runs -- on a small workload."""
    import json
    env = _env()
    env["MAGIC_BENCH_FORCE_DIST"] = "1"
    env["MASTER_PORT"] = str(_free_port())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--scale", "0.02", "--knn-steps", "1", "--no-cpu-baseline", "--no-entity", "--no-configs",
                          "--rows-file", os.devnull],
                         env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["knn"]["value"] > 0
    assert line["config"]["parallelism"].startswith("dp1")
