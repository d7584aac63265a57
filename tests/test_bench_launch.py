"""`python bench.py --gpus N` launches its own ranks (the plain command must work for N in {1, 2, 4, 8}).
Here on the CPU with --dry-run: everything but the GPU work -- the parent spawning one process per rank before anything
touches a GPU, the gloo process group on 127.0.0.1, the parameter-block broadcast, the contiguous partition, the
reductions, ONE JSON line relayed from rank 0, the worst exit code."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(extra, env=None, timeout=240):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **(env or {})))
    return r


@pytest.mark.parametrize("n", [1, 2, 4])
def test_plain_command_launches_n_ranks(n):
    r = _run(["--gpus", str(n), "--dry-run", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                    # ONE JSON line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["checksum"]["frames"] == 64 * n and j["replicas_only"] is False
    assert j["checksum"]["sum_first_frames"] == sum(64 * k for k in range(n))      # contiguous blocks of 64
    assert abs(j["max_dt_over_ranks"] - 1e-3 * n) < 1e-9                # MAX over ranks


def test_under_a_launcher_the_same_script_is_a_rank():
    """what the driver does for N > 1: torch.distributed.run starts the ranks, bench.py must not spawn again"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2", "--dry-run"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                               "MASTER_PORT": "29534"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
