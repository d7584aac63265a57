"""`python bench.py --gpus N` launches its own ranks (the plain command must work for N in {1, 2, 4, 8}).
Here on the CPU with --dry-run: everything but the GPU work -- the parent spawning one process per rank before anything
touches a GPU, the gloo process group on 127.0.0.1, the parameter-block broadcast, the contiguous partition, the
reductions, ONE JSON line relayed from rank 0, the worst exit code."""
import json
import os
import subprocess
import sys
import time

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
    # the record itself shows N ranks took part, and it carries the 1080p window-19 leg and the batched chain at EVERY N
    # (north_star: 640x480 AND 1920x1080 batches at 1/2/4/8 GPUs), reduced like the headline: all units / MAX time over ranks
    assert j["ranks_seen"] == n and len(j["devices"]) == n and sorted(d["rank"] for d in j["devices"]) == list(range(n))
    assert len({d["pid"] for d in j["devices"]}) == n                   # one process per GPU
    fhd, chain = j["roofline"]["fhd_w19"], j["also"]["vga_chain_batch64"]
    assert fhd["n_gpus"] == n and chain["n_gpus"] == n
    assert fhd["k1_ms_per_rank"] == [10.0 * (r + 1) for r in range(n)] and fhd["k1_avg_launch_ms"] == 10.0 * n
    assert abs(fhd["process_mpix_s"] - n * 32 * 1920 * 1080 * 3 / (0.03 * n) / 1e6) < 1e-6
    assert abs(chain["batched_mpix_s"] - n * 64 * 640 * 480 / (1.7 * n) / 1e3) < 1e-6
    assert j["rccl"]["error"] is None and "REPLICAS ONLY" not in j["config"]["sharding"]


@pytest.mark.parametrize("n", [2, 4])
def test_rccl_failure_falls_back_in_process_and_is_flagged(n):
    """SURVEY 8(e) "Fallback": RCCL does not come up -> the SAME processes go on over gloo (no re-exec, no restart), the
    parameter blocks every rank formed itself are compared, the line carries a number and says [REPLICAS ONLY] + the error"""
    r = _run(["--gpus", str(n), "--dry-run", "--force-rccl-failure"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["replicas_only"] is True and j["rccl"]["ok"] is False and j["rccl"]["wanted"] is True
    assert "forced by --force-rccl-failure" in j["rccl"]["error"] and j["rccl"]["backend_used"].startswith("gloo")
    assert "[REPLICAS ONLY]" in j["config"]["sharding"] and "forced" in j["config"]["sharding"]
    assert j["rccl"]["ranks_whose_block_differs_from_rank0"] == 0
    # everything else is as in the healthy run: N ranks, contiguous blocks, the reductions
    assert j["ranks_seen"] == n and j["checksum"]["frames"] == 64 * n and len({d["pid"] for d in j["devices"]}) == n
    assert j["roofline"]["fhd_w19"]["n_gpus"] == n


def test_strong_scaling_cuts_a_fixed_batch():
    """BASELINE config 4 as literally written: 512 frames sharded N ways (--total-frames), next to the weak-scaling default"""
    r = _run(["--gpus", "4", "--dry-run", "--total-frames", "512"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert j["scaling"] == "strong" and j["checksum"]["frames"] == 512 and j["config"]["frames_per_gpu"] == 128
    assert j["checksum"]["sum_first_frames"] == 0 + 128 + 256 + 384


def test_ranks_that_hang_are_ended_at_the_deadline(tmp_path):
    """the parent's overall deadline: ranks that never finish are killed (exact PIDs) and the exit code is non-zero"""
    env = {"KDE_BENCH_TEST_HANG": "1"}
    t0 = time.time()
    r = _run(["--gpus", "2", "--dry-run", "--launch-timeout", "6"], env=env, timeout=120)
    assert r.returncode != 0 and "did not finish within --launch-timeout" in r.stderr
    assert time.time() - t0 < 60


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
