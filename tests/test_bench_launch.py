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


@pytest.mark.parametrize("n", [1, 2, 4, 8])
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


def test_config4_as_written_512_frames_8_ways():
    """BASELINE config 4 literally: 512 frames over 8 ranks = 64 each, the run the driver makes on the 8-GPU node"""
    r = _run(["--gpus", "8", "--dry-run", "--total-frames", "512"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert j["n_gpus"] == 8 and j["ranks_seen"] == 8 and len({d["pid"] for d in j["devices"]}) == 8
    assert j["scaling"] == "strong" and j["checksum"]["frames"] == 512 and j["config"]["frames_per_gpu"] == 64
    assert j["checksum"]["sum_first_frames"] == sum(64 * k for k in range(8))


def test_an_empty_shard_is_refused_by_every_rank_before_any_collective():
    """ceil(T/N) blocks: 5 frames over 4 ranks are 2, 2, 1, 0 -- every rank sees that itself and leaves at once"""
    t0 = time.time()
    r = _run(["--gpus", "4", "--dry-run", "--total-frames", "5"])
    assert r.returncode != 0 and "without a frame" in r.stderr and "[3]" in r.stderr
    assert time.time() - t0 < 25          # no rank waited in a barrier for the one that left


def _line(r):
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_a_probe_that_never_returns_costs_the_default_deadline_and_a_flag_not_the_record():
    """RCCL bring-up that HANGS on one of eight ranks.  The launcher tries RCCL first in disposable
    probe processes; the one that sleeps forever is killed at the default 60 s deadline and the real ranks never touch RCCL:
    a full line, flagged [REPLICAS ONLY] with the probe's reason, well inside the driver's 600 s limit"""
    t0 = time.time()
    j = _line(_run(["--gpus", "8", "--dry-run"], env={"KDE_RCCL_PROBE_TEST": "hang:5"}, timeout=200))
    el = time.time() - t0
    assert 55 < el < 90, el
    assert j["replicas_only"] is True and j["rccl"]["ok"] is False and j["rccl"]["timeout_s"] == 60.0
    assert j["rccl"]["probe"]["ok"] is False and "rank 5" in j["rccl"]["probe"]["reason"] and "deadline" in j["rccl"]["probe"]["reason"]
    assert j["rccl"]["probe"]["by"] == "launcher" and j["rccl"]["bring_up_s"] is None       # RCCL never touched by the ranks
    assert "[REPLICAS ONLY]" in j["config"]["sharding"] and "rank 5" in j["config"]["sharding"]
    assert j["ranks_seen"] == 8 and j["checksum"]["frames"] == 64 * 8


def test_a_failed_probe_ends_the_others_at_once():
    t0 = time.time()
    j = _line(_run(["--gpus", "4", "--dry-run"], env={"KDE_RCCL_PROBE_TEST": "fail:2"}))
    assert time.time() - t0 < 30 and j["replicas_only"] is True
    assert "rank 2 exit code 3" in j["rccl"]["error"] and j["rccl"]["probe"]["seconds"] < 10


def _torchrun(n, port, extra, env):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run"] + extra,
                          capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))


def test_under_a_launcher_every_rank_probes_for_itself():
    """the driver's command form: no parent of ours, so every rank starts the probe of its own rank before its first HIP call
    and the verdicts are exchanged over gloo; a probe that hangs is killed at --rccl-timeout"""
    t0 = time.time()
    j = _line(_torchrun(4, 29535, ["--rccl-timeout", "5"], {"KDE_RCCL_PROBE_TEST": "hang:3"}))
    assert time.time() - t0 < 60
    assert j["replicas_only"] is True and j["rccl"]["probe"]["by"] == "every rank, for itself"
    assert "rank 3" in j["rccl"]["probe"]["reason"] and j["rccl"]["bring_up_s"] is None


def test_the_in_process_bring_up_is_bounded_too():
    """clean probes, then new_group / the first all-reduce hangs in the rank itself (helper thread): the main thread gives up
    at the deadline, the ranks agree over gloo, the stuck rank leaves through os._exit after the line is out"""
    t0 = time.time()
    j = _line(_torchrun(2, 29536, ["--rccl-timeout", "5"], {"KDE_RCCL_PROBE_TEST": "ok", "KDE_RCCL_INPROC_TEST": "hang:1"}))
    assert time.time() - t0 < 60
    assert j["replicas_only"] is True and j["rccl"]["probe"]["ok"] is True and 4.5 < j["rccl"]["bring_up_s"] < 12
    assert "abandoned" in j["rccl"]["error"]


def test_an_asymmetric_failure_does_not_wait_for_the_deadline():
    """RCCL fails on ONE rank only -- it says so in the gloo store, the healthy ranks stop waiting at once"""
    j = _line(_torchrun(2, 29537, ["--rccl-timeout", "40"], {"KDE_RCCL_INPROC_TEST": "fail:1"}))
    assert j["replicas_only"] is True and j["rccl"]["bring_up_s"] < 10
    assert "rank 1" in j["rccl"]["error"] and "forced by KDE_RCCL_INPROC_TEST" in j["rccl"]["error"]


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
