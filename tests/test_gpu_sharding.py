"""The N > 1 path with real filtering: two ranks (own processes, gloo group, both on cuda:0 -- the box has one GPU)
each filter their block of the batch through the HIP library; the gathered result must be bit-identical to one
process filtering the whole batch.  And the C++ multi-device host (examples/shard_replay: one thread per device,
ncclBroadcast of the parameter block) with its own partition-independence check."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT
from gpu_util import dev, host

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n_frames", [7, 4])
def test_two_ranks_filter_their_shards_bit_identically(torch_cuda, synth, tmp_path, n_frames):
    from kinectdepthmapenhancement_amd import filters as F
    port, out = _free_port(), str(tmp_path / "gathered.npy")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), str(r), "2", str(port),
                               str(n_frames), "300", out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    logs = [p.communicate(timeout=240)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    gathered = np.load(out)
    # the same batch in ONE process
    bgr, depth = synth.make_batch(300, n_frames, 160, 120)
    p = F.JointBilateralFilter.default_params()
    p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = 11, 3.0, 7.65, 20.0
    jbf = F.JointBilateralFilter(160, 120, p, max_batch=n_frames)
    single = host(jbf.process_batch(dev(torch_cuda, depth), dev(torch_cuda, bgr)))
    assert gathered.shape == single.shape
    assert np.array_equal(gathered.view(np.uint32), single.view(np.uint32))      # frames are independent units: bitwise


def test_cpp_shard_replay_with_rccl_broadcast(torch_cuda):
    exe = os.path.join(ROOT, "examples", "shard_replay")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    r = subprocess.run([exe, "--frames", "6", "--width", "160", "--height", "120", "--steps", "2", "--verify"],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["verified"] is True and line["tables_match_rank0"] is True
    assert line["params_broadcast"].startswith("rccl") and line["devices"] >= 1 and line["frames"] == 6
    assert all(len(d["pci_bus_id"]) >= 7 for d in line["per_device"])


def test_cpp_shard_replay_goes_on_without_rccl(torch_cuda):
    """SURVEY 8(e) fallback in the C++ host: no communicator -> same process, every device thread forms the block itself"""
    exe = os.path.join(ROOT, "examples", "shard_replay")
    r = subprocess.run([exe, "--frames", "6", "--width", "160", "--height", "120", "--steps", "2", "--verify", "--force-rccl-failure"],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["params_broadcast"].startswith("replicas only") and "forced" in line["params_broadcast"]
    assert line["verified"] is True and line["tables_match_rank0"] is True and line["mpixels_per_s"] > 0


def test_bench_py_launches_two_ranks_itself_and_equals_two_single_runs(torch_cuda):
    """`python bench.py --gpus 2` (no launcher) spawns its two ranks; on this one-GPU box both use cuda:0 over a gloo group.
    128 frames, and the checksum of the filtered output equals the sum of the two shards run as N = 1 jobs."""
    bench = os.path.join(ROOT, "bench.py")
    common = ["--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extra", "--no-verify", "--wakeup-ms", "0"]

    def run(extra):
        r = subprocess.run([sys.executable, bench] + extra + common, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        return json.loads(lines[0])

    two = run(["--gpus", "2", "--backend", "gloo", "--share-device"])
    a = run(["--gpus", "1", "--first-frame", "0"])
    b = run(["--gpus", "1", "--first-frame", "64"])
    assert two["n_gpus"] == 2 and two["checksum"]["frames"] == 128 and two["config"]["frames_per_gpu"] == 64
    assert two["checksum"]["sum_filtered_mm"] == a["checksum"]["sum_filtered_mm"] + b["checksum"]["sum_filtered_mm"]
    assert a["checksum"]["sum_filtered_mm"] != b["checksum"]["sum_filtered_mm"]
    assert two["value"] > 0 and two["scaling"] == "weak"
    assert two["ranks_seen"] == 2 and len({d["pid"] for d in two["devices"]}) == 2 and two["distinct_devices"] == 1     # one GPU, shared
    assert two["config"]["timed_call"].startswith("kde_jbf_process_batch") and two["boundary_vs_split"]["outputs_bit_identical"]


def _bench(extra, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_four_ranks_carry_the_1080p_and_chain_legs(torch_cuda):
    """what the one N > 1 run of the driver must deliver: at world > 1 the line still carries the 1080p
    window-19 pass and the batched chain, every rank having run them on its GPU, reduced like the headline.  Here 4 ranks
    share the box's one GPU over gloo, so the aggregate is about ONE GPU's rate -- the keys, the reductions and the checksum
    are what is checked."""
    j = _bench(["--gpus", "4", "--backend", "gloo", "--share-device", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-verify",
                "--wakeup-ms", "0", "--frames-per-gpu", "16"])
    assert j["n_gpus"] == 4 and j["ranks_seen"] == 4 and j["checksum"]["frames"] == 64 and j["rccl"]["wanted"] is False
    fhd, chain = j["roofline"]["fhd_w19"], j["also"]["vga_chain_batch64"]
    assert fhd["n_gpus"] == 4 and len(fhd["k1_ms_per_rank"]) == 4 and fhd["window"] == 19 and fhd["width"] == 1920
    assert fhd["process_mpix_s"] > 0 and 0 < fhd["frac"] < 1 and fhd["k1_avg_launch_ms"] == max(fhd["k1_ms_per_rank"])
    assert chain["n_gpus"] == 4 and len(chain["batched_ms_per_rank"]) == 4 and chain["frames_bit_identical_to_single_calls"] is True
    assert "cpu_baseline" not in j and "k1_w19_content_dependence" not in j["also"]          # N = 1 only


def test_rccl_failure_on_the_gpu_falls_back_in_process(torch_cuda):
    """the fallback with real filtering: RCCL "fails" on both ranks, the same two processes go on over gloo, the line is
    flagged and its checksum equals the healthy run's"""
    common = ["--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extra", "--no-verify", "--wakeup-ms", "0", "--frames-per-gpu", "8"]
    bad = _bench(["--gpus", "2", "--share-device", "--force-rccl-failure"] + common)
    good = _bench(["--gpus", "2", "--share-device", "--backend", "gloo"] + common)
    assert bad["replicas_only"] is True and "[REPLICAS ONLY]" in bad["config"]["sharding"] and bad["rccl"]["ok"] is False
    assert good["replicas_only"] is False
    assert bad["checksum"] == good["checksum"] and bad["value"] > 0


def test_one_rank_under_a_launcher_brings_rccl_up(torch_cuda):
    """N = 1 through the distributed path (what torch.distributed.run gives a rank): the RCCL group really is created on the
    box's GPU, the broadcast and the reductions go through it, and the line names the device by its PCI address"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0",
                        "--no-extra", "--no-verify", "--wakeup-ms", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    rc = j["rccl"]
    assert (rc["wanted"], rc["ok"], rc["backend_used"], rc["error"], rc["ranks_whose_block_differs_from_rank0"]) == (True, True, "nccl", None, 0)
    assert rc["probe"] is None and 0 <= rc["bring_up_s"] < rc["timeout_s"] == 60.0      # one rank: no probe; the bounded bring-up ran
    assert j["ranks_seen"] == 1 and j["devices"][0]["arch"].startswith("gfx950") and len(j["devices"][0]["pci_bus_id"]) >= 7
    assert j["replicas_only"] is False and "(nccl)" in j["config"]["sharding"]


def test_a_real_rccl_failure_is_caught_by_the_probes(torch_cuda):
    """the closest a one-GPU box gets to a broken node: two ranks asked to build an RCCL communicator on the SAME device, which
    RCCL refuses (or never completes).  The launcher's disposable probes -- real processes, real new_group("nccl") on the GPU --
    fail or are killed at the deadline; the two ranks then never touch RCCL, filter their shards over gloo, and the line is
    complete, flagged, and its checksum equals the healthy gloo run's"""
    common = ["--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extra", "--no-verify", "--wakeup-ms", "0", "--frames-per-gpu", "8"]
    t0 = time.time()
    bad = _bench(["--gpus", "2", "--share-device", "--backend", "nccl", "--rccl-timeout", "45"] + common)
    assert time.time() - t0 < 240
    good = _bench(["--gpus", "2", "--share-device", "--backend", "gloo"] + common)
    assert bad["rccl"]["probe"]["ok"] is False and bad["rccl"]["probe"]["by"] == "launcher" and bad["rccl"]["bring_up_s"] is None
    assert bad["replicas_only"] is True and "[REPLICAS ONLY]" in bad["config"]["sharding"] and "RCCL probe" in bad["rccl"]["error"]
    assert bad["checksum"] == good["checksum"] and bad["value"] > 0 and bad["ranks_seen"] == 2


@pytest.mark.parametrize("threads", [2, 8])
def test_cpp_host_threads_share_one_device(torch_cuda, threads):
    """kde_hip.h's threading contract, exercised: G host threads on device 0, each with its own stream, kde_jbf handle and shard
    buffers, all starting their timed steps together (the code path eight GPUs take).  The per-frame hashes must equal the
    one-thread runs of the same frames (one block, two half blocks): bit for bit, whatever the interleaving of the launches"""
    exe = os.path.join(ROOT, "examples", "shard_replay")
    r = subprocess.run([exe, "--share-device", str(threads), "--frames", "40", "--width", "320", "--height", "240", "--steps", "6", "--warmup", "2",
                        "--wakeup-ms", "20", "--verify"], capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["host_threads"] == threads and line["devices"] == 1 and len(line["per_device"]) == threads
    assert len({d["pci_bus_id"] for d in line["per_device"]}) == 1
    assert line["verified"] is True and line["tables_match_rank0"] is True and line["mpixels_per_s"] > 0
    assert line["params_broadcast"].startswith("replicas only (--share-device")


def test_cpp_host_bounds_ncclCommInitAll(torch_cuda):
    """a communicator bring-up that never returns (stand-in: the helper thread sleeps) costs --rccl-timeout seconds, not the run"""
    exe = os.path.join(ROOT, "examples", "shard_replay")
    t0 = time.time()
    r = subprocess.run([exe, "--frames", "6", "--width", "160", "--height", "120", "--steps", "2", "--verify", "--rccl-timeout", "3"],
                       capture_output=True, text=True, timeout=240, env=dict(os.environ, KDE_SHARD_REPLAY_TEST_HANG="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "not finished after 3 s, abandoned" in line["params_broadcast"] and line["verified"] is True
    assert time.time() - t0 < 120


def test_under_a_launcher_every_rank_probes_real_rccl_for_itself(torch_cuda):
    """the driver's N > 1 command form (torch.distributed.run starts the ranks, no parent of ours) with a REAL RCCL failure: two
    ranks on the box's one GPU.  Every rank starts the probe of its own rank before its first HIP call; the probes fail (or are
    killed at the deadline), the verdicts are exchanged over gloo, and the same two processes filter their shards over gloo"""
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--backend", "nccl",
                        "--rccl-timeout", "45", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extra", "--no-verify", "--wakeup-ms", "0",
                        "--frames-per-gpu", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert time.time() - t0 < 300
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert j["rccl"]["probe"]["ok"] is False and j["rccl"]["probe"]["by"] == "every rank, for itself" and j["rccl"]["bring_up_s"] is None
    assert j["replicas_only"] is True and "[REPLICAS ONLY]" in j["config"]["sharding"] and j["ranks_seen"] == 2 and j["value"] > 0
