"""The N > 1 path with real filtering: two ranks (own processes, gloo group, both on cuda:0 -- the box has one GPU)
each filter their block of the batch through the HIP library; the gathered result must be bit-identical to one
process filtering the whole batch.  And the C++ multi-device host (examples/shard_replay: one thread per device,
ncclBroadcast of the parameter block) with its own partition-independence check."""
import json
import os
import socket
import subprocess
import sys

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
