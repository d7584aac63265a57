"""Disposable RCCL bring-up probe (SURVEY.md §8e "Fallback").

    RANK=r WORLD_SIZE=N LOCAL_RANK=r MASTER_ADDR=127.0.0.1 MASTER_PORT=p python kinectdepthmapenhancement_amd/rccl_probe.py

One short-lived process per rank: gloo rendezvous -> `new_group("nccl")` on the rank's own device -> one all-reduce ->
exit 0.  A communicator that cannot be built over xGMI usually does not raise, it HANGS (one rank never joins the
bootstrap ring), and a process that hangs inside RCCL cannot be recovered in place; so the bring-up is tried first in
processes that may simply be killed (exact PIDs) at a deadline.  The ranks that do the real work start RCCL themselves
only after every probe came back clean, and otherwise never touch it: they go on over gloo, flagged [REPLICAS ONLY] with
the probe's reason.  Nothing is ever re-executed in a process that has initialised the GPU.

`run_probes` is the launcher side: `bench.py`'s parent (which never touches a GPU) starts all N probes itself; under
`torch.distributed.run` every rank starts the probe of its own rank before its first HIP call and the outcomes are
exchanged over gloo (sharding.ShardComm).

KDE_RCCL_PROBE_TEST = ok | fail | hang | hang:<rank> | fail:<rank> replaces the GPU part with a stand-in (CPU tests of
the deadline and of the flagged line); it is read by the probe process only.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import tempfile
import time
from typing import Dict, List, Optional, Sequence

DEFAULT_DEADLINE_S = 60.0


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _stand_in(mode: str, rank: int) -> Optional[int]:
    """the test stand-in's exit code for this rank, None = hang"""
    kind, _, who = mode.partition(":")
    mine = who == "" or int(who) == rank
    if kind == "hang" and mine:
        return None
    if kind == "fail" and mine:
        return 3
    return 0


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    mode = os.environ.get("KDE_RCCL_PROBE_TEST", "")
    if mode:
        rc = _stand_in(mode, rank)
        if rc is None:
            while True:                 # the launcher's deadline must end us
                time.sleep(1.0)
        if rc:
            print(f"rccl_probe: rank {rank}: stand-in failure (KDE_RCCL_PROBE_TEST={mode})", file=sys.stderr)
        return rc
    import datetime

    import torch
    import torch.distributed as dist
    t0 = time.time()
    try:
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=DEFAULT_DEADLINE_S))
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
        g = dist.new_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=DEFAULT_DEADLINE_S))
        t = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(t, group=g)         # the first collective builds the communicator over xGMI
        torch.cuda.synchronize(dev)
        if int(t.item()) != world:
            raise RuntimeError(f"all-reduce of 1 over {world} ranks returned {t.item()}")
    except Exception as e:                  # noqa: BLE001 -- the LAST line of the log is what the launcher quotes
        msg = " ".join(str(e).split())
        k = msg.find("Duplicate GPU")       # RCCL's own reason, when it gives one, instead of the wrapper text around it
        print(f"rccl_probe: rank {rank} FAILED: {type(e).__name__}: {(msg[k:] if k >= 0 else msg)[:300]}", file=sys.stderr)
        sys.stderr.flush()
        os._exit(4)                         # no communicator teardown: the process is disposable
    print(f"rccl_probe: rank {rank}/{world} ok in {time.time() - t0:.1f} s", file=sys.stderr)
    sys.stderr.flush()
    os._exit(0)


def probe_env(rank: int, world: int, local_rank: int, port: int) -> Dict[str, str]:
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE",
              "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "TORCH_NCCL_ASYNC_ERROR_HANDLING"):
        env.pop(k, None)                # a probe under torch.distributed.run is a plain process with a store of its own
    return env


def start_probe(rank: int, world: int, local_rank: int, port: int) -> subprocess.Popen:
    # run as a script (not -m): the probe needs nothing of the package; its stderr goes to a file, never a pipe it could fill
    log = tempfile.TemporaryFile()
    q = subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=probe_env(rank, world, local_rank, port),
                         stdout=log, stderr=log)
    q.kde_log = log
    return q


def wait_probes(procs: Sequence[subprocess.Popen], ranks: Sequence[int], deadline_s: float, peer_failed=None) -> Dict:
    """wait for the probes; those still running at the deadline -- or once another probe has failed, since they would only
    wait for it -- are killed (exact PIDs).  `peer_failed()` (optional) reports a failure seen elsewhere (the other ranks'
    probes under torch.distributed.run).  -> {"ok", "reason", "seconds"}"""
    t0 = time.time()
    cut = None
    while any(q.poll() is None for q in procs):
        if time.time() - t0 >= deadline_s:
            cut = f"still in the bring-up at the {deadline_s:g} s deadline"
            break
        if any(q.poll() not in (None, 0) for q in procs) or (peer_failed is not None and peer_failed()):
            cut = "ended because another rank's probe failed"
            break
        time.sleep(0.05)
    bad: List[str] = []
    for r, q in zip(ranks, procs):
        rc = q.poll()
        if rc is None:
            q.kill()
            q.wait()
            bad.append(f"rank {r} {cut}, killed")
        elif rc != 0:
            q.kde_log.seek(0)
            tail = q.kde_log.read().decode(errors="replace").strip().splitlines()[-1:]
            bad.append(f"rank {r} exit code {rc}" + (f": {tail[0][:200]}" if tail else ""))
    for q in procs:
        q.kde_log.seek(0)
        sys.stderr.write(q.kde_log.read().decode(errors="replace"))
        q.kde_log.close()
    bad.sort(key=lambda b: "another rank" in b)         # the ranks that failed by themselves first
    return {"ok": not bad, "reason": "; ".join(bad)[:600] if bad else None, "seconds": round(time.time() - t0, 2)}


def run_probes(world: int, deadline_s: float = DEFAULT_DEADLINE_S, share_device: bool = False) -> Dict:
    """launcher side: all N probes from one parent that never touches a GPU"""
    port = free_port()
    procs = [start_probe(r, world, 0 if share_device else r, port) for r in range(world)]
    return wait_probes(procs, list(range(world)), deadline_s)


if __name__ == "__main__":
    sys.exit(main())
