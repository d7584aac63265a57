"""Frame-batch data parallelism over the GPUs of one node.

Frames are independent units: a batch of N frames is cut into contiguous blocks of ceil(N/G)
frames, one block per rank (one process per GPU), with NO data-path collective.  The only exchange
is one broadcast of the parameter block from rank 0 at start-up (RCCL over xGMI when the backend
is "nccl", gloo in the CPU tests) so every rank provably filters with identical parameters and
spatial table, plus an optional all-reduce of a few scalars for the report.
"""
from __future__ import annotations

import datetime
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from ._native import JbfParams

MAX_WINDOW = 31
_HEAD = 8 + 2 + 9          # jbf params, (rows, cols), K[9]
BLOCK_LEN = _HEAD + MAX_WINDOW * MAX_WINDOW


def partition(n_frames: int, world_size: int) -> List[Tuple[int, int]]:
    """[(first_frame, count)] per rank: contiguous blocks of ceil(N/G), trailing ranks may be short/empty."""
    if n_frames < 0 or world_size < 1:
        raise ValueError("partition: n_frames >= 0 and world_size >= 1 required")
    per = -(-n_frames // world_size) if n_frames else 0
    out = []
    for r in range(world_size):
        start = min(r * per, n_frames)
        out.append((start, max(0, min(per, n_frames - start))))
    return out


def pack_params(p: JbfParams, rows: int = 0, cols: int = 0, K: Optional[Sequence[float]] = None,
                table: Optional[np.ndarray] = None) -> np.ndarray:
    blk = np.zeros(BLOCK_LEN, np.float64)
    blk[0:8] = [p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma, p.presmooth,
                p.presmooth_kernel_size, p.presmooth_sigma_color, p.presmooth_sigma_spatial]
    blk[8:10] = [rows, cols]
    if K is not None:
        blk[10:19] = np.asarray(K, np.float64).reshape(9)
    if table is not None:
        t = np.asarray(table, np.float32).reshape(-1)
        blk[_HEAD:_HEAD + t.size] = t
    return blk


def unpack_params(blk: np.ndarray):
    p = JbfParams()
    p.window_size = int(blk[0])
    p.spatial_sigma, p.color_sigma, p.depth_sigma = float(blk[1]), float(blk[2]), float(blk[3])
    p.presmooth, p.presmooth_kernel_size = int(blk[4]), int(blk[5])
    p.presmooth_sigma_color, p.presmooth_sigma_spatial = float(blk[6]), float(blk[7])
    rows, cols = int(blk[8]), int(blk[9])
    K = blk[10:19].reshape(3, 3).copy()
    w = p.window_size
    table = blk[_HEAD:_HEAD + w * w].astype(np.float32).reshape(w, w)
    return p, rows, cols, K, table


def broadcast_params(block: np.ndarray, device: Optional[torch.device] = None, src: int = 0,
                     group=None) -> np.ndarray:
    """One broadcast of the parameter block from `src`; returns the received block on every rank.

    With an uninitialised process group (single process) it is the identity."""
    if not (dist.is_available() and dist.is_initialized()):
        return block.copy()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" \
            else torch.device("cpu")
    t = torch.from_numpy(np.array(block, np.float64, copy=True)).to(device)     # a copy: the caller's block must stay its own
    dist.broadcast(t, src=src, group=group)                                      # (on the CPU .to() would alias it)
    return t.cpu().numpy()


def allreduce_sum(values: Sequence[float], device: Optional[torch.device] = None, group=None) -> np.ndarray:
    v = np.asarray(values, np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return v.copy()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" \
            else torch.device("cpu")
    t = torch.from_numpy(v.copy()).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def allreduce_max(value: float, device: Optional[torch.device] = None, group=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" \
            else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


class ShardComm:
    """The process groups of a sharded run (one process per GPU), with SURVEY 8(e)'s fallback built in.

    A gloo group over 127.0.0.1 is always created first: it is the rendezvous, it carries the agreement on whether RCCL is
    usable, and it is what the run falls back to.  With backend "nccl" every rank then tries to bring up an RCCL group on
    its own device (`new_group(backend="nccl", device_id=...)` + one all-reduce as a smoke test) INSIDE THE SAME PROCESS;
    the ranks exchange their outcome over gloo, and RCCL is used for the parameter broadcast and the report's reductions
    only if it came up on EVERY rank (the barriers around timed regions stay host-side, see barrier()).  Otherwise nothing is re-executed or restarted (a process that has touched
    the GPU must not be replaced): every rank computes the parameter block itself ("replicas only"), the blocks are
    compared over gloo, `rccl_error` holds the first exception text and the caller flags the run.

    Without RANK in the environment (plain single process) every method is the identity."""

    def __init__(self, backend: str = "nccl", local_rank: int = 0, use_gpu: bool = True, force_rccl_failure: bool = False,
                 timeout_s: float = 600.0):
        self.active = "RANK" in os.environ
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if self.active else 1
        self.rank = int(os.environ.get("RANK", "0")) if self.active else 0
        self.local_rank = local_rank
        self.group = None               # the group the broadcast / reductions / barriers use (None = the gloo world group)
        self.backend_used = "single process"
        self.rccl_error: Optional[str] = None
        self.rccl_wanted = backend == "nccl"
        if not self.active:
            return
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=timeout_s))
        self.backend_used = "gloo"
        if not self.rccl_wanted:
            return
        if not use_gpu and not force_rccl_failure:
            self.backend_used = "gloo (dry run: RCCL not attempted)"
            return
        err, g = None, None
        try:
            if force_rccl_failure:
                raise RuntimeError("RCCL initialisation failure forced by --force-rccl-failure (test of the fallback)")
            dev = torch.device("cuda", local_rank)
            g = dist.new_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=timeout_s))
            t = torch.ones(1, dtype=torch.float64, device=dev)
            dist.all_reduce(t, group=g)                 # the first collective builds the communicator over xGMI
            torch.cuda.synchronize(dev)
            if int(t.item()) != self.world:
                raise RuntimeError(f"RCCL all-reduce of 1 over {self.world} ranks returned {t.item()}")
        except Exception as e:          # noqa: BLE001 -- whatever RCCL / HIP raised: the fallback takes it from here
            err = f"{type(e).__name__}: {e}"[:400]
        outcome = [None] * self.world
        dist.all_gather_object(outcome, err)            # over gloo: RCCL is used only if it came up on every rank
        failed = [(r, e) for r, e in enumerate(outcome) if e]
        if failed:
            self.rccl_error = f"rank {failed[0][0]}: {failed[0][1]}" + (f" (+{len(failed) - 1} more ranks)" if len(failed) > 1 else "")
            self.backend_used = "gloo (RCCL unavailable)"
        else:
            self.group, self.backend_used = g, "nccl"

    @property
    def replicas_only(self) -> bool:
        return self.rccl_error is not None

    def _device(self):
        return torch.device("cuda", self.local_rank) if self.group is not None else torch.device("cpu")

    def barrier(self) -> None:
        """host-side rendezvous of the ranks (always over the gloo group).  The barriers that bracket a timed region must not
        touch the GPU: an RCCL barrier is a kernel plus a host wait of a few milliseconds, long enough for an MI355X to drop
        its clock -- measured: the first timed steps right behind it ran 22 % slow and the headline lost 10 %
        (tools/exp_dist_overhead.sh).  RCCL carries the parameter broadcast and the reductions, where latency is irrelevant."""
        if self.active:
            dist.barrier()

    def broadcast_params(self, block: np.ndarray, src: int = 0) -> np.ndarray:
        return broadcast_params(block, self._device(), src, self.group) if self.active else block.copy()

    def allreduce_sum(self, values: Sequence[float]) -> np.ndarray:
        return allreduce_sum(values, self._device(), self.group) if self.active else np.asarray(values, np.float64).copy()

    def allreduce_max(self, value: float) -> float:
        return allreduce_max(value, self._device(), self.group) if self.active else float(value)

    def gather_objects(self, obj) -> list:
        """every rank's `obj`, on every rank (report data only: goes over the gloo group)"""
        if not self.active:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def close(self) -> None:
        if self.active and dist.is_initialized():
            dist.destroy_process_group()
