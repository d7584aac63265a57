"""Frame-batch data parallelism over the GPUs of one node.

Frames are independent units: a batch of N frames is cut into contiguous blocks of ceil(N/G)
frames, one block per rank (one process per GPU), with NO data-path collective.  The only exchange
is one broadcast of the parameter block from rank 0 at start-up (RCCL over xGMI when the backend
is "nccl", gloo in the CPU tests) so every rank provably filters with identical parameters and
spatial table, plus an optional all-reduce of a few scalars for the report.
"""
from __future__ import annotations

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
    t = torch.from_numpy(np.ascontiguousarray(block, np.float64)).to(device)
    dist.broadcast(t, src=src, group=group)
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
