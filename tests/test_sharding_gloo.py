"""The N>1 path on CPU: world_size-2 gloo process group, contiguous frame blocks, one parameter
broadcast, no data-path collective.  This file covers the partition / broadcast / reduce LOGIC without a
GPU (the per-rank filter is a stand-in); tests/test_gpu_sharding.py runs the same two-rank layout with every rank really
filtering its block through the HIP library and requires a result bit-identical to one process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kinectdepthmapenhancement_amd import sharding
from kinectdepthmapenhancement_amd._native import JbfParams


def test_partition_contiguous_blocks():
    assert sharding.partition(512, 8) == [(i * 64, 64) for i in range(8)]
    assert sharding.partition(5, 4) == [(0, 2), (2, 2), (4, 1), (5, 0)]
    assert sharding.partition(0, 3) == [(0, 0)] * 3
    for n, g in [(1, 1), (7, 2), (100, 8), (64, 6)]:
        parts = sharding.partition(n, g)
        assert sum(c for _, c in parts) == n
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(g - 1))
    with pytest.raises(ValueError):
        sharding.partition(4, 0)


def test_pack_unpack_roundtrip():
    p = JbfParams(11, 3.0, 7.65, 20.0, 1, 5, 30.0, 30.0)
    tab = np.arange(121, dtype=np.float32).reshape(11, 11)
    K = np.arange(9, dtype=np.float64)
    q, rows, cols, K2, tab2 = sharding.unpack_params(sharding.pack_params(p, 15, 20, K, tab))
    assert (q.window_size, q.presmooth, q.presmooth_kernel_size, rows, cols) == (11, 1, 5, 15, 20)
    assert abs(q.color_sigma - 7.65) < 1e-6 and np.array_equal(K2.reshape(9), K) and np.array_equal(tab2, tab)
    assert np.array_equal(sharding.broadcast_params(np.ones(4)), np.ones(4))      # no process group: identity


def _worker(rank, world, port, n_frames, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 owns the parameters; the others start from garbage and must end up identical
        if rank == 0:
            p = JbfParams(11, 3.0, 7.65, 20.0, 1, 5, 30.0, 30.0)
            blk = sharding.pack_params(p, 15, 20, np.eye(3).reshape(9), np.full((11, 11), 0.5, np.float32))
        else:
            blk = np.full(sharding.BLOCK_LEN, -1.0)
        got = sharding.broadcast_params(blk)
        p, rows, cols, K, tab = sharding.unpack_params(got)
        start, count = sharding.partition(n_frames, world)[rank]
        # stand-in for the per-rank filter: every rank "processes" its own frames only
        done = np.zeros(n_frames)
        done[start:start + count] = 1
        total = sharding.allreduce_sum(done)
        tmax = sharding.allreduce_max(float(rank + 1))
        ret[rank] = (p.window_size, rows, cols, float(tab.sum()), start, count, total.tolist(), tmax)
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo_broadcast_and_disjoint_shards():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, 7, ret), nprocs=2, join=True)
    assert ret[0][:4] == ret[1][:4] == (11, 15, 20, 60.5)
    assert (ret[0][4], ret[0][5]) == (0, 4) and (ret[1][4], ret[1][5]) == (4, 3)
    assert ret[0][6] == [1.0] * 7          # every frame processed exactly once across the ranks
    assert ret[0][7] == ret[1][7] == 2.0   # max over ranks


def _comm_worker(rank, world, port, force_fail, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    comm = sharding.ShardComm("nccl", local_rank=rank, use_gpu=False, force_rccl_failure=force_fail)
    try:
        blk = np.full(sharding.BLOCK_LEN, float(rank + 1))          # every rank forms "its own" block
        got = comm.broadcast_params(blk)                            # rank 0's arrives everywhere
        differs = int(comm.allreduce_sum([0.0 if np.array_equal(blk, got) else 1.0])[0])
        comm.barrier()
        ret[rank] = (comm.active, comm.replicas_only, comm.backend_used, comm.rccl_error, float(got[0]), differs,
                     comm.allreduce_max(float(rank)), comm.gather_objects({"rank": rank}))
    finally:
        comm.close()


@pytest.mark.parametrize("force_fail", [False, True])
def test_shardcomm_gloo_rendezvous_and_in_process_fallback(force_fail):
    """sharding.ShardComm on two CPU ranks: the gloo group is the rendezvous; without a GPU RCCL is not attempted (dry run), with
    a forced RCCL failure every rank agrees on the fallback -- same processes, gloo for the broadcast / reductions, `replicas_only`
    and the first rank's exception text on EVERY rank (SURVEY 8e "Fallback")."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_comm_worker, args=(2, port, force_fail, ret), nprocs=2, join=True)
    for r in (0, 1):
        active, replicas, backend, err, first, differs, tmax, objs = ret[r]
        assert active and first == 1.0 and differs == 1 and tmax == 1.0            # rank 0's block; exactly one rank's own differs
        assert objs == [{"rank": 0}, {"rank": 1}]
        if force_fail:
            assert replicas and backend.startswith("gloo (RCCL unavailable)") and err.startswith("rank 0: RuntimeError") and "+1 more" in err
        else:
            assert not replicas and err is None and backend.startswith("gloo (dry run")
