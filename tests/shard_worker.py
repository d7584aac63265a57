"""One rank of the 2-rank sharding test (tests/test_gpu_sharding.py): started as its own process BEFORE it touches
the GPU, joins a gloo process group, receives the parameter block from rank 0, filters ITS block of the global batch
on cuda:0 through the HIP library, and sends the result to rank 0, which writes all frames to an .npy file.

    python tests/shard_worker.py <rank> <world> <port> <n_frames> <first_seed> <out.npy>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, n_frames, first_seed = (int(v) for v in sys.argv[1:6])
    out_path = sys.argv[6]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from kinectdepthmapenhancement_amd import filters, sharding, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        W, H = 160, 120
        if rank == 0:
            p = filters.JointBilateralFilter.default_params()
            p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = 11, 3.0, 7.65, 20.0
            probe = filters.JointBilateralFilter(8, 8, p)
            blk = sharding.pack_params(p, table=probe.spatial_table())
        else:
            blk = np.full(sharding.BLOCK_LEN, -1.0)                   # garbage until the broadcast
        p, _, _, _, table0 = sharding.unpack_params(sharding.broadcast_params(blk))
        first, count = sharding.partition(n_frames, world)[rank]
        mine = np.zeros((0, H, W), np.float32)
        if count:
            bgr, depth = synth.make_batch(first_seed + first, count, W, H)
            jbf = filters.JointBilateralFilter(W, H, p, max_batch=count)
            assert np.array_equal(jbf.spatial_table(), table0), "this rank's table differs from rank 0's"
            out = jbf.process_batch(torch.from_numpy(depth).cuda(), torch.from_numpy(bgr).cuda())
            mine = out.cpu().numpy()
        # rank 0 collects the shards (a report-side gather, not part of the data path)
        parts = [None] * world
        dist.gather_object(mine, parts if rank == 0 else None, dst=0)
        if rank == 0:
            np.save(out_path, np.concatenate(parts, 0))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
