#!/usr/bin/env python3
"""BASELINE config 5: the full chain DimensionConvertor -> JointBilateralFilter -> RegionGrowingBilateralFilter
(fed with the JBF output and its back-projection) on one frame, timed per stage with HIP events; also the
streaming feeders (projectiveToReal, Buffer2D::updateData) against their HBM roofline."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(torch, fn, iters):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--stream-frames", type=int, default=32)
    a = ap.parse_args()
    import torch
    from kinectdepthmapenhancement_amd import filters as F, synth
    W, H = a.width, a.height
    bgr, depth = synth.make_frame(77, W, H)
    K = synth.intrinsics(W, H)
    color, d = torch.from_numpy(bgr).cuda(), torch.from_numpy(depth).cuda()
    conv = F.DimensionConvertor(); conv.setCameraParameters(K, W, H)
    jbf = F.JointBilateralFilter(W, H)
    rg = F.RegionGrowingBilateralFilter(W, H); rg.SetParametor(15, 20, K)
    pts = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    res = {"width": W, "height": H, "pixels": W * H}
    px = W * H

    res["projectiveToReal_ms"] = timed(torch, lambda: conv.projectiveToReal(d, pts), a.iters)
    res["jbf_process_ms"] = timed(torch, lambda: jbf.Process(d, color), a.iters)
    filt = jbf.getFiltered_Device()
    conv.projectiveToReal(filt, pts)
    res["rgbf_process_ms"] = timed(torch, lambda: rg.Process(filt, pts, color), a.iters)

    def chain():
        conv.projectiveToReal(d, pts)          # K2 on the input (main.cpp:168)
        jbf.Process(d, color)                  # K0 + K1
        conv.projectiveToReal(filt, pts)       # K2 on the JBF output
        rg.Process(filt, pts, color)           # K5-K10
    res["chain_ms"] = timed(torch, chain, a.iters)
    res["chain_mpix_s"] = px / res["chain_ms"] / 1e3
    # the same chain captured once into a hipGraph and replayed (launch-bound at small frames)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            chain()
    torch.cuda.current_stream().wait_stream(side)
    res["chain_graph_ms"] = timed(torch, g.replay, a.iters)

    # streaming feeders on a batch (HBM-bound): algorithmic bytes / time against 8 TB/s
    n = a.stream_frames
    db = d[None].repeat(n, 1, 1).contiguous()
    pb = torch.empty((n, H, W, 3), dtype=torch.float32, device="cuda")
    ms = timed(torch, lambda: conv.projectiveToReal(db, pb), a.iters)
    res["p2r_batch"] = {"frames": n, "ms": ms, "GBs": 16.0 * n * px / ms / 1e6, "hbm_frac": 16.0 * n * px / ms / 1e6 / 8000}
    # K3: the point-cloud forms (24 B/px) and the interpolating form of projectiveToReal (16 B/px)
    qb = torch.empty_like(pb)
    ms = timed(torch, lambda: conv.realToProjective(pb, qb), a.iters)
    res["r2p_batch"] = {"frames": n, "ms": ms, "GBs": 24.0 * n * px / ms / 1e6, "hbm_frac": 24.0 * n * px / ms / 1e6 / 8000}
    ms = timed(torch, lambda: conv.projectiveToReal(qb, pb), a.iters)
    res["p2r_points_batch"] = {"frames": n, "ms": ms, "GBs": 24.0 * n * px / ms / 1e6, "hbm_frac": 24.0 * n * px / ms / 1e6 / 8000}
    ms = timed(torch, lambda: conv.projectiveToRealInterp(db, pb), a.iters)
    res["p2r_interp_batch"] = {"frames": n, "ms": ms, "GBs": 16.0 * n * px / ms / 1e6, "hbm_frac": 16.0 * n * px / ms / 1e6 / 8000}
    del qb
    buf = F.Buffer2D(W, H)
    if (W * H) % 2 == 0:
        ms = timed(torch, lambda: buf.updateData(db), a.iters)
        byts = (4.0 * n + 16.0) * px
        res["buffer2d_update_sequence"] = {"frames": n, "ms": ms, "GBs": byts / ms / 1e6, "hbm_frac": byts / ms / 1e6 / 8000}
    ms = timed(torch, lambda: buf.updateData(d), a.iters)
    res["buffer2d_update_single"] = {"ms": ms, "GBs": 20.0 * px / ms / 1e6, "hbm_frac": 20.0 * px / ms / 1e6 / 8000}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
