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
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hooks"))
import ab as _ab                                    # noqa: E402
_ab.use_ab_library_if_switched()                    # a KDE_* A/B switch in the environment -> tools/hooks/libkde_hip_ab.so


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
    ap.add_argument("--wakeup-ms", type=float, default=150.0, help="untimed load before anything is measured (tools/wake.py)")
    a = ap.parse_args()
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from wake import wake
    from kinectdepthmapenhancement_amd import filters as F, synth
    W, H = a.width, a.height
    bgr, depth = synth.make_frame(77, W, H)
    K = synth.intrinsics(W, H)
    color, d = torch.from_numpy(bgr).cuda(), torch.from_numpy(depth).cuda()
    conv = F.DimensionConvertor(); conv.setCameraParameters(K, W, H)
    jbf = F.JointBilateralFilter(W, H)
    rg = F.RegionGrowingBilateralFilter(W, H); rg.SetParametor(15, 20, K)
    pts = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    wake(torch, a.wakeup_ms)
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

    # streaming feeders on a batch (HBM-bound): algorithmic bytes / time against 8 TB/s.  Every working set below is
    # beyond the 256 MB Infinity Cache except the single-frame update, which is reported as cache-resident.
    n = a.stream_frames
    db = d[None].repeat(n, 1, 1).contiguous()
    pb = torch.empty((n, H, W, 3), dtype=torch.float32, device="cuda")

    def entry(ms, bytes_per_px, frames, note=None):
        byts = bytes_per_px * frames * px
        e = {"frames": frames, "ms": ms, "GBs": byts / ms / 1e6, "hbm_frac": byts / ms / 1e6 / 8000, "working_set_MB": byts / 1e6}
        if note:
            e["note"] = note
        return e

    res["p2r_batch"] = entry(timed(torch, lambda: conv.projectiveToReal(db, pb), a.iters), 16.0, n)
    # K3: the point-cloud forms (24 B/px) and the interpolating form of projectiveToReal (16 B/px)
    qb = torch.empty_like(pb)
    res["r2p_batch"] = entry(timed(torch, lambda: conv.realToProjective(pb, qb), a.iters), 24.0, n)
    res["p2r_points_batch"] = entry(timed(torch, lambda: conv.projectiveToReal(qb, pb), a.iters), 24.0, n)
    res["p2r_interp_batch"] = entry(timed(torch, lambda: conv.projectiveToRealInterp(db, pb), a.iters), 16.0, n)
    # the ceiling of this chip for a 1:1 read:write stream: a float4 copy of the same 1.6 GB (tools/hooks, not product code)
    import importlib.util
    spec = importlib.util.spec_from_file_location("kde_hooks", os.path.join(os.path.dirname(os.path.abspath(__file__)), "hooks", "hooks.py"))
    hooks = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hooks)
    st = torch.cuda.current_stream().cuda_stream
    ms = timed(torch, lambda: hooks.hbm_copy(pb, qb, st), a.iters)
    res["float4_copy_ceiling"] = {"ms": ms, "GBs": 2.0 * pb.numel() * 4 / ms / 1e6, "hbm_frac": 2.0 * pb.numel() * 4 / ms / 1e6 / 8000,
                                  "working_set_MB": 2.0 * pb.numel() * 4 / 1e6}
    del qb
    buf = F.Buffer2D(W, H)
    ms = timed(torch, lambda: buf.updateData(db), a.iters)
    e = entry(ms, 4.0 + 16.0 / n, n)
    e["algorithmic_bytes_per_px"] = 4.0 * n + 16.0
    res["buffer2d_update_sequence"] = e
    res["buffer2d_update_single"] = entry(timed(torch, lambda: buf.updateData(d), a.iters), 20.0, 1,
                                          "41 MB working set: resident in the 256 MB Infinity Cache, not an HBM figure")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
