#!/usr/bin/env python3
"""The config-5 chain on a BATCH of independent frames (north_star's unit): projectiveToReal -> JointBilateralFilter ->
projectiveToReal -> RegionGrowingBilateralFilter with every stage taking the whole batch per launch
(kde_rgbf_process_batch), against the same frames pushed through the single-frame calls one after the other.
    python tools/bench_chain_batch.py [--frames 64] [--width 640 --height 480] [--spdsr]"""
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


def run(torch, F, synth, W, H, n, iters, distinct=8, spdsr=False):
    bgr, depth = synth.make_batch(500, min(distinct, n), W, H)
    reps = -(-n // bgr.shape[0])
    bgr, depth = np.tile(bgr, (reps, 1, 1, 1))[:n], np.tile(depth, (reps, 1, 1))[:n]
    K = synth.intrinsics(W, H)
    color, d = torch.from_numpy(bgr).cuda(), torch.from_numpy(depth).cuda()
    conv = F.DimensionConvertor(); conv.setCameraParameters(K, W, H)
    jbf = F.JointBilateralFilter(W, H, max_batch=n)
    cls = F.SPDepthSuperResolution if spdsr else F.RegionGrowingBilateralFilter
    rgb = cls(W, H, max_batch=n); rgb.SetParametor(15, 20, K)
    rg1 = cls(W, H); rg1.SetParametor(15, 20, K)
    jbf1 = F.JointBilateralFilter(W, H)
    pts = torch.empty((n, H, W, 3), dtype=torch.float32, device="cuda")
    filt = torch.empty((n, H, W), dtype=torch.float32, device="cuda")

    def chain_batch():
        conv.projectiveToReal(d, pts)
        jbf.process_batch(d, color, filt)
        conv.projectiveToReal(filt, pts)
        rgb.process_batch(filt, pts, color)

    f1 = jbf1.getFiltered_Device()
    p1 = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")

    def chain_single_frames():
        for f in range(n):
            conv.projectiveToReal(d[f], p1)
            jbf1.Process(d[f], color[f])
            conv.projectiveToReal(f1, p1)
            rg1.Process(f1, p1, color[f])

    px = n * W * H
    res = {"frames": n, "width": W, "height": H, "pipeline": "SPDepthSuperResolution" if spdsr else "RegionGrowingBilateralFilter"}
    res["batched_ms"] = timed(torch, chain_batch, iters)
    res["batched_ms_per_frame"] = res["batched_ms"] / n
    res["batched_mpix_s"] = px / res["batched_ms"] / 1e3
    res["single_frame_calls_ms"] = timed(torch, chain_single_frames, max(2, iters // 4))
    res["single_frame_calls_ms_per_frame"] = res["single_frame_calls_ms"] / n
    res["single_frame_calls_mpix_s"] = px / res["single_frame_calls_ms"] / 1e3
    conv.projectiveToReal(filt, pts)
    res["pipeline_only_batched_ms_per_frame"] = timed(torch, lambda: rgb.process_batch(filt, pts, color), iters) / n
    # frame 0 and the last frame of the batch equal their single-frame calls to the bit
    chain_batch()
    got = rgb.getRefinedDepth_Device().clone()
    same = True
    for f in (0, n - 1):
        conv.projectiveToReal(d[f], p1)
        jbf1.Process(d[f], color[f])
        conv.projectiveToReal(f1, p1)
        rg1.Process(f1, p1, color[f])
        a, b = got[f].cpu().numpy(), rg1.getRefinedDepth_Device().cpu().numpy()
        same = same and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    res["frames_bit_identical_to_single_calls"] = bool(same)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--spdsr", action="store_true")
    ap.add_argument("--wakeup-ms", type=float, default=150.0)
    a = ap.parse_args()
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from wake import wake
    from kinectdepthmapenhancement_amd import filters as F, synth
    wake(torch, a.wakeup_ms)
    print(json.dumps(run(torch, F, synth, a.width, a.height, a.frames, a.iters, spdsr=a.spdsr)))


if __name__ == "__main__":
    main()
