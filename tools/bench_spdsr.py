#!/usr/bin/env python3
"""SPDepthSuperResolution::Process (SPDepthSuperResolution.cpp:57-190: DASP x5 + SP, EdgeRefining, per-superpixel plane
fit, plane projection, 20 mrf sweeps) on one frame, timed with HIP events."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hooks"))
import ab as _ab                                    # noqa: E402
_ab.use_ab_library_if_switched()                    # a KDE_* A/B switch in the environment -> tools/hooks/libkde_hip_ab.so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--rows", type=int, default=15)
    ap.add_argument("--cols", type=int, default=20)
    ap.add_argument("--iters", type=int, default=20)
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
    pts = torch.empty((H, W, 3), dtype=torch.float32, device="cuda")
    conv.projectiveToReal(d, pts)
    sr = F.SPDepthSuperResolution(W, H)
    sr.SetParametor(a.rows, a.cols, K)
    wake(torch, a.wakeup_ms)
    sr.Process(d, pts, color)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        sr.Process(d, pts, color)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    # single calls with a host synchronisation after each (what a caller that needs the frame sees): wall clock
    import time
    import zlib
    t0 = time.perf_counter()
    for _ in range(a.iters):
        sr.Process(d, pts, color)
        torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters * 1e3
    print(json.dumps({"width": W, "height": H, "rows": a.rows, "cols": a.cols, "spdsr_process_ms": ms, "spdsr_process_wall_ms_synced": wall,
                      "mpix_s": W * H / ms / 1e3, "resident_mode": os.environ.get("KDE_SPDSR_RESIDENT", "product default"),
                      "crc_optimized": zlib.crc32(sr.getOptimizedPoints_Device().cpu().numpy().tobytes())}))


if __name__ == "__main__":
    main()
