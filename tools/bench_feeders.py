#!/usr/bin/env python3
"""The HBM-bound feeders on working sets far beyond the 256 MB Infinity Cache: Buffer2D::updateData
over F distinct 1080p frames fused into one read-modify-write (4 F + 16 B/pixel; F = 128 -> 1.09 GB), projectiveToReal on
the same frames (16 B/pixel -> 4.2 GB) and the float4-copy ceiling on the same bytes.  Every frame is a distinct
allocation region with distinct content, so nothing is served from a cache by address or by construction.
    python tools/bench_feeders.py [--frames 128]      (tools/profile_round.sh pmc_feeders runs it under rocprofv3 --pmc)"""
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
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--wakeup-ms", type=float, default=150.0)
    a = ap.parse_args()
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from wake import wake
    from kinectdepthmapenhancement_amd import filters as F, synth
    W, H, n = a.width, a.height, a.frames
    px = W * H
    _, depth = synth.make_frame(91, W, H)
    base = torch.from_numpy(depth).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    # F distinct frames: the scene plus per-frame sensor noise (most samples pass the 1 % gate of updateWaitedDepth)
    seq = base[None] * (1.0 + 0.002 * torch.randn((n, H, W), device="cuda", generator=g))
    seq = torch.where(base[None] > 50, seq, torch.zeros_like(seq)).contiguous()
    K = synth.intrinsics(W, H)
    conv = F.DimensionConvertor(); conv.setCameraParameters(K, W, H)
    buf = F.Buffer2D(W, H)
    wake(torch, a.wakeup_ms)
    res = {"frames": n, "width": W, "height": H}

    def entry(ms, nbytes):
        return {"ms": ms, "algorithmic_MB": nbytes / 1e6, "GBs": nbytes / ms / 1e6, "hbm_frac": nbytes / ms / 1e6 / 8000}

    buf.insertData(seq[0])
    res["buffer2d_update_sequence"] = entry(timed(torch, lambda: buf.updateData(seq), a.iters), (4.0 * n + 16.0) * px)
    pts = torch.empty((n, H, W, 3), dtype=torch.float32, device="cuda")
    res["projectiveToReal_depth"] = entry(timed(torch, lambda: conv.projectiveToReal(seq, pts), a.iters), 16.0 * n * px)
    import importlib.util
    spec = importlib.util.spec_from_file_location("kde_hooks", os.path.join(os.path.dirname(os.path.abspath(__file__)), "hooks", "hooks.py"))
    hooks = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hooks)
    st = torch.cuda.current_stream().cuda_stream
    dst = torch.empty_like(seq)
    res["float4_copy_of_the_sequence"] = entry(timed(torch, lambda: hooks.hbm_copy(seq, dst, st), a.iters), 8.0 * n * px)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
