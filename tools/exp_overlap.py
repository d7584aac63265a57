#!/usr/bin/env python3
"""Experiment: a batch cut into P sub-batches, each K0 -> K1 on its own stream (fork / join on the caller's stream)."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wake import wake
from kinectdepthmapenhancement_amd import filters as F, synth
W, H, N = 640, 480, 64
bgr, depth = synth.make_batch(0, 64, W, H)
color, d = torch.from_numpy(bgr).cuda(), torch.from_numpy(depth).cuda()
p = F.JointBilateralFilter.default_params()
p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma = 11, 3.0, 7.65, 20.0
out = torch.empty((N, H, W), dtype=torch.float32, device="cuda")
main = torch.cuda.current_stream()
streams = [torch.cuda.Stream() for _ in range(16)]
objs = {P: [F.JointBilateralFilter(W, H, params=p, max_batch=N // P) for _ in range(P)] for P in (1, 2, 4, 8, 16)}

def split(P):
    n = N // P
    def step():
        fork = torch.cuda.Event(); fork.record(main)
        for i in range(P):
            s = streams[i]
            s.wait_event(fork)
            with torch.cuda.stream(s):
                objs[P][i].process_batch(d[i * n:(i + 1) * n], color[i * n:(i + 1) * n], out[i * n:(i + 1) * n])
            e = torch.cuda.Event(); e.record(s)
            main.wait_event(e)
    return step

def serial():
    objs[1][0].process_batch(d, color, out)

def timed(step, iters=20):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

wake(torch, 150.0)
res = {}
serial(); torch.cuda.synchronize(); ref = out.clone()
for rnd in range(3):
    res.setdefault("serial", []).append(timed(serial))
    for P in (1, 2, 4, 8, 16):
        out.zero_()
        res.setdefault(f"split_{P}", []).append(timed(split(P)))
        assert torch.equal(out, ref)
print(json.dumps({k: [round(x, 4) for x in v] for k, v in res.items()}))
