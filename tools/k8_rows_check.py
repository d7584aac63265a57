"""CRCs of labels / cluster records / float centres of DepthAdaptiveSuperpixel::Segmentation on the measurement build (tools/hooks/
libkde_hip_ab.so): with KDE_K8_ROWS=1 analyzeClusters runs in its row-coalesced form (tests/test_gpu_dasp_ers.py compares with the
product library; tools/bench_spdsr.py times it).   python tools/k8_rows_check.py <repo root>"""
import os, sys, zlib, numpy as np, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "hooks"))
import ab; ab.use_ab_library()
from kinectdepthmapenhancement_amd import filters as F, synth
from oracle import oracle as O
out = {}
for (w, h, g, it) in ((1920, 1080, (15, 20), 5), (640, 480, (15, 20), 5), (203, 131, (5, 7), 3), (640, 480, (10, 8), 2), (320, 240, (6, 8), 4)):
    bgr, depth = synth.make_frame(33, w, h); K = synth.intrinsics(w, h)
    pts = O.p2r_depth(depth, K).view(np.float32).reshape(h, w, 3)
    d = F.DepthAdaptiveSuperpixel(w, h); d.SetParametor(g[0], g[1], K)
    d.Segmentation(torch.from_numpy(bgr).cuda(), torch.from_numpy(np.ascontiguousarray(pts)).cuda(), 100.0, 20.0, 200.0, it)
    torch.cuda.synchronize()
    lab = d.getLabelDevice().cpu().numpy(); mean = d.getMeanDataDevice().cpu().numpy(); cen = d.getCentersDevice().cpu().numpy()
    out[f"{w}x{h}_{g}_{it}"] = (zlib.crc32(lab.tobytes()), zlib.crc32(mean.tobytes()), zlib.crc32(cen.tobytes()))
print(out)
