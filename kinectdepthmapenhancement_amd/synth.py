"""Deterministic synthetic RGB-D frames for tests and benchmarks.

The reference's only intended input (input/depth.xml) is absent, so parity and throughput runs use
this generator: a gradient background plus 24 random axis-aligned rectangles (colour + sloped depth
plane, painter's order), +-4 levels of colour noise, depth noise following the model of the
commented-out generator in main.cpp:127-130 (half-width 0.45*2.85*(z/10)^2/10000 mm), an 8-px
"shadow" band of invalid depth left of every rectangle and 1 % invalid salt.

Randomness is counter-based (a PCG-style 32-bit output permutation of seed/stream/index), so a
frame is a pure function of (seed, width, height) and is vectorised in numpy.  No reference code.
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _pcg_hash(x: np.ndarray) -> np.ndarray:
    """PCG-RXS-M-XS 32-bit output function applied to one LCG step (uint64 carrier, 32-bit values)."""
    x = (x * np.uint64(747796405) + np.uint64(2891336453)) & _M32
    w = (((x >> ((x >> np.uint64(28)) + np.uint64(4))) ^ x) * np.uint64(277803737)) & _M32
    return ((w >> np.uint64(22)) ^ w) & _M32


def rand_u32(seed: int, stream: int, index) -> np.ndarray:
    idx = np.asarray(index, dtype=np.uint64)
    key = _pcg_hash(np.uint64((seed * 0x9E3779B1 + stream * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF))
    return _pcg_hash((idx + key) & _M32 ^ _pcg_hash((idx >> np.uint64(32)) + key + np.uint64(1)))


def rand_unit(seed: int, stream: int, index) -> np.ndarray:
    """uniform in [0,1) with 24 random bits (exact in float32)."""
    return (rand_u32(seed, stream, index) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def intrinsics(width: int, height: int) -> np.ndarray:
    """Kinect v1 nominal pinhole matrix scaled to the frame (Kinect/Kinect.cpp:89-95: fx = fy =
    ZPD/(2*ZPPS) = 120/(2*0.1042) at 640 px, cx = W/2, cy = H/2)."""
    f = 120.0 / (2.0 * 0.1042) * (width / 640.0)
    return np.array([[f, 0.0, width / 2.0], [0.0, f, height / 2.0], [0.0, 0.0, 1.0]], np.float64)


def make_frame(seed: int, width: int = 640, height: int = 480, n_rect: int = 24,
               clean: bool = False):
    """Returns (bgr uint8 [H,W,3], depth float32 [H,W] in mm; 0 = invalid).

    clean=True also returns the noise-free depth (the stand-in for main.cpp's `averaged_depth`).
    """
    H, W = height, width
    yy, xx = np.mgrid[0:H, 0:W]
    xf = xx.astype(np.float32)
    yf = yy.astype(np.float32)
    bgr = np.empty((H, W, 3), np.float32)
    bgr[..., 0] = 40.0 + 120.0 * xf / max(W - 1, 1)
    bgr[..., 1] = 60.0 + 100.0 * yf / max(H - 1, 1)
    bgr[..., 2] = 200.0 - 90.0 * (xf + yf) / max(W + H - 2, 1)
    depth = (3000.0 + 0.4 * xf * (640.0 / W) - 0.2 * yf * (480.0 / H)).astype(np.float32)
    valid = np.ones((H, W), bool)

    r = rand_unit(seed, 1, np.arange(n_rect * 16)).reshape(n_rect, 16)
    for k in range(n_rect):
        rw = int((0.05 + 0.35 * r[k, 0]) * W)
        rh = int((0.05 + 0.35 * r[k, 1]) * H)
        x0 = int(r[k, 2] * max(W - rw, 1))
        y0 = int(r[k, 3] * max(H - rh, 1))
        x1, y1 = min(x0 + max(rw, 1), W), min(y0 + max(rh, 1), H)
        col = np.floor(r[k, 4:7] * 256.0)
        base = 800.0 + 3200.0 * r[k, 7]
        sx = (r[k, 8] - 0.5) * (640.0 / W)
        sy = (r[k, 9] - 0.5) * (480.0 / H)
        band = max(int(round(8 * W / 640.0)), 1)
        valid[y0:y1, max(x0 - band, 0):x0] = False          # occlusion shadow left of the object
        bgr[y0:y1, x0:x1, :] = col
        depth[y0:y1, x0:x1] = (base + sx * (xf[y0:y1, x0:x1] - x0) + sy * (yf[y0:y1, x0:x1] - y0))
        valid[y0:y1, x0:x1] = True

    pix = np.arange(H * W, dtype=np.uint64).reshape(H, W)
    cn = np.stack([rand_u32(seed, 2 + c, pix) % np.uint64(9) for c in range(3)], -1).astype(np.float32) - 4.0
    bgr_u8 = np.clip(np.rint(bgr + cn), 0, 255).astype(np.uint8)

    truth = depth.copy()
    half = (0.45 * 2.85 * (depth / 10.0) ** 2 / 10000.0).astype(np.float32)
    u = rand_unit(seed, 5, pix) * 2.0 - 1.0
    noisy = (depth + half * u).astype(np.float32)
    salt = rand_unit(seed, 6, pix) < 0.01
    invalid = ~valid | salt
    noisy[invalid] = 0.0
    truth[~valid] = 0.0
    if clean:
        return bgr_u8, noisy, truth
    return bgr_u8, noisy


def make_batch(first_seed: int, n: int, width: int = 640, height: int = 480):
    """n frames with seeds first_seed .. first_seed+n-1: (bgr [n,H,W,3] u8, depth [n,H,W] f32)."""
    bgr = np.empty((n, height, width, 3), np.uint8)
    depth = np.empty((n, height, width), np.float32)
    for i in range(n):
        bgr[i], depth[i] = make_frame(first_seed + i, width, height)
    return bgr, depth
