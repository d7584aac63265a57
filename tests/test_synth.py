import numpy as np


def test_frames_are_deterministic_and_distinct(synth):
    a = synth.make_frame(7, 160, 120)
    b = synth.make_frame(7, 160, 120)
    c = synth.make_frame(8, 160, 120)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not np.array_equal(a[1], c[1])
    bgr, depth = a
    assert bgr.dtype == np.uint8 and bgr.shape == (120, 160, 3) and depth.dtype == np.float32
    inv = (depth == 0).mean()
    assert 0.005 < inv < 0.2                      # shadow bands + 1 % salt
    assert depth[depth > 0].min() > 50 and depth.max() < 15000


def test_noise_model_half_width(synth):
    bgr, noisy, clean = synth.make_frame(3, 160, 120, clean=True)
    m = (noisy > 0) & (clean > 0)
    half = 0.45 * 2.85 * (clean[m] / 10.0) ** 2 / 10000.0     # main.cpp:127-130
    assert np.all(np.abs(noisy[m] - clean[m]) <= half * 1.0001 + 1e-3)
    assert np.abs(noisy[m] - clean[m]).max() > 0.5 * half.max()


def test_intrinsics_scale_with_width(synth):
    K = synth.intrinsics(640, 480)
    assert abs(K[0, 0] - 575.8157) < 1e-3 and K[0, 2] == 320 and K[1, 2] == 240
    K2 = synth.intrinsics(1920, 1080)
    assert abs(K2[0, 0] - 3 * K[0, 0]) < 1e-9 and K2[0, 2] == 960 and K2[1, 2] == 540
