"""Device wake-up for the measurement tools: an idle MI355X sits at its lowest clock level and needs ~100 ms of load to
reach the clock it then holds ("Clock ramp").  wake(torch) keeps the GPU busy for `ms` milliseconds
with untimed launches of the library's own VALU-bound filter (window 11 on a 16 x 640x480 batch) before anything is
measured."""
import time


def wake(torch, ms: float = 150.0) -> int:
    if ms <= 0:
        return 0
    from kinectdepthmapenhancement_amd import filters
    p = filters.JointBilateralFilter.default_params()
    p.window_size, p.spatial_sigma, p.color_sigma, p.depth_sigma, p.presmooth = 11, 3.0, 7.65, 20.0, 0
    jbf = filters.JointBilateralFilter(640, 480, p, max_batch=16)
    depth = torch.full((16, 480, 640), 1500.0, dtype=torch.float32, device="cuda")
    color = torch.full((16, 480, 640, 3), 90, dtype=torch.uint8, device="cuda")
    out = torch.empty_like(depth)
    n = 0
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(10):
            jbf.filter_batch(depth, color, out)
        torch.cuda.synchronize()
        n += 10
    return n
