// kde_host_math.h — the host-side arithmetic behind the reference's "a factor that underflowed to exactly 0 is not
// multiplied in" rule (Q1) and the spatial table.  Pure C++ (no HIP): kde_api.cpp includes it, and
// tests/sanitize/host_driver.cpp compiles it with -fsanitize=address,undefined next to the CPU oracle.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace kde {

// smallest float x with exp(-x) rounding to 0 in IEEE binary32 (round to nearest): x > 150 ln 2
inline float exp_zero_threshold()
{
    // exp(-x) rounds to 0 in binary32 iff exp(-x) < 2^-150 iff x > 150 ln 2
    const double t = 150.0 * 0.693147180559945309417232121458;
    float f = (float)t;
    while ((double)f <= t) f = std::nextafterf(f, INFINITY);
    while ((double)std::nextafterf(f, 0.0f) > t) f = std::nextafterf(f, 0.0f);
    return f;
}

// calcSpatialFilter: JointBilateralFilter.cpp:31-40 / EdgeRefinedSuperpixel.cpp:46-55 (powf(x, 2.0f) written x*x)
inline void spatial_table(int window, float sigma, float* table)
{
    for (int i = 0; i < window; i++)
        for (int j = 0; j < window; j++) {
            const float fx = (float)(j - window / 2), fy = (float)(i - window / 2);
            const float dis_x = fx * fx, dis_y = fy * fy;
            table[i * window + j] = expf(-(dis_x + dis_y) / (2.0f * (sigma * sigma)));
        }
}

// smallest non-negative float q with q / den >= thr (float division); +inf if none
inline float smallest_q_reaching(float den, float thr)
{
    if (!(den > 0.0f)) return 0.0f;
    uint32_t lo = 0, hi = 0x7f800000u;   // bit patterns of +0 .. +inf are ordered like the values
    auto val = [](uint32_t b) { float f; memcpy(&f, &b, 4); return f; };
    if (!(val(hi) / den >= thr)) return INFINITY;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (val(mid) / den >= thr) hi = mid;
        else lo = mid + 1;
    }
    return val(lo);
}

// smallest integer colour distance cd in [0, 3*255^2] with (float)cd / den >= thr; 195076 (= 3*255^2 + 1) if none
inline int smallest_cd_reaching(float den, float thr)
{
    int lo = 0, hi = 195076;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if ((float)mid / den >= thr) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

}  // namespace kde
