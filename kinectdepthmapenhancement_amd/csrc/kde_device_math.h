// kde_device_math.h — small device functions shared by the product kernels and the test-hook library
// (tools/hooks/test_hooks.hip), so that a test can call exactly the code a kernel runs.
#pragma once
#include <hip/hip_runtime.h>

namespace kde {

// sqrtf() of an integer-valued float in [0, 2^24): v_rsq_f32 estimate, one Heron step on the exact fma residual.
// The library sqrtf() spends 14 instructions (denormal scaling, two +-1 ulp probes, class test) to be correctly
// rounded for every float; on this domain 6 are enough -- tests/test_gpu_dasp_ers.py checks ALL 2^24 arguments
// against sqrtf bit for bit (px*px + py*py of pixel offsets, the only argument K7 has, is such an integer).
__device__ __forceinline__ float sqrt_int24(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);          // 1/sqrt(x), 1 ulp; +inf for x == 0
    const float r = x * y;
    const float e = __builtin_fmaf(-r, r, x);          // exact residual x - r^2
    const float r1 = __builtin_fmaf(e, 0.5f * y, r);
    return x == 0.0f ? 0.0f : r1;
}

// expf() for the reference-shaped kernels.  The reference's rule "a factor that underflowed to exactly 0 is skipped"
//  is stated for an expf() that rounds to the binary32 denormal grid: 0 iff the true value is below 2^-150
// (x < -150 ln 2 = -103.972).  The library expf() returns 0 as soon as the true value is below the smallest denormal
// 2^-149 (x < -103.28), so a weight of half a unit to one unit of the grid would be lost.  Below -64 the argument is
// shifted by 64 and the (normal) result multiplied by e^-64: that product is rounded to the denormal grid once.
__device__ __forceinline__ float exp_denormal(float x)
{
    const bool shifted = x < -64.0f;
    const float r = expf(shifted ? x + 64.0f : x);
    return shifted ? r * 0x1.969d48p-93f : r;
}

// x / d for 0 <= x < 2^24 by a multiplication: with l = ceil(log2 d), m = floor(2^(24+l) / d) + 1 and sh = 24 + l,
// floor(x * m / 2^sh) == floor(x / d) (m * d = 2^sh + e with 0 < e <= 2^l, so the product overshoots x / d by less than
// 1 / d).  Two instructions (v_mad_u64_u32, v_lshrrev_b64) instead of the ~20 of a 32-bit division.  ok == 0 (operands
// beyond the proven range) falls back to the division.
struct FastDiv24 {
    uint32_t m, sh, d, ok;
};
inline FastDiv24 make_fastdiv24(uint32_t d, uint64_t max_dividend)
{
    FastDiv24 f{0u, 0u, d, 0u};
    if (d == 0 || d >= (1u << 24) || max_dividend >= (1ull << 24)) return f;
    uint32_t l = 0;
    while ((1u << l) < d) l++;
    f.sh = 24 + l;
    f.m = (uint32_t)((1ull << f.sh) / d + 1);      // < 2^(25+l) / d + 1 <= 2^25 + 1: fits
    f.ok = 1;
    return f;
}
__device__ __forceinline__ uint32_t fastdiv24(uint32_t x, const FastDiv24& f)
{
    return f.ok ? (uint32_t)(((uint64_t)x * f.m) >> f.sh) : x / f.d;
}

// Workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  Mapping workgroup `lin` of `nblk` to the
// tile xcd_band_id(lin, nblk) gives XCD k the k-th contiguous eighth of the (raster-ordered) tile list, so the halo
// rows two neighbouring tiles share are fetched into one L2 instead of two.  Bijective for any nblk.
__device__ __forceinline__ unsigned xcd_band_id(unsigned lin, unsigned nblk)
{
    const unsigned per = nblk / 8, rem = nblk % 8;
    const unsigned xcd = lin % 8, slot = lin / 8;
    return xcd * per + (xcd < rem ? xcd : rem) + slot;
}

}  // namespace kde
