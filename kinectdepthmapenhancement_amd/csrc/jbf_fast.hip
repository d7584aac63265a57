// jbf_fast.hip — tuned gfx950 variants of K1 (joint_bilateral_filtering,
// JointBilateralFilter/JointBilateralFilter.cu:4-83) for the common windows.
//
// The kernel is VALU/transcendental-bound (2 exp per tap against 11 B/pixel of traffic), so the work
// per (pixel, tap) is cut to the minimum the reference's semantics allow:
//   * colour distance as |a|^2 + |b|^2 - 2 a.b with ONE v_dot4_u32_u8 per tap on packed BGRX; -|b|^2 is
//     staged in LDS pre-biased so that (a.b << 1) + bias (one v_lshl_add_u32) IS the bit pattern of the
//     float 2^23 + 2^18 - |b|^2 + 2 a.b, i.e. no int->float convert;
//   * weights live in the log2 domain: arg1 = log2(S_ij) - kc*cd, arg2 = arg1 - kd*(d_q - avg)^2, one
//     v_exp_f32 per tap and pass; pass-1 arguments stay in registers for pass 2 (windows <= 7);
//   * taps that are invalid (depth <= 50 mm or outside the image) cost no branch or select: the scalar kernels
//     give them a bias of 0xFF000000 (-1.7e38, weight exactly 0), the packed kernels multiply their weight by
//     v = clamp(2 d) = 0 inside the accumulating fma;
//   * the reference's "a factor that underflowed to exactly 0 is not multiplied in" rule (Q1) is decided
//     on the ARGUMENT (integer colour distance / |d_q - avg|) against host-computed thresholds, so it does
//     not depend on how small numbers are rounded by the exp hardware; in the packed kernels the decision is
//     a {0,1} mask produced by the VOP3P clamp bit (no v_cmp / v_cndmask);
//   * every weight is formed at 2^24 times its value (kScaleLog2 is folded into the log2(S) table on the host, so it
//     costs nothing): v_exp_f32 flushes results below 2^-126, which at this scale is exactly the float32
//     underflow-to-zero point 2^-150 of the reference's product S*cf*df -- weights the reference still represents
//     (as denormals) are kept with full precision, weights it rounds to 0 vanish here too, and the two decisions
//     "sum of weights > 0" / "denominator == 0" (JointBilateralFilter.cu:39,76) fall where the reference's fall.
//     The scale cancels in both quotients;
//   * each thread owns PX horizontally adjacent pixels and walks the window row by row from registers.
// Depth + guide + |b|^2 tiles (with halo) are staged once per workgroup in LDS; the workgroup->tile map
// keeps each XCD on a contiguous band of tiles so halos are shared in that XCD's L2.
#include "kde_internal.h"
#include "kde_device_math.h"

#include <cstddef>
#include <type_traits>

namespace kde {
namespace {

constexpr uint32_t kMagic = 0x4B000000u;      // float 2^23
constexpr uint32_t kOff = 1u << 18;           // keeps 2 a.b - |b|^2 + kOff positive (|b|^2 <= 195075 < 2^18)
constexpr uint32_t kInvalidBias = 0xFF000000u; // as a float: -1.7e38; as an unsigned int: above every valid code
constexpr float kBiasF = 8388608.0f + 262144.0f;
constexpr double kScaleLog2 = 24.0;           // weights are summed at 2^24 scale (see the header comment)

struct FastArgs {
    const float* depth;
    const uint8_t* guide;
    float* out;
    int width, height, n;
    int tiles_x, tiles_y;
    FastDiv24 div_tiles, div_tx;   // workgroup -> (frame, tile row, tile column) by multiplication (kde_device_math.h)
    float kc;        // log2(e) / (2 sigma_c^2)
    float sd;        // sqrt(log2(e) / (2 sigma_d^2))
    float t_skip;    // depth factor skipped when |d_q - avg| * sd >= t_skip
    float t2_skip;   // t_skip^2 (3e38 when nothing is skipped) for the packed-arithmetic form of the rule
    int cd_skip;     // colour factor skipped when cd >= cd_skip
    int vec4;        // width % 4 == 0 and 16-byte aligned frames: the packed kernels load 4-pixel groups
    KDE_STAGE(float* stage_avg; unsigned* stage_counters; int stage_force;)   // tools/hooks/libkde_hip_stage.so only
    // log2 of the spatial table (zeros of the table -> factor skipped -> log2 = 0).  Scalar kernels read
    // tab[i*WIN + j]; packed kernels read the pair (ls[i][j of p0], ls[i][j of p1]) of unit u at
    // tab[(i*WIN + u)*2 .. +1], so the addend of the argument fma is one aligned SGPR pair.
    // 21 x 21 x 2 floats.  The whole struct is the kernel's argument block and stays below HIP's 4 KB limit for it (window 23
    // would need 4232 bytes for the table alone): windows 23..31 read the same pairs from tab_dev, a device copy the handle
    // uploaded once (still wave-uniform addresses: scalar loads)
    const float* tab_dev;
    __attribute__((aligned(8))) float tab[882];
};

__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot4(a, b, c, false);
}


template <int WIN, int PX, int BX, int BY, bool CACHE, bool CSKIP>
__global__ __launch_bounds__(BX* BY) void jbf_fast_kernel(const FastArgs a)
{
    constexpr int R = WIN / 2;
    constexpr int NT = BX * BY;
    constexpr int TW = BX * PX, TH = BY;
    constexpr int LW = TW + 2 * R, LH = TH + 2 * R;
    constexpr int P = LW;              // row pitch in dwords
    constexpr int SEG = PX + 2 * R;    // row segment a thread needs
    __shared__ float s_d[LH * P];
    __shared__ uint32_t s_c[LH * P];
    __shared__ uint32_t s_n[LH * P];

    // ---- workgroup -> (tile, frame): XCD k works on the k-th contiguous eighth of the tile list ----
    const unsigned nblk = gridDim.x;
    const unsigned lin = blockIdx.x;
    const unsigned per = nblk / 8, rem = nblk % 8;
    const unsigned xcd = lin % 8, slot = lin / 8;
    const unsigned id = xcd * per + (xcd < rem ? xcd : rem) + slot;   // bijective for any nblk
    const unsigned tiles = (unsigned)a.tiles_x * a.tiles_y;
    const unsigned frame_i = fastdiv24(id, a.div_tiles);
    const unsigned tile = id - frame_i * tiles;
    const int tyi = (int)fastdiv24(tile, a.div_tx), txi = tile - tyi * a.tiles_x;
    const int x0 = txi * TW, y0 = tyi * TH;

    const size_t frame = (size_t)frame_i * a.width * a.height;
    const float* __restrict__ depth = a.depth + frame;
    const uint8_t* __restrict__ guide = a.guide + frame * 3;
    const int tid = threadIdx.x;

    for (int i = tid; i < LW * LH; i += NT) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float d = 0.0f;
        uint32_t c = 0;
        if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
            const size_t q = (size_t)gy * a.width + gx;
            d = depth[q];
            const uint8_t* p = guide + q * 3;
            c = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
        }
        const bool valid = d > 50.0f;
        s_d[ly * P + lx] = valid ? d : 0.0f;
        s_c[ly * P + lx] = c;
        s_n[ly * P + lx] = valid ? (kMagic + kOff) - dot4(c, c, 0) : kInvalidBias;
    }
    __syncthreads();

    const int tx = tid % BX, ty = tid / BX;
    const int xb = x0 + tx * PX, y = y0 + ty;
    if (xb >= a.width || y >= a.height) return;

    uint32_t cc[PX];
    float cA[PX], skipA[PX];
    uint32_t thrU[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) {
        cc[k] = s_c[(ty + R) * P + tx * PX + R + k];
        const uint32_t na = dot4(cc[k], cc[k], 0);
        cA[k] = -a.kc * (kBiasF + (float)na);                              // exact integer < 2^24, one rounding
        // cA carries that one rounding error into every tap of this pixel as a common factor 2^delta, which
        // cancels in sum(f*d)/sum(f) -- provided a SKIPPED colour factor (exactly 1) carries it too:
        skipA[k] = __builtin_fmaf(kBiasF + (float)na, a.kc, cA[k]);        // = delta (exact product error)
        thrU[k] = CSKIP ? (kMagic + kOff) + na - (uint32_t)a.cd_skip : 0;   // u <= thrU  <=>  cd >= cd_skip
    }

    float arg[CACHE ? WIN * WIN * PX : 1];
    float wsum[PX], wgt[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) wsum[k] = wgt[k] = 0.0f;

    // one tap of pass 1: returns the log2-domain argument of S*cf (or -huge for an invalid tap)
    auto arg1_of = [&](uint32_t cq, uint32_t nq, int k, float ls) -> float {
        // u = bits of the float 2^23 + 2^18 + 2 a.b - |b|^2  (v_dot4_u32_u8 + v_lshl_add_u32)
        const uint32_t u = (dot4(cq, cc[k], 0) << 1) + nq;
        float a1 = __builtin_fmaf(__uint_as_float(u), a.kc, cA[k]);          // = -kc * cd
        if (CSKIP) a1 = (u <= thrU[k]) ? skipA[k] : a1;   // Q1: underflowed colour factor skipped (invalid codes compare above)
        return a1 + ls;
    };

    auto pass1_row = [&](int i) {
        float dr[SEG];
        uint32_t cr[SEG], nr[SEG];
        const int rb = (ty + i) * P + tx * PX;
#pragma unroll
        for (int q = 0; q < SEG; q++) {
            dr[q] = s_d[rb + q];
            cr[q] = s_c[rb + q];
            nr[q] = s_n[rb + q];
        }
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            const float ls = a.tab[i * WIN + j];
#pragma unroll
            for (int k = 0; k < PX; k++) {
                const float a1 = arg1_of(cr[j + k], nr[j + k], k, ls);
                if (CACHE) arg[(i * WIN + j) * PX + k] = a1;
                const float f = __builtin_amdgcn_exp2f(a1);
                wsum[k] = __builtin_fmaf(dr[j + k], f, wsum[k]);
                wgt[k] += f;
            }
        }
    };
    if (CACHE) {
#pragma unroll
        for (int i = 0; i < WIN; i++) pass1_row(i);
    } else {
#pragma unroll 1
        for (int i = 0; i < WIN; i++) pass1_row(i);
    }

    float c2[PX], num[PX], den[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) {
        c2[k] = wsum[k] / wgt[k];                // window average; wgt == 0 handled at the store
        num[k] = den[k] = 0.0f;
    }

    auto pass2_row = [&](int i) {
        float dr[SEG];
        uint32_t cr[SEG], nr[SEG];
        const int rb = (ty + i) * P + tx * PX;
#pragma unroll
        for (int q = 0; q < SEG; q++) {
            dr[q] = s_d[rb + q];
            if (!CACHE) {
                cr[q] = s_c[rb + q];
                nr[q] = s_n[rb + q];
            }
        }
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            const float ls = CACHE ? 0.0f : a.tab[i * WIN + j];
#pragma unroll
            for (int k = 0; k < PX; k++) {
                const float a1 = CACHE ? arg[(i * WIN + j) * PX + k] : arg1_of(cr[j + k], nr[j + k], k, ls);
                // (d_q - avg) first, THEN scale: fma(d_q, sd, -avg*sd) loses ~ulp(avg*sd) and that error is
                // not common to the taps of a pixel
                const float t = (dr[j + k] - c2[k]) * a.sd;
                float a2 = __builtin_fmaf(-t, t, a1);
                a2 = (__builtin_fabsf(t) >= a.t_skip) ? a1 : a2;     // Q1: underflowed depth factor is skipped
                const float f = __builtin_amdgcn_exp2f(a2);
                num[k] = __builtin_fmaf(dr[j + k], f, num[k]);
                den[k] += f;
            }
        }
    };
    if (CACHE) {
#pragma unroll
        for (int i = 0; i < WIN; i++) pass2_row(i);
    } else {
#pragma unroll 1
        for (int i = 0; i < WIN; i++) pass2_row(i);
    }

    float* __restrict__ o = a.out + frame + (size_t)y * a.width + xb;
#pragma unroll
    for (int k = 0; k < PX; k++) {
        float r = 0.0f;
        if (wgt[k] > 0.0f) r = (den[k] == 0.0f) ? 0.0f : num[k] / den[k];
        if (xb + k < a.width) o[k] = r;
        KDE_STAGE(if (a.stage_avg && xb + k < a.width) a.stage_avg[frame + (size_t)y * a.width + xb + k] = c2[k];)
    }
}


// ---------------------------------------------------------------------------------------------------
// Packed-math form.  A wave64 f32 VALU instruction costs 4 cycles on a gfx950 SIMD whether it is scalar
// or v_pk_*_f32, so every float add/mul/fma below works on a PAIR of horizontally adjacent pixels
// (p0 = even column, p1 = p0 + 1) and is written on 2-vectors; the row segment arrives from LDS as
// aligned 64-bit pairs (col 2m, col 2m+1).  A "unit" = one tap for each pixel of the pair, chosen so that
// both taps sit in ONE aligned pair of the segment:
//    straight  (j even):  p0 tap j   = lo(m),  p1 tap j     = hi(m)
//    swapped   (j odd>1): p0 tap j   = hi(m),  p1 tap j - 2 = lo(m)      (op_sel swap, no data movement)
//    leftover:            p0 tap 1   = hi(m0), p1 tap WIN-2 = lo(m1)     (one v_pk_mov_b32)
// The spatial weight costs nothing per tap: log2(S[i][j]) of the unit's two taps is the (SGPR-pair) addend of
// the argument fma.
// Per unit (= 2 taps), window 11, in 4-cycle issue slots (v_exp_f32 counts 2):
//   pass 1: 2 v_dot4 + 2 v_lshl_add + 2 v_exp + 6 packed (-cd, mask, mask*kc, arg fma, 2 accumulating fma) = 14.3
//   pass 2: the same argument (8) + 2 v_exp + 8 packed (d-avg, *sd, T2-t^2, mask, t*mask, arg fma, 2 acc.)   = 20.3
// (the select form of the Q1 rules cost 17 + 24: profiles/r01_sweep_k1_clamp_form.log has the A/B.)
// ---------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 bcast(float v) { return f2{v, v}; }
// packed add / multiply with the VOP3P clamp bit: the result is clamped to [0, 1] (NaN -> 0, DX10_CLAMP is on)
__device__ __forceinline__ f2 pk_add_clamp(f2 a, f2 b)
{
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 pk_mul_clamp(f2 a, f2 b)
{
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// second launch bound = waves per SIMD the register allocator must leave room for: the four rule-specialised bodies
// below share one kernel, and without it the window-19 instance grew from 131 to 182 VGPRs (3 -> 2 waves per SIMD)
// (windows >= 23: at least 3 waves per SIMD, i.e. <= 168 VGPRs -- without the bound the instances without the colour rule
//  came out at 228 / 248 registers, 2 waves per SIMD)
// CR = window rows whose pass-1 arguments are kept for pass 2 (CACHE: all of them, the windows <= 7; 0: recomputed).  0 < CR < WIN
// is r05's partial form for windows 9..13: the first CR rows are unrolled and their arguments stay in registers (two waves per
// SIMD asked for: 256 registers per lane), the remaining rows run the recomputing loop.  Measured slower (see kVariants): only
// the measurement build instantiates it.
template <int WIN, int NP, int BX, int BY, bool CACHE, bool CSKIP, bool VL, bool ELIDE_ON = true, int CR = (CACHE ? WIN : 0)>
__global__ __launch_bounds__(BX* BY) __attribute__((amdgpu_waves_per_eu(WIN >= 23 ? 3 : ((CR > 0 && !CACHE) ? 2 : 1)))) void jbf_pk_kernel(const FastArgs a)
{
    static_assert(CR >= 0 && CR <= WIN && (!CACHE || CR == WIN), "cached rows");
    constexpr int R = WIN / 2;
    constexpr int PX = 2 * NP;
    constexpr int NT = BX * BY;
    constexpr int TW = BX * PX, TH = BY;
    constexpr int LH = TH + 2 * R;
    // LDS column 0 is image column o = xb0 - RA with xb0 = (tile index)*TW - S1.  S1 shifts the tiles one
    // column left when R is odd so that a thread's first window column (xb - R) is even (pairs stay aligned),
    // and RA >= R is chosen so that o is a multiple of 4: the loader then moves 4-pixel groups with
    // 16-byte depth loads and 12-byte colour loads that are aligned in the image.
    // VL = false keeps the plain geometry (RA = R, one pixel per loader iteration) for A/B measurements.
    // (S1 = 1 would shift the tiles one column left for odd radii to keep the pairs 8-byte aligned; measured
    //  slower than unaligned pair reads because of the extra, nearly empty tile column it costs)
    constexpr int S1 = 0;
    constexpr int RA = !VL ? R : ((R + 3) / 4 * 4);
    // SH: with the vector loader an odd RA - R (radii 5, 9: windows 11, 19) would start every thread's row segment on an odd
    // LDS column and turn the ds_read_b64 of the pass loops into pairs of ds_read_b32 (measured 5-19 % slower in r02).
    // The staged rows are therefore shifted right by one dword: the global loads stay 16-byte / 12-byte groups aligned in
    // the image, the LDS writes of a group become four single dwords, and the pass loops keep their 64-bit reads.
    constexpr int SH = VL ? ((RA - R) & 1) : 0;
    constexpr bool ALIGNED = ((RA - R + SH) % 2) == 0;     // row segments start on an even LDS column
    constexpr int G = (RA + TW + R + 3) / 4;          // 4-pixel groups per staged row (vector loader)
    constexpr int P = VL ? (SH + 4 * G + 1) / 2 * 2 : (TW + 2 * R);   // row pitch in dwords (even; a multiple of 4 when SH == 0)
    constexpr int SEGP = NP + R;                      // aligned pairs in the row segment of a thread
    constexpr int HALF = (WIN - 1) / 2;
    static_assert(RA >= R && P % 2 == 0 && (!VL || ((RA + S1) % 4 == 0 && TW % 4 == 0 && (SH != 0 || P % 4 == 0))), "tile geometry");
    __shared__ __attribute__((aligned(16))) float s_d[LH * P];
    __shared__ __attribute__((aligned(16))) uint32_t s_c[LH * P];
    __shared__ __attribute__((aligned(16))) uint32_t s_n[LH * P];

    const unsigned nblk = gridDim.x;
    const unsigned lin = blockIdx.x;
    const unsigned per = nblk / 8, rem = nblk % 8;
    const unsigned xcd = lin % 8, slot = lin / 8;
    const unsigned id = xcd * per + (xcd < rem ? xcd : rem) + slot;
    const unsigned tiles = (unsigned)a.tiles_x * a.tiles_y;
    const unsigned frame_i = fastdiv24(id, a.div_tiles);
    const unsigned tile = id - frame_i * tiles;
    const int tyi = (int)fastdiv24(tile, a.div_tx), txi = tile - tyi * a.tiles_x;
    const int x0 = txi * TW - S1, y0 = tyi * TH;
    const int o = x0 - RA;

    const size_t frame = (size_t)frame_i * a.width * a.height;
    const float* __restrict__ depth = a.depth + frame;
    const uint8_t* __restrict__ guide = a.guide + frame * 3;
    const int tid = threadIdx.x;

    // Tile statistics for the rule elision below: range of the valid depths and per-channel range of the colours of
    // the in-image pixels this workgroup stages (tile + halo = every tap and every centre the workgroup can see).
    // Positive floats order like their bit patterns, so everything is reduced as uint32.
    // (Only for windows >= 9: at windows 5 and 7 a thread has 25 / 49 units of work and the statistics cost more than
    //  the elided instructions return -- measured 33.9 vs 28.8 us on one 1080p frame at window 5.)
    // ELIDE_ON = false ("-noelide" variants): the same kernel without the statistics and with the full-rule body everywhere --
    // the data-independent floor of the window, measured next to the default by bench.py
    constexpr bool ELIDE = WIN >= 9 && ELIDE_ON;
    __shared__ uint32_t s_stat[8];           // dmin, dmax, (min, max) of b, g, r
    if (ELIDE) {
        if (tid < 8) s_stat[tid] = (tid == 0 || (tid >= 2 && !(tid & 1))) ? 0xffffffffu : 0u;
        __syncthreads();                     // the initial values are in place before any wavefront's atomics below
    }
    uint32_t st_dmin = 0x7f800000u, st_dmax = 0u, st_cmin[3] = {255u, 255u, 255u}, st_cmax[3] = {0u, 0u, 0u};
    auto stat = [&](float dv, uint32_t c, bool inside) {        // dv: staged depth (0 = invalid)
        if (!ELIDE) return;
        if (dv > 0.0f) {
            st_dmin = min(st_dmin, __float_as_uint(dv));
            st_dmax = max(st_dmax, __float_as_uint(dv));
        }
        if (inside) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const uint32_t v = (c >> (8 * ch)) & 0xffu;
                st_cmin[ch] = min(st_cmin[ch], v);
                st_cmax[ch] = max(st_cmax[ch], v);
            }
        }
    };

    if (!VL) {
        for (int i = tid; i < P * LH; i += NT) {
            const int ly = i / P, lx = i - ly * P;
            const int gx = o + lx, gy = y0 - R + ly;
            float d = 0.0f;
            uint32_t c = 0;
            const bool inside = gx >= 0 && gx < a.width && gy >= 0 && gy < a.height;
            if (inside) {
                const size_t q = (size_t)gy * a.width + gx;
                d = depth[q];
                const uint8_t* p = guide + q * 3;
                c = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
            }
            const bool valid = d > 50.0f;
            s_d[i] = valid ? d : 0.0f;
            s_c[i] = c;
            s_n[i] = (kMagic + kOff) - dot4(c, c, 0);
            stat(valid ? d : 0.0f, c, inside);
        }
    }
    for (int g = tid; VL && g < G * LH; g += NT) {
        const int ly = g / G, gi = g - ly * G;
        const int gx = o + 4 * gi, gy = y0 - R + ly;
        float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        uint32_t c[4] = {0u, 0u, 0u, 0u};
        if (gy >= 0 && gy < a.height) {
            const size_t q = (size_t)gy * a.width + gx;
            if (a.vec4 && gx >= 0 && gx + 3 < a.width) {
                const float4 dv = *reinterpret_cast<const float4*>(depth + q);
                const uint32_t* cp = reinterpret_cast<const uint32_t*>(guide + q * 3);
                const uint32_t w0 = cp[0], w1 = cp[1], w2 = cp[2];      // 4 packed BGR pixels
                d[0] = dv.x; d[1] = dv.y; d[2] = dv.z; d[3] = dv.w;
                c[0] = w0 & 0xffffffu;
                c[1] = (w0 >> 24) | ((w1 & 0xffffu) << 8);
                c[2] = (w1 >> 16) | ((w2 & 0xffu) << 16);
                c[3] = w2 >> 8;
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (gx + k >= 0 && gx + k < a.width) {
                        d[k] = depth[q + k];
                        const uint8_t* p = guide + (q + k) * 3;
                        c[k] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
                    }
                }
            }
        }
        uint32_t nn[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool valid = d[k] > 50.0f;
            d[k] = valid ? d[k] : 0.0f;
            nn[k] = (kMagic + kOff) - dot4(c[k], c[k], 0);
            stat(d[k], c[k], gy >= 0 && gy < a.height && gx + k >= 0 && gx + k < a.width);
        }
        const int li = ly * P + 4 * gi + SH;
        if constexpr (SH == 0) {
            *reinterpret_cast<float4*>(&s_d[li]) = make_float4(d[0], d[1], d[2], d[3]);
            *reinterpret_cast<uint4*>(&s_c[li]) = make_uint4(c[0], c[1], c[2], c[3]);
            *reinterpret_cast<uint4*>(&s_n[li]) = make_uint4(nn[0], nn[1], nn[2], nn[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                s_d[li + k] = d[k];
                s_c[li + k] = c[k];
                s_n[li + k] = nn[k];
            }
        }
    }
    if (ELIDE) {   // wavefront reduction (xor butterflies), then one LDS atomic per wavefront and statistic
        uint32_t v[8] = {st_dmin, st_dmax, st_cmin[0], st_cmax[0], st_cmin[1], st_cmax[1], st_cmin[2], st_cmax[2]};
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const uint32_t ov = (uint32_t)__shfl_xor((int)v[k], m, 64);
                v[k] = (k & 1) ? max(v[k], ov) : min(v[k], ov);
            }
        }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (k & 1) atomicMax(&s_stat[k], v[k]);
                else atomicMin(&s_stat[k], v[k]);
            }
        }
    }
    __syncthreads();

    // Rule elision (workgroup-uniform): the reference's "factor that underflowed to 0 is skipped" tests (Q1) cost 2 packed
    // instructions per unit and pass for the colour rule and 3 per unit for the depth rule.  Neither can trip anywhere
    // in this tile when the tile's own value ranges stay below the thresholds:
    //   colour: cd(p, q) <= sum_c (max_c - min_c)^2 over the staged pixels           < cd_skip
    //   depth:  |d_q - avg_p| <= max - min of the staged valid depths (avg_p is a mean of them), x sd  < t_skip
    // The full-rule code is kept for the tiles that straddle a colour or depth edge.
    bool tile_c = CSKIP, tile_d = true;
    if (ELIDE) {
        const int rb = (int)s_stat[3] - (int)s_stat[2], rg = (int)s_stat[5] - (int)s_stat[4], rr = (int)s_stat[7] - (int)s_stat[6];
        if (s_stat[3] >= s_stat[2]) tile_c = CSKIP && (rb * rb + rg * rg + rr * rr >= a.cd_skip);
        const float range = __uint_as_float(s_stat[1]) - __uint_as_float(s_stat[0]);       // -inf when no valid depth was staged
        // avg_p is a float32 quotient of float32 sums: it can leave [dmin, dmax] by the rounding of those sums, at most
        // about (taps + 8) / 2 ulps of dmax -- an absolute slack that a purely relative margin on the range does not cover
        // when the depths are large and sigma_d is small (1e4 mm at sigma_d = 1 mm)
        const float slack = __uint_as_float(s_stat[1]) * (float)(WIN * WIN + 8) * 0x1p-24f;
        tile_d = !((range * 1.0001f + slack) * a.sd < a.t_skip);
    }
    KDE_STAGE(if (a.stage_force) { tile_c = CSKIP; tile_d = true; })
    KDE_STAGE(if (a.stage_counters && tid == 0) {
        // the body the dispatch below picks for this tile: bit 0 = colour rule compiled in, bit 1 = depth rule
        int body;
        if constexpr (!ELIDE) body = (CSKIP ? 1 : 0) + 2;
        else if constexpr (WIN >= 15) body = ((CSKIP && tile_c) || tile_d) ? (CSKIP ? 1 : 0) + 2 : 0;
        else body = ((CSKIP && tile_c) ? 1 : 0) + (tile_d ? 2 : 0);
        atomicAdd(&a.stage_counters[body], 1u);
    })

    const int tx = tid % BX, ty = tid / BX;
    const int xb = x0 + tx * PX, y = y0 + ty;
    if (xb >= a.width || y >= a.height) return;
    const int sx = SH + RA - R + tx * PX;  // LDS column of this thread's first window column (even)

    uint32_t cc[PX];
    f2 negC[NP];          // -(2^23 + 2^18 + |a|^2): F + negC = -cd exactly (integers < 2^24)
#pragma unroll
    for (int pp = 0; pp < NP; pp++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = 2 * pp + h;
            cc[k] = s_c[(ty + R) * P + sx + R + k];
            const uint32_t na = dot4(cc[k], cc[k], 0);
            negC[pp][h] = -(kBiasF + (float)na);
        }
    }

    f2 kc2 = bcast(a.kc);
    asm volatile("" : "+v"(kc2));   // keep kc in a VGPR pair: the fma's only scalar operand is then the log2(S) SGPR pair
    f2 arg[CR > 0 ? CR * WIN * NP : 1];
    f2 wsum[NP], wgt[NP];
#pragma unroll
    for (int pp = 0; pp < NP; pp++) wsum[pp] = wgt[pp] = bcast(0.0f);

    // Q1 rules as packed arithmetic on {0,1} masks made by the clamp bit:
    //   colour: m = clamp(cd_skip - cd) is 1 for cd < cd_skip and 0 otherwise (integers), a1 = fma(-cd, kc*m, ls)
    //   depth:  q = T2 - t^2 (one rounding, sign exact), m = clamp(q * 2^100), a2 = fma(-(t*m), t, a1)
    //   invalid taps: colour code and depth 0 stay finite, and the tap enters the sums with v = clamp(2 d) = 0
    const f2 Tc = bcast((float)a.cd_skip);
    const f2 T2 = bcast(a.t2_skip);
    const f2 kBig = bcast(0x1p100f);

    // argument (log2 domain, without the row factor) of S*cf for the two taps of unit u of pair pp
    auto unit_arg = [&](auto cs_tag, const u2* cp, const u2* np, int pp, int i, int u) -> f2 {
        constexpr bool CS = decltype(cs_tag)::value;
        uint32_t c0, c1, n0, n1;
        if (u <= HALF) {                       // straight
            c0 = cp[pp + u].x; c1 = cp[pp + u].y; n0 = np[pp + u].x; n1 = np[pp + u].y;
        } else if (u < WIN - 1) {              // swapped
            const int m = pp + (u - HALF);
            c0 = cp[m].y; c1 = cp[m].x; n0 = np[m].y; n1 = np[m].x;
        } else {                               // leftover
            c0 = cp[pp].y; c1 = cp[pp + HALF].x; n0 = np[pp].y; n1 = np[pp + HALF].x;
        }
        const uint32_t u0 = (dot4(c0, cc[2 * pp], 0) << 1) + n0;
        const uint32_t u1 = (dot4(c1, cc[2 * pp + 1], 0) << 1) + n1;
        // -cd exactly, then ONE rounding in the fma: a1 = log2(S[i][j]) - kc*cd
        const f2 ncd = f2{__uint_as_float(u0), __uint_as_float(u1)} + negC[pp];
        f2 lsj;
        if constexpr (WIN > 21) {
            // the device copy is read through the CONSTANT address space: never written while the kernel runs, wave-uniform
            // address -> one s_load per pair, into the SGPR pair the argument fma takes (a plain global pointer made the
            // compiler issue per-lane global_load_dwordx4 and hold the pairs in VGPRs: 248 instead of ~150 registers)
            typedef const f2 __attribute__((address_space(4)))* cf2p;
            lsj = *(cf2p)(uintptr_t)(a.tab_dev + (i * WIN + u) * 2);
        } else if constexpr (CR > 0 && !CACHE) {
            // the partial-keep instances: read the pair straight from the kernel-argument segment (the argument block IS
            // FastArgs).  Through `a.tab` the compiler copied the whole 3.6 KB block to scratch in the instances with the colour rule.
            typedef const f2 __attribute__((address_space(4)))* cf2p;
            typedef const char __attribute__((address_space(4)))* ccp;
            lsj = *(cf2p)((ccp)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(FastArgs, tab) + (size_t)((i * WIN + u) * 2) * sizeof(float));
        } else {
            lsj = *reinterpret_cast<const f2*>(&a.tab[(i * WIN + u) * 2]);
        }
        if (CS) return pk_fma(ncd, pk_add_clamp(ncd, Tc) * kc2, lsj);   // Q1: underflowed colour factor skipped
        return pk_fma(ncd, kc2, lsj);
    };
    auto unit_depth = [&](const f2* dp, int pp, int u) -> f2 {
        if (u <= HALF) return dp[pp + u];
        if (u < WIN - 1) return __builtin_shufflevector(dp[pp + (u - HALF)], dp[pp + (u - HALF)], 1, 0);
        return f2{dp[pp].y, dp[pp + HALF].x};
    };

    auto pass1_row = [&](auto cs_tag, auto keep_tag, int i) {
        constexpr bool KEEP = decltype(keep_tag)::value;
        f2 dp[SEGP], vp[SEGP];
        u2 cp[SEGP], np[SEGP];
        const int rb = (ty + i) * P + sx;
#pragma unroll
        for (int m = 0; m < SEGP; m++) {
            dp[m] = ALIGNED ? *reinterpret_cast<const f2*>(&s_d[rb + 2 * m]) : f2{s_d[rb + 2 * m], s_d[rb + 2 * m + 1]};
            cp[m] = ALIGNED ? *reinterpret_cast<const u2*>(&s_c[rb + 2 * m]) : u2{s_c[rb + 2 * m], s_c[rb + 2 * m + 1]};
            np[m] = ALIGNED ? *reinterpret_cast<const u2*>(&s_n[rb + 2 * m]) : u2{s_n[rb + 2 * m], s_n[rb + 2 * m + 1]};
            vp[m] = pk_add_clamp(dp[m], dp[m]);      // 1 for a valid tap (d > 50), 0 for d == 0
        }
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
#pragma unroll
            for (int u = 0; u < WIN; u++) {
                const f2 a1 = unit_arg(cs_tag, cp, np, pp, i, u);
                if constexpr (KEEP) arg[(i * NP + pp) * WIN + u] = a1;
                const f2 f = f2{__builtin_amdgcn_exp2f(a1.x), __builtin_amdgcn_exp2f(a1.y)};
                wsum[pp] = pk_fma(unit_depth(dp, pp, u), f, wsum[pp]);
                wgt[pp] = pk_fma(unit_depth(vp, pp, u), f, wgt[pp]);
            }
        }
    };
    f2 c2[NP], num[NP], den[NP];

    auto pass2_row = [&](auto cs_tag, auto ds_tag, auto keep_tag, int i) {
        constexpr bool DS = decltype(ds_tag)::value;
        constexpr bool KEEP = decltype(keep_tag)::value;
        f2 dp[SEGP], vp[SEGP];
        u2 cp[SEGP], np[SEGP];
        const int rb = (ty + i) * P + sx;
#pragma unroll
        for (int m = 0; m < SEGP; m++) {
            dp[m] = ALIGNED ? *reinterpret_cast<const f2*>(&s_d[rb + 2 * m]) : f2{s_d[rb + 2 * m], s_d[rb + 2 * m + 1]};
            vp[m] = pk_add_clamp(dp[m], dp[m]);
            if constexpr (!KEEP) {
                cp[m] = ALIGNED ? *reinterpret_cast<const u2*>(&s_c[rb + 2 * m]) : u2{s_c[rb + 2 * m], s_c[rb + 2 * m + 1]};
                np[m] = ALIGNED ? *reinterpret_cast<const u2*>(&s_n[rb + 2 * m]) : u2{s_n[rb + 2 * m], s_n[rb + 2 * m + 1]};
            }
        }
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
#pragma unroll
            for (int u = 0; u < WIN; u++) {
                f2 a1;
                if constexpr (KEEP) a1 = arg[(i * NP + pp) * WIN + u];
                else a1 = unit_arg(cs_tag, cp, np, pp, i, u);
                const f2 dq = unit_depth(dp, pp, u);
                const f2 t = (dq - c2[pp]) * bcast(a.sd);    // subtract first: see the scalar kernel
                f2 a2;
                if (DS) {
                    const f2 m = pk_mul_clamp(pk_fma(-t, t, T2), kBig);    // Q1: underflowed depth factor skipped
                    a2 = pk_fma(-(t * m), t, a1);
                } else {
                    a2 = pk_fma(-t, t, a1);                                  // the tile's depth range cannot reach the threshold
                }
                const f2 f = f2{__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                num[pp] = pk_fma(dq, f, num[pp]);
                den[pp] = pk_fma(unit_depth(vp, pp, u), f, den[pp]);
            }
        }
    };
    // both passes, specialised on which Q1 rules this tile needs
    auto run = [&](auto cs_tag, auto ds_tag) {
        // rows [0, CR): unrolled, arguments kept; rows [CR, WIN): the recomputing loop
#pragma unroll
        for (int i = 0; i < CR; i++) pass1_row(cs_tag, std::true_type{}, i);
#pragma unroll 1
        for (int i = CR; i < WIN; i++) pass1_row(cs_tag, std::false_type{}, i);
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
            // window averages (wgt == 0 handled at the store); v_rcp_f32 is 1 ulp, the same order as the
            // summation-order noise already present in wsum
            c2[pp] = wsum[pp] * f2{__builtin_amdgcn_rcpf(wgt[pp].x), __builtin_amdgcn_rcpf(wgt[pp].y)};
            num[pp] = den[pp] = bcast(0.0f);
        }
#pragma unroll
        for (int i = 0; i < CR; i++) pass2_row(cs_tag, ds_tag, std::true_type{}, i);
#pragma unroll 1
        for (int i = CR; i < WIN; i++) pass2_row(cs_tag, ds_tag, std::false_type{}, i);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    // Four bodies for the small windows.  For the wide ones only {both rules, no rule}: a body with exactly one rule made
    // the window-19 instance 182-188 VGPRs instead of 131 (2 instead of 3 waves per SIMD), which costs more than the
    // partial elision gains.
    if constexpr (!ELIDE) {
        run(std::integral_constant<bool, CSKIP>{}, T_{});
    } else if constexpr (WIN >= 15) {
        if ((CSKIP && tile_c) || tile_d) run(std::integral_constant<bool, CSKIP>{}, T_{});
        else run(F_{}, F_{});
    } else {
        if (!CSKIP || !tile_c) {
            if (tile_d) run(F_{}, T_{});
            else run(F_{}, F_{});
        } else {
            if (tile_d) run(T_{}, T_{});
            else run(T_{}, F_{});
        }
    }

    float* __restrict__ op = a.out + frame + (size_t)y * a.width + xb;
#pragma unroll
    for (int pp = 0; pp < NP; pp++) {
        float r0 = 0.0f, r1 = 0.0f;
        if (wgt[pp].x > 0.0f) r0 = (den[pp].x == 0.0f) ? 0.0f : num[pp].x * __builtin_amdgcn_rcpf(den[pp].x);
        if (wgt[pp].y > 0.0f) r1 = (den[pp].y == 0.0f) ? 0.0f : num[pp].y * __builtin_amdgcn_rcpf(den[pp].y);
        if (xb + 2 * pp >= 0 && xb + 2 * pp < a.width) op[2 * pp] = r0;
        if (xb + 2 * pp + 1 < a.width) op[2 * pp + 1] = r1;
        KDE_STAGE(if (a.stage_avg) {
            float* sp = a.stage_avg + frame + (size_t)y * a.width + xb;
            if (xb + 2 * pp >= 0 && xb + 2 * pp < a.width) sp[2 * pp] = c2[pp].x;
            if (xb + 2 * pp + 1 < a.width) sp[2 * pp + 1] = c2[pp].y;
        })
    }
}

template <int WIN, int NP, int BX, int BY, bool CACHE, bool VL, bool ELIDE_ON = true, int CR = (CACHE ? WIN : 0)>
int launch_pk_variant(const JbfLaunch& l, const FastArgs& fa, bool cskip, hipStream_t s)
{
    FastArgs a = fa;
    a.tiles_x = ceil_div(l.width, BX * NP * 2);
    a.tiles_y = ceil_div(l.height, BY);
    const long long blocks = (long long)a.tiles_x * a.tiles_y * l.n;
    a.div_tiles = make_fastdiv24((uint32_t)(a.tiles_x * a.tiles_y), (uint64_t)blocks);
    a.div_tx = make_fastdiv24((uint32_t)a.tiles_x, (uint64_t)a.tiles_x * a.tiles_y);
    if (blocks > 0x7fffffffLL) return fail(KDE_ERR_INVALID, "jbf: batch too large for one launch");
    if (cskip)
        hipLaunchKernelGGL((jbf_pk_kernel<WIN, NP, BX, BY, CACHE, true, VL, ELIDE_ON, CR>), dim3((unsigned)blocks), dim3(BX * BY), 0, s, a);
    else
        hipLaunchKernelGGL((jbf_pk_kernel<WIN, NP, BX, BY, CACHE, false, VL, ELIDE_ON, CR>), dim3((unsigned)blocks), dim3(BX * BY), 0, s, a);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

template <int WIN, int PX, int BX, int BY, bool CACHE>
int launch_variant(const JbfLaunch& l, const FastArgs& fa, bool cskip, hipStream_t s)
{
    FastArgs a = fa;
    a.tiles_x = ceil_div(l.width, BX * PX);
    a.tiles_y = ceil_div(l.height, BY);
    const long long blocks = (long long)a.tiles_x * a.tiles_y * l.n;
    a.div_tiles = make_fastdiv24((uint32_t)(a.tiles_x * a.tiles_y), (uint64_t)blocks);
    a.div_tx = make_fastdiv24((uint32_t)a.tiles_x, (uint64_t)a.tiles_x * a.tiles_y);
    if (blocks > 0x7fffffffLL) return fail(KDE_ERR_INVALID, "jbf: batch too large for one launch");
    if (cskip)
        hipLaunchKernelGGL((jbf_fast_kernel<WIN, PX, BX, BY, CACHE, true>), dim3((unsigned)blocks), dim3(BX * BY), 0, s, a);
    else
        hipLaunchKernelGGL((jbf_fast_kernel<WIN, PX, BX, BY, CACHE, false>), dim3((unsigned)blocks), dim3(BX * BY), 0, s, a);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

struct Variant {
    const char* name;
    int window;
    bool packed;           // packed kernels read the table as (tap of p0, tap of p1) pairs per unit
    int (*launch)(const JbfLaunch&, const FastArgs&, bool, hipStream_t);
};

// name: w<window>-<pk|sc><pixels per thread>-<threads x>x<threads y>-<c = pass-1 arguments cached in registers | r = recomputed>
#define V(WIN, PX, BX, BY, CACHE) \
    {"w" #WIN "-sc" #PX "-" #BX "x" #BY "-" #CACHE, WIN, false, &launch_variant<WIN, PX, BX, BY, CACHE>}
#define K(WIN, NP, BX, BY, CACHE) \
    {"w" #WIN "-pk" #NP "-" #BX "x" #BY "-" #CACHE "-v4", WIN, true, &launch_pk_variant<WIN, NP, BX, BY, CACHE, true>}
#define KS(WIN, NP, BX, BY, CACHE) \
    {"w" #WIN "-pk" #NP "-" #BX "x" #BY "-" #CACHE "-v1", WIN, true, &launch_pk_variant<WIN, NP, BX, BY, CACHE, false>}
#define KN(WIN, NP, BX, BY, CACHE) \
    {"w" #WIN "-pk" #NP "-" #BX "x" #BY "-" #CACHE "-v4-noelide", WIN, true, &launch_pk_variant<WIN, NP, BX, BY, CACHE, true, false>}
// r05 experiment: the first CR window rows keep their pass-1 arguments (VGPRs + AccVGPRs at two waves per SIMD)
#define KC(WIN, NP, BX, BY, CR) \
    {"w" #WIN "-pk" #NP "-" #BX "x" #BY "-keep" #CR "-v4", WIN, true, &launch_pk_variant<WIN, NP, BX, BY, false, true, true, CR>}
const Variant kVariants[] = {
    // the FIRST variant listed for a window is the built-in choice (interleaved A/B sweep on MI355X,
    // profiles/r03_sweep_k1_variants.log); the others stay selectable for the tile sweep of BASELINE config 3.
    // r03: the vector loader ("-v4": 16-byte depth + 12-byte colour loads, LDS rows shifted so that the pass loops keep
    // their 64-bit reads) is the default at every window: 10.11 vs 10.17 ms at window 19, 1.102 vs 1.101 at window 11
    K(5, 1, 16, 16, true),  K(5, 1, 32, 8, true),   KS(5, 1, 16, 16, true),  K(5, 2, 32, 8, true),  K(5, 1, 32, 8, false),
    V(5, 2, 32, 8, true),   V(5, 1, 64, 4, true),
    K(7, 1, 16, 16, true),  K(7, 1, 32, 8, true),   KS(7, 1, 32, 8, true),  K(7, 1, 32, 8, false),  V(7, 1, 64, 4, true),
    // tiles (pixels) of BASELINE config 3's sweep: 64x16 / 32x16 / 64x8 / 128x8 with 4 or 2 pixels per thread (packed pairs),
    // and the scalar kernels at 2 and 1 pixels per thread (32x16, 32x8)
    K(11, 2, 16, 16, false), KS(11, 2, 16, 16, false), KS(11, 1, 16, 16, false), K(11, 1, 16, 16, false), K(11, 1, 32, 8, false),
    KS(11, 1, 32, 8, false), KS(11, 2, 32, 8, false), V(11, 2, 16, 16, false), V(11, 1, 32, 8, false),
    KN(11, 2, 16, 16, false),      // the default without tile-level rule elision: the window's data-independent floor
    K(19, 1, 16, 16, false), KS(19, 1, 16, 16, false), KS(19, 2, 16, 16, false), K(19, 2, 16, 16, false), K(19, 1, 32, 8, false),
    KS(19, 1, 32, 8, false), KS(19, 2, 32, 8, false), V(19, 2, 16, 16, false), V(19, 1, 32, 8, false),
    KN(19, 1, 16, 16, false),
    // r04: the other windows the ABI accepts (kde_jbf_params.window_size is a run-time argument in the reference,
    // JointBilateralFilter.cu:10,18-19).  Up to window 21 the log2(S) table travels in the kernel-argument block; windows 23-31
    // (further down) read it from a device table.  Same kernel template, vector loader; the pass-1 arguments stay in registers at window 3 and are recomputed from window 9 on.
    // First listed = built-in choice (tools/sweep_jbf.py, profiles/r04_sweep_k1_windows.log)
    K(3, 1, 16, 16, true),   K(3, 2, 16, 16, true),
    K(9, 2, 16, 16, false),  K(9, 1, 16, 16, false),
    K(13, 2, 16, 16, false), K(13, 1, 16, 16, false),
    K(15, 1, 16, 16, false), K(15, 2, 16, 16, false),
    K(17, 1, 16, 16, false), K(17, 2, 16, 16, false),
    K(21, 1, 16, 16, false), K(21, 2, 16, 16, false),
    // windows 23..31: the log2(S) table no longer fits the argument block and comes from a device copy (FastArgs::tab_dev)
    K(23, 1, 16, 16, false), K(25, 1, 16, 16, false), K(27, 1, 16, 16, false), K(29, 1, 16, 16, false), K(31, 1, 16, 16, false),
#ifdef KDE_AB_SWITCHES
    // r05: part of the pass-1 arguments kept in registers at two waves per SIMD.  Bit-identical, and
    // SLOWER at windows 11 and 13 (+5 .. +12 %), a wash at window 9 (-0.6 % on 64 x 640x480, +1.6 .. +5 % at 1080p):
    // profiles/r05_sweep_k1_variants.log (tools/ab_k1_keep.py).  The allocator keeps everything in the 256 architectural
    // VGPRs (agpr_count 0: on gfx950 the file is unified, AccVGPRs would only add capacity at ONE wave per SIMD), so what is
    // bought is 8 of 20 issue slots on the kept units and what is paid is the occupancy (2 instead of 4-6 waves per SIMD:
    // the v_exp_f32 / LDS latency is no longer covered) and 2-3 x the code per body.  Measurement build only; K1 is closed.
    KC(9, 1, 16, 16, 9), KC(9, 1, 16, 16, 5), KC(9, 2, 16, 16, 4), KC(11, 1, 16, 16, 5), KC(11, 1, 16, 16, 7), KC(11, 2, 16, 16, 3),
    KC(13, 1, 16, 16, 4), KC(13, 1, 16, 16, 6),
#endif
};
static_assert(sizeof(FastArgs) <= 4096, "FastArgs is passed by value as the kernel-argument block");
#undef V
#undef K
#undef KS
#undef KN
#undef KC
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

}  // namespace

int jbf_fast_variant_count() { return kNumVariants; }
const char* jbf_fast_variant_name(int v) { return (v >= 0 && v < kNumVariants) ? kVariants[v].name : "?"; }
int jbf_fast_variant_window(int v) { return (v >= 0 && v < kNumVariants) ? kVariants[v].window : 0; }

// first listed variant of a window that can serve the launch is the built-in choice
int jbf_fast_default_variant(const JbfLaunch& l)
{
    for (int v = 0; v < kNumVariants; v++)
        if (kVariants[v].window == l.window) return v;
    return -1;
}

bool jbf_fast_supported(const JbfLaunch& l)
{
    if (l.color_sigma == 0.0f || l.depth_sigma == 0.0f) return false;
    const double kc = 1.4426950408889634 / (double)l.color_den;
    if (!(kc * 1.0e38 > 1.0e4) || !(kc < 1.0e30)) return false;   // the invalid-tap bias must drive exp2 to 0
    return jbf_fast_default_variant(l) >= 0;
}

// log2 of the spatial table at the 2^24 scale, in the layout a variant reads: scalar kernels tab[i*W + j]; packed kernels the
// pair (tap of p0, tap of p1) of unit u at tab[(i*W + u)*2 .. +1] (see jbf_pk_kernel).  out: W*W (scalar) / 2*W*W (packed) floats
void jbf_fast_fill_table(int W, const float* table_host, bool packed, float* out)
{
    const int HALF = (W - 1) / 2;
    auto lg = [&](int i, int j) {
        const float sv = table_host[i * W + j];
        return (float)(((sv == 0.0f) ? 0.0 : std::log2((double)sv)) + kScaleLog2);   // S == 0 -> factor skipped
    };
    if (!packed) {
        for (int i = 0; i < W; i++)
            for (int j = 0; j < W; j++) out[i * W + j] = lg(i, j);
        return;
    }
    for (int i = 0; i < W; i++)
        for (int u = 0; u < W; u++) {   // unit -> (tap of p0, tap of p1)
            const int j0 = u <= HALF ? 2 * u : (u < W - 1 ? 2 * (u - HALF) + 1 : 1);
            const int j1 = u <= HALF ? 2 * u : (u < W - 1 ? 2 * (u - HALF) - 1 : W - 2);
            out[(i * W + u) * 2] = lg(i, j0);
            out[(i * W + u) * 2 + 1] = lg(i, j1);
        }
}

bool jbf_fast_needs_device_table(int window) { return window > 21; }

int launch_jbf_fast(const JbfLaunch& l, int variant, const float* table_host, hipStream_t s)
{
    if (variant < 0 || variant >= kNumVariants || kVariants[variant].window != l.window)
        return fail(KDE_ERR_INVALID, "jbf: variant %d does not implement window %d", variant, l.window);
    FastArgs a;
    memset(&a, 0, sizeof(a));
    a.depth = l.depth;
    a.guide = l.guide;
    a.out = l.out;
    a.width = l.width;
    a.height = l.height;
    a.n = l.n;
    const double log2e = 1.4426950408889634;
    a.kc = (float)(log2e / (double)l.color_den);
    a.sd = (float)std::sqrt(log2e / (double)l.depth_den);
    a.t_skip = std::isinf(l.d2_skip) ? INFINITY : (float)(std::sqrt((double)l.d2_skip) * std::sqrt(log2e / (double)l.depth_den));
    a.t2_skip = std::isinf(a.t_skip) ? 3.0e38f : a.t_skip * a.t_skip;
    a.cd_skip = l.cd_skip;
    a.vec4 = (l.width % 4 == 0) && ((reinterpret_cast<uintptr_t>(l.depth) & 15u) == 0) &&
             ((reinterpret_cast<uintptr_t>(l.guide) & 3u) == 0);
    KDE_STAGE(a.stage_avg = g_stage.jbf_avg; a.stage_counters = g_stage.counters; a.stage_force = g_stage.force_full_rules;)
    if (jbf_fast_needs_device_table(l.window)) {
        if (!l.log2_pk_dev) return fail(KDE_ERR_INVALID, "jbf: window %d needs the handle's device copy of the log2 table", l.window);
        a.tab_dev = l.log2_pk_dev;
    } else {
        jbf_fast_fill_table(l.window, table_host, kVariants[variant].packed, a.tab);
    }
    const bool cskip = l.cd_skip <= 195075;
    return kVariants[variant].launch(l, a, cskip, s);
}

}  // namespace kde
