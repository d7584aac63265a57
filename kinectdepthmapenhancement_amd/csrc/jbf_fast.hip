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
//   * taps that are invalid (depth <= 50 mm or outside the image) carry a bias of 0xFF000000 (-1.7e38),
//     which drives the argument to -1.7e38*kc and the weight to exactly 0 with no branch or select;
//   * the reference's "a factor that underflowed to exactly 0 is not multiplied in" rule (Q1) is decided
//     on the ARGUMENT (integer colour distance / |d_q - avg|) against host-computed thresholds, so it does
//     not depend on how small numbers are rounded by the exp hardware;
//   * each thread owns PX horizontally adjacent pixels and walks the window row by row from registers.
// Depth + guide + |b|^2 tiles (with halo) are staged once per workgroup in LDS; the workgroup->tile map
// keeps each XCD on a contiguous band of tiles so halos are shared in that XCD's L2.
#include "kde_internal.h"

namespace kde {
namespace {

constexpr uint32_t kMagic = 0x4B000000u;      // float 2^23
constexpr uint32_t kOff = 1u << 18;           // keeps 2 a.b - |b|^2 + kOff positive (|b|^2 <= 195075 < 2^18)
constexpr uint32_t kInvalidBias = 0xFF000000u; // as a float: -1.7e38; as an unsigned int: above every valid code
constexpr float kBiasF = 8388608.0f + 262144.0f;

struct FastArgs {
    const float* depth;
    const uint8_t* guide;
    float* out;
    int width, height, n;
    int tiles_x, tiles_y;
    float kc;        // log2(e) / (2 sigma_c^2)
    float sd;        // sqrt(log2(e) / (2 sigma_d^2))
    float t_skip;    // depth factor skipped when |d_q - avg| * sd >= t_skip
    int cd_skip;     // colour factor skipped when cd >= cd_skip
    float ls[361];   // log2 of the spatial table (zeros of the table -> factor skipped -> log2 = 0)
    float lsx[31];   // separable form, log2 domain: ls[i][j] = lsx[i] + lsx[j] (only when no table entry is 0)
    float sy[31];    // 2^lsx[i]: per-row linear factor applied to the row's partial sums
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
    const unsigned frame_i = id / tiles;
    const unsigned tile = id - frame_i * tiles;
    const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
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
            const float ls = a.ls[i * WIN + j];
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
            const float ls = CACHE ? 0.0f : a.ls[i * WIN + j];
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
// The spatial weight is applied separably: lsx[j] rides in the per-unit constant of the argument fma and
// the row factor sy[i] scales the row's partial sums once per row, so no per-tap spatial operation is left.
// Per tap: pass 1 = v_dot4 + v_lshl_add + v_exp (8 cycles) + 2 packed ops (4 per unit: add, fma, fma, add);
// pass 2 = v_cmp + v_cndmask + v_exp + 2.5 packed ops (5 per unit: add, mul, fma, fma, add).
// ---------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 bcast(float v) { return f2{v, v}; }

template <int WIN, int NP, int BX, int BY, bool CACHE, bool CSKIP>
__global__ __launch_bounds__(BX* BY) void jbf_pk_kernel(const FastArgs a)
{
    constexpr int R = WIN / 2;
    constexpr int PX = 2 * NP;
    constexpr int NT = BX * BY;
    constexpr int TW = BX * PX, TH = BY;
    constexpr int LW = TW + 2 * R, LH = TH + 2 * R;
    constexpr int P = LW;                 // even: TW and 2R are even
    constexpr int SEGP = NP + R;          // aligned pairs in the row segment of a thread
    constexpr int HALF = (WIN - 1) / 2;
    __shared__ __attribute__((aligned(16))) float s_d[LH * P];
    __shared__ __attribute__((aligned(16))) uint32_t s_c[LH * P];
    __shared__ __attribute__((aligned(16))) uint32_t s_n[LH * P];

    const unsigned nblk = gridDim.x;
    const unsigned lin = blockIdx.x;
    const unsigned per = nblk / 8, rem = nblk % 8;
    const unsigned xcd = lin % 8, slot = lin / 8;
    const unsigned id = xcd * per + (xcd < rem ? xcd : rem) + slot;
    const unsigned tiles = (unsigned)a.tiles_x * a.tiles_y;
    const unsigned frame_i = id / tiles;
    const unsigned tile = id - frame_i * tiles;
    const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
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

    // tap indices (j of p0, j of p1) of unit u
    auto unit_j0 = [](int u) { return u <= HALF ? 2 * u : (u < WIN - 1 ? 2 * (u - HALF) + 1 : 1); };
    auto unit_j1 = [](int u) { return u <= HALF ? 2 * u : (u < WIN - 1 ? 2 * (u - HALF) - 1 : WIN - 2); };

    uint32_t cc[PX], thrU[PX];
    f2 negC[NP];          // -(2^23 + 2^18 + |a|^2): F + negC = -cd exactly (integers < 2^24)
#pragma unroll
    for (int pp = 0; pp < NP; pp++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = 2 * pp + h;
            cc[k] = s_c[(ty + R) * P + tx * PX + R + k];
            const uint32_t na = dot4(cc[k], cc[k], 0);
            negC[pp][h] = -(kBiasF + (float)na);
            thrU[k] = CSKIP ? (kMagic + kOff) + na - (uint32_t)a.cd_skip : 0;
        }
    }

    f2 arg[CACHE ? WIN * WIN * NP : 1];
    f2 wsum[NP], wgt[NP];
#pragma unroll
    for (int pp = 0; pp < NP; pp++) wsum[pp] = wgt[pp] = bcast(0.0f);

    // argument (log2 domain, without the row factor) of S*cf for the two taps of unit u of pair pp
    auto unit_arg = [&](const u2* cp, const u2* np, int pp, int u) -> f2 {
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
        // -cd exactly, then ONE rounding in the fma: a1 = lsx[j] - kc*cd  (log2 of Sx_j * cf)
        const f2 ncd = f2{__uint_as_float(u0), __uint_as_float(u1)} + negC[pp];
        const f2 lsj = f2{a.lsx[unit_j0(u)], a.lsx[unit_j1(u)]};
        f2 a1 = pk_fma(ncd, bcast(a.kc), lsj);
        if (CSKIP) {   // Q1: an underflowed colour factor is skipped (invalid codes compare above the threshold)
            a1.x = (u0 <= thrU[2 * pp]) ? lsj.x : a1.x;
            a1.y = (u1 <= thrU[2 * pp + 1]) ? lsj.y : a1.y;
        }
        return a1;
    };
    auto unit_depth = [&](const f2* dp, int pp, int u) -> f2 {
        if (u <= HALF) return dp[pp + u];
        if (u < WIN - 1) return __builtin_shufflevector(dp[pp + (u - HALF)], dp[pp + (u - HALF)], 1, 0);
        return f2{dp[pp].y, dp[pp + HALF].x};
    };

    auto pass1_row = [&](int i) {
        f2 dp[SEGP];
        u2 cp[SEGP], np[SEGP];
        const int rb = (ty + i) * P + tx * PX;
#pragma unroll
        for (int m = 0; m < SEGP; m++) {
            dp[m] = *reinterpret_cast<const f2*>(&s_d[rb + 2 * m]);
            cp[m] = *reinterpret_cast<const u2*>(&s_c[rb + 2 * m]);
            np[m] = *reinterpret_cast<const u2*>(&s_n[rb + 2 * m]);
        }
        const f2 syi = bcast(a.sy[i]);
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
            f2 rs = bcast(0.0f), rw = bcast(0.0f);
#pragma unroll
            for (int u = 0; u < WIN; u++) {
                const f2 a1 = unit_arg(cp, np, pp, u);
                if (CACHE) arg[(i * NP + pp) * WIN + u] = a1;
                const f2 f = f2{__builtin_amdgcn_exp2f(a1.x), __builtin_amdgcn_exp2f(a1.y)};
                rs = pk_fma(unit_depth(dp, pp, u), f, rs);
                rw = rw + f;
            }
            wsum[pp] = pk_fma(syi, rs, wsum[pp]);
            wgt[pp] = pk_fma(syi, rw, wgt[pp]);
        }
    };
    if (CACHE) {
#pragma unroll
        for (int i = 0; i < WIN; i++) pass1_row(i);
    } else {
#pragma unroll 1
        for (int i = 0; i < WIN; i++) pass1_row(i);
    }

    f2 c2[NP], num[NP], den[NP];
#pragma unroll
    for (int pp = 0; pp < NP; pp++) {
        c2[pp] = wsum[pp] / wgt[pp];             // window averages; wgt == 0 handled at the store
        num[pp] = den[pp] = bcast(0.0f);
    }

    auto pass2_row = [&](int i) {
        f2 dp[SEGP];
        u2 cp[SEGP], np[SEGP];
        const int rb = (ty + i) * P + tx * PX;
#pragma unroll
        for (int m = 0; m < SEGP; m++) {
            dp[m] = *reinterpret_cast<const f2*>(&s_d[rb + 2 * m]);
            if (!CACHE) {
                cp[m] = *reinterpret_cast<const u2*>(&s_c[rb + 2 * m]);
                np[m] = *reinterpret_cast<const u2*>(&s_n[rb + 2 * m]);
            }
        }
        const f2 syi = bcast(a.sy[i]);
#pragma unroll
        for (int pp = 0; pp < NP; pp++) {
            f2 rn = bcast(0.0f), rd = bcast(0.0f);
#pragma unroll
            for (int u = 0; u < WIN; u++) {
                const f2 a1 = CACHE ? arg[(i * NP + pp) * WIN + u] : unit_arg(cp, np, pp, u);
                const f2 dq = unit_depth(dp, pp, u);
                const f2 t = (dq - c2[pp]) * bcast(a.sd);    // subtract first: see the scalar kernel
                f2 a2 = pk_fma(-t, t, a1);
                a2.x = (__builtin_fabsf(t.x) >= a.t_skip) ? a1.x : a2.x;   // Q1: underflowed depth factor skipped
                a2.y = (__builtin_fabsf(t.y) >= a.t_skip) ? a1.y : a2.y;
                const f2 f = f2{__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                rn = pk_fma(dq, f, rn);
                rd = rd + f;
            }
            num[pp] = pk_fma(syi, rn, num[pp]);
            den[pp] = pk_fma(syi, rd, den[pp]);
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
    for (int pp = 0; pp < NP; pp++) {
        float r0 = 0.0f, r1 = 0.0f;
        if (wgt[pp].x > 0.0f) r0 = (den[pp].x == 0.0f) ? 0.0f : num[pp].x / den[pp].x;
        if (wgt[pp].y > 0.0f) r1 = (den[pp].y == 0.0f) ? 0.0f : num[pp].y / den[pp].y;
        if (xb + 2 * pp < a.width) o[2 * pp] = r0;
        if (xb + 2 * pp + 1 < a.width) o[2 * pp + 1] = r1;
    }
}

template <int WIN, int NP, int BX, int BY, bool CACHE>
int launch_pk_variant(const JbfLaunch& l, const FastArgs& fa, bool cskip, hipStream_t s)
{
    FastArgs a = fa;
    a.tiles_x = ceil_div(l.width, BX * NP * 2);
    a.tiles_y = ceil_div(l.height, BY);
    const long long blocks = (long long)a.tiles_x * a.tiles_y * l.n;
    if (blocks > 0x7fffffffLL) return fail(KDE_ERR_INVALID, "jbf: batch too large for one launch");
    if (cskip)
        hipLaunchKernelGGL((jbf_pk_kernel<WIN, NP, BX, BY, CACHE, true>), dim3((unsigned)blocks), dim3(BX * BY), 0, s, a);
    else
        hipLaunchKernelGGL((jbf_pk_kernel<WIN, NP, BX, BY, CACHE, false>), dim3((unsigned)blocks), dim3(BX * BY), 0, s, a);
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
    bool separable_only;   // packed kernels apply the spatial weight separably: no table entry may be 0
    int (*launch)(const JbfLaunch&, const FastArgs&, bool, hipStream_t);
};

// name: w<window>-<pk|sc><pixels per thread>-<threads x>x<threads y>-<c = pass-1 arguments cached in registers | r = recomputed>
#define V(WIN, PX, BX, BY, CACHE) \
    {"w" #WIN "-sc" #PX "-" #BX "x" #BY "-" #CACHE, WIN, false, &launch_variant<WIN, PX, BX, BY, CACHE>}
#define K(WIN, NP, BX, BY, CACHE) \
    {"w" #WIN "-pk" #NP "-" #BX "x" #BY "-" #CACHE, WIN, true, &launch_pk_variant<WIN, NP, BX, BY, CACHE>}
const Variant kVariants[] = {
    K(5, 1, 32, 8, true),   K(5, 2, 32, 8, true),   K(5, 1, 16, 16, true),  K(5, 1, 32, 8, false),
    V(5, 2, 32, 8, true),   V(5, 1, 64, 4, true),
    K(7, 1, 32, 8, true),   K(7, 1, 32, 8, false),  V(7, 1, 64, 4, true),   V(7, 2, 32, 8, false),
    K(11, 1, 32, 8, false), K(11, 2, 32, 8, false), K(11, 1, 16, 16, false), K(11, 2, 16, 16, false),
    V(11, 2, 16, 16, false), V(11, 1, 64, 4, false),
    K(19, 1, 32, 8, false), K(19, 2, 32, 8, false), K(19, 1, 16, 16, false), K(19, 2, 16, 16, false),
    V(19, 2, 16, 16, false), V(19, 1, 64, 4, false),
};
#undef V
#undef K
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

}  // namespace

int jbf_fast_variant_count() { return kNumVariants; }
const char* jbf_fast_variant_name(int v) { return (v >= 0 && v < kNumVariants) ? kVariants[v].name : "?"; }
int jbf_fast_variant_window(int v) { return (v >= 0 && v < kNumVariants) ? kVariants[v].window : 0; }

static bool table_has_zero(const JbfLaunch& l)
{
    for (int i = 0; i < l.window * l.window; i++)
        if (l.table_host[i] == 0.0f) return true;
    return false;
}

// first listed variant of a window that can serve the launch is the built-in choice
int jbf_fast_default_variant(const JbfLaunch& l)
{
    const bool zero = table_has_zero(l);
    for (int v = 0; v < kNumVariants; v++)
        if (kVariants[v].window == l.window && !(zero && kVariants[v].separable_only)) return v;
    return -1;
}

bool jbf_fast_supported(const JbfLaunch& l)
{
    if (l.color_sigma == 0.0f || l.depth_sigma == 0.0f) return false;
    const double kc = 1.4426950408889634 / (double)l.color_den;
    if (!(kc * 1.0e38 > 1.0e4) || !(kc < 1.0e30)) return false;   // the invalid-tap bias must drive exp2 to 0
    return jbf_fast_default_variant(l) >= 0;
}

int launch_jbf_fast(const JbfLaunch& l, int variant, const float* table_host, hipStream_t s)
{
    if (variant < 0 || variant >= kNumVariants || kVariants[variant].window != l.window)
        return fail(KDE_ERR_INVALID, "jbf: variant %d does not implement window %d", variant, l.window);
    if (kVariants[variant].separable_only && table_has_zero(l))
        return fail(KDE_ERR_INVALID, "jbf: variant %s needs a spatial table without underflowed (0) entries", kVariants[variant].name);
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
    a.cd_skip = l.cd_skip;
    for (int i = 0; i < l.window * l.window; i++) {
        const float sv = table_host[i];
        a.ls[i] = (sv == 0.0f) ? 0.0f : (float)std::log2((double)sv);   // S == 0 -> factor skipped
    }
    // separable form of the spatial table: S[i][j] = exp(-((j-r)^2 + (i-r)^2) / (2 sigma^2))
    for (int j = 0; j < l.window; j++) {
        const double dj = (double)(j - l.window / 2);
        const double lg = -(dj * dj) / (2.0 * (double)l.spatial_sigma * (double)l.spatial_sigma) * log2e;
        a.lsx[j] = (float)lg;
        a.sy[j] = (float)std::exp2(lg);
    }
    const bool cskip = l.cd_skip <= 195075;
    return kVariants[variant].launch(l, a, cskip, s);
}

}  // namespace kde
