// jbf_kernels.hip — K0 (colour pre-smoothing), K1 (joint bilateral depth filter) and the sibling
// MRF filter for gfx950.  Reference: JointBilateralFilter/JointBilateralFilter.cu:4-83, :283-290;
// cv::gpu::bilateralFilter (OpenCV 2.4.3 gpu) at the call site :285; MarkovRandomField.cu:4-40.
//
// Layout: one workgroup = one output tile of one frame (blockIdx.z = frame); depth and guide
// tiles (+halo) are staged once in LDS, out-of-image and invalid (<= 50 mm) taps are stored as
// depth 0 so the inner loops carry no bounds tests.  No MFMA: this is a stencil.
#include "kde_internal.h"
#include "kde_device_math.h"

namespace kde {
namespace {

constexpr int kTileX = 32;
constexpr int kTileY = 8;
constexpr int kThreads = kTileX * kTileY;

__device__ __forceinline__ uint32_t load_bgrx(const uint8_t* __restrict__ img, size_t pix)
{
    const uint8_t* p = img + pix * 3;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

__device__ __forceinline__ int color_dist2(uint32_t a, uint32_t b)
{
    int d0 = (int)(a & 0xffu) - (int)(b & 0xffu);
    int d1 = (int)((a >> 8) & 0xffu) - (int)((b >> 8) & 0xffu);
    int d2 = (int)((a >> 16) & 0xffu) - (int)((b >> 16) & 0xffu);
    return d0 * d0 + d1 * d1 + d2 * d2;
}

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

// --------------------------------------------------------------------------------------------
// K1, generic form: any odd window <= 31, one pixel per thread, literal arithmetic
// (division + expf per tap).  This is the reference-shaped fallback; the tuned variants for the
// common windows live below it.
// --------------------------------------------------------------------------------------------
struct JbfDev {
    const float* depth;
    const uint8_t* guide;
    float* out;
    const float* s_eff;
    int width, height, window;
    float color_den, depth_den;   // 2*sigma^2
    int color_on, depth_on;
    int cd_skip;
    float d2_skip;
    KDE_STAGE(float* stage_avg;)      // tools/hooks/libkde_hip_stage.so only
};

__global__ __launch_bounds__(kThreads) void jbf_generic_kernel(JbfDev a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = a.window / 2;
    const int LW = kTileX + 2 * R, LH = kTileY + 2 * R;
    float* sd = reinterpret_cast<float*>(smem);
    uint32_t* sc = reinterpret_cast<uint32_t*>(sd + LW * LH);
    float* ss = reinterpret_cast<float*>(sc + LW * LH);

    const size_t frame = (size_t)blockIdx.z * a.width * a.height;
    const float* __restrict__ depth = a.depth + frame;
    const uint8_t* __restrict__ guide = a.guide + frame * 3;
    const int x0 = blockIdx.x * kTileX, y0 = blockIdx.y * kTileY;
    const int tid = threadIdx.x;

    for (int i = tid; i < LW * LH; i += kThreads) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float d = 0.0f;
        uint32_t c = 0;
        if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
            const size_t q = (size_t)gy * a.width + gx;
            d = depth[q];
            if (!(d > 50.0f)) d = 0.0f;
            c = load_bgrx(guide, q);
        }
        sd[i] = d;
        sc[i] = c;
    }
    for (int i = tid; i < a.window * a.window; i += kThreads) ss[i] = a.s_eff[i];
    __syncthreads();

    const int tx = tid & (kTileX - 1), ty = tid / kTileX;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= a.width || y >= a.height) return;

    const uint32_t cc = sc[(ty + R) * LW + tx + R];
    float w_average = 0.0f, weight = 0.0f;
    for (int i = 0; i < a.window; i++) {
        for (int j = 0; j < a.window; j++) {
            const int li = (ty + i) * LW + tx + j;
            const float dq = sd[li];
            if (dq > 50.0f) {
                const int cd = color_dist2(cc, sc[li]);
                float filter = ss[i * a.window + j];
                if (a.color_on && cd < a.cd_skip) filter *= exp_denormal(-(float)cd / a.color_den);
                w_average += dq * filter;
                weight += filter;
            }
        }
    }
    float result = 0.0f;
    if (weight > 0.0f) {
        w_average /= weight;
        float numerator = 0.0f, denominator = 0.0f;
        for (int i = 0; i < a.window; i++) {
            for (int j = 0; j < a.window; j++) {
                const int li = (ty + i) * LW + tx + j;
                const float dq = sd[li];
                if (dq > 50.0f) {
                    const int cd = color_dist2(cc, sc[li]);
                    float filter = ss[i * a.window + j];
                    if (a.color_on && cd < a.cd_skip) filter *= exp_denormal(-(float)cd / a.color_den);
                    const float dd = dq - w_average;
                    const float d2 = dd * dd;
                    // NaN d2 fails the '<' test on purpose: the reference multiplies the NaN in
                    if (a.depth_on && !(d2 >= a.d2_skip)) filter *= exp_denormal(-d2 / a.depth_den);
                    numerator += dq * filter;
                    denominator += filter;
                }
            }
        }
        result = (denominator == 0.0f) ? 0.0f : numerator / denominator;
    }
    a.out[frame + (size_t)y * a.width + x] = result;
    KDE_STAGE(if (a.stage_avg) a.stage_avg[frame + (size_t)y * a.width + x] = weight > 0.0f ? w_average : __builtin_nanf("");)
}

// --------------------------------------------------------------------------------------------
// K0 — cv::gpu::bilateralFilter on packed 8UC3 (OpenCV 2.4.3 gpu; call site JointBilateralFilter.cu:285).
// The weight exp(space2*ss + n1^2*sc) depends only on (space2, n1) with n1 = L1 colour distance in
// [0,765]; the host tabulates it with the same float expression the CPU restatement uses, and the
// accumulation order / fma / IEEE division are the same, so the u8 result is bit-identical by construction.
//
// Persistent workgroups: the 15 KB table is staged in LDS ONCE per workgroup, which then walks 32x16
// tiles (PX = 2 horizontally adjacent pixels per thread; n1 by one v_sad_u8, channels by v_cvt_f32_ubyteN).
// Reflect-101 borders are resolved while staging the tile, so the tap loop has no bounds tests.
// --------------------------------------------------------------------------------------------
struct PreDev {
    const uint8_t* src;
    uint8_t* dst;
    const float* lut;
    int width, height, n;
    int tiles_x, tiles_y;
    int band_walk;          // tiles are walked in XCD bands (row-pair kernel)
    FastDiv24 div_tpf, div_tx;      // tile index -> (frame, tile of the frame) -> (tile row, tile column) by multiplication
};

constexpr int kPreBX = 16, kPreBY = 16;

// D = old with byte `byte` replaced by saturate_u8(round-half-even(v))
__device__ __forceinline__ uint32_t cvt_pk_u8(float v, uint32_t byte, uint32_t old)
{
    uint32_t r;
    asm("v_cvt_pk_u8_f32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(byte), "v"(old));
    return r;
}
typedef float pre_f2 __attribute__((ext_vector_type(2)));
constexpr int kPreTH = kPreBY;

// PX = horizontally adjacent pixels per thread: 4 amortises the byte -> float conversions best, 2 halves the
// registers (7 instead of 4 waves per SIMD) -- the kernel is latency-bound on its LDS table gathers
#ifndef KDE_K0_RP_WAVES
#define KDE_K0_RP_WAVES 7      // waves per SIMD the allocator must leave room for (radii 1, 2): 72 VGPRs, measured 0.103 vs 0.127 (no cap, 78 VGPRs) / 0.114 (8) ms on 64 x VGA
#endif
template <int R, int PX>
__global__ __launch_bounds__(kPreBX* kPreBY, (R <= 2 ? KDE_K0_RP_WAVES : 0)) void presmooth_kernel(PreDev a)
{
    constexpr int kPrePX = PX, kPreTW = kPreBX * PX;
    constexpr int NT = kPreBX * kPreBY;
    constexpr int LW = kPreTW + 2 * R, LH = kPreTH + 2 * R;
    constexpr int LUT_N = (R * R + 1) * 766;
    constexpr int SEG = kPrePX + 2 * R;
    __shared__ float lut[LUT_N];
    // rows are read as 16-byte vectors (4 pixels): LW * 4 bytes and the per-thread offset are multiples of 16, and
    // consecutive lanes read consecutive vectors, which is bank-conflict free (dword reads at stride 4 are 8-way)
    constexpr int LP = (LW + 3) / 4 * 4;                 // LDS row pitch
    __shared__ __attribute__((aligned(16))) uint32_t sc[LH * LP + 4];

    const int tid = threadIdx.x;
    for (int i = tid; i < LUT_N; i += NT) lut[i] = a.lut[i];
    // weight of the centre tap (exp(0) = 1 unless the sigmas are degenerate): wave-uniform, kept in a scalar register
    const float w_centre = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(a.lut[0])));

    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const int total = tiles_per_frame * a.n;
    const int tx = tid % kPreBX, ty = tid / kPreBX;

    // Staging slots of this thread (tile-independent LDS coordinates), and a register prefetch of the NEXT tile:
    // its global loads are issued before the current tile is computed and are only waited for when they are
    // written to LDS, so their latency hides behind the tap loop.
    // A thread stages column sx of rows sy0, sy0 + RPP, sy0 + 2 RPP, ... of the tile + halo (RPP rows per pass, LW threads per
    // row; NT - RPP * LW threads idle): slot k then differs from slot 0 by a wave-uniform amount -- k * RPP rows -- in the
    // image (folded into the scalar base address) and in LDS (an immediate offset), so a thread keeps ONE byte offset and ONE
    // LDS address for all its slots, and interior tiles stage with no per-slot address arithmetic at all.
    constexpr int RPP = NT / LW;
    constexpr int NSLOT = (LH + RPP - 1) / RPP;
    const int sy0 = tid / LW, sx = tid - sy0 * LW;
    const bool stager = sy0 < RPP;
    const uint32_t lds0 = (uint32_t)(sy0 * LP + sx);
    const uint32_t rel0 = ((uint32_t)sy0 * (uint32_t)a.width + (uint32_t)sx) * 3u;
    const uint32_t out_rel = ((uint32_t)ty * (uint32_t)a.width + (uint32_t)(tx * kPrePX)) * 3u;
    const size_t last_pix = (size_t)a.n * a.width * a.height - 1;     // its 4-byte read would leave the buffer
    // Tile walk: linear index L (this workgroup's k-th tile is blockIdx.x + k * gridDim.x) -> tile xcd_band_id(L, total), so
    // that each XCD works through a contiguous band of the tile list and horizontally adjacent tiles -- which share their
    // halo columns and, being only 96 bytes wide, their 128-byte lines -- meet in ONE L2 (KDE_K0_BAND_WALK, see launch_presmooth)
    auto tile_of = [&](int L) { return a.band_walk ? (int)xcd_band_id((unsigned)L, (unsigned)total) : L; };
    // band_walk == 2 ("run walk", r05): XCD bands as above, but a workgroup takes RUNS of kRun horizontally adjacent tiles one
    // after the other instead of every (workgroups per XCD)-th tile.  Four 32-pixel tiles are 384 bytes = three 128-byte lines
    // of a row: inside a run every line a tile shares with its neighbour is re-read by the SAME workgroup a moment later (an L2
    // hit, never two workgroups queueing on one line at the same time, which is what made the plain band walk slower on large
    // batches), and concurrently running workgroups of an XCD only share the one halo line between two runs.
    // -> this workgroup's it-th tile, or -1 when it has none left (all scalars)
    constexpr int kRun = 4;
    // the band of this workgroup's XCD, the workgroups that share it, and how many FULL runs each of them gets: what is left
    // after those (fewer than kRun x workgroups tiles) is dealt tile by tile, so that no workgroup ends up a whole run
    // behind the others (64 x 640x480: 1200 runs over 224 workgroups per XCD would be 6 runs for some and 5 for the rest --
    // 24 against 21.4 tiles on average; with the remainder dealt singly it is 22)
    const unsigned w_xcd = blockIdx.x % 8u, w_slot = blockIdx.x / 8u;
    const unsigned w_per = (unsigned)total / 8u, w_rem = (unsigned)total % 8u;
    const unsigned w_start = w_xcd * w_per + (w_xcd < w_rem ? w_xcd : w_rem), w_len = w_per + (w_xcd < w_rem ? 1u : 0u);
    const unsigned w_wgs = (gridDim.x + 7u - w_xcd) / 8u;                 // workgroups that run on this XCD
    const unsigned w_full = a.band_walk == 2 ? w_len / ((unsigned)kRun * w_wgs) : 0u;
    auto tile_index = [&](int it) -> int {
        if (a.band_walk != 2) {
            const long long L = (long long)blockIdx.x + (long long)it * gridDim.x;
            return L < total ? tile_of((int)L) : -1;
        }
        unsigned pos;
        if ((unsigned)it < w_full * kRun) pos = (unsigned)kRun * (w_slot + ((unsigned)it / kRun) * w_wgs) + (unsigned)it % kRun;
        else pos = w_full * kRun * w_wgs + w_slot + ((unsigned)it - w_full * kRun) * w_wgs;
        return pos < w_len ? (int)(w_start + pos) : -1;
    };
    // (frame, tile row, tile column) of linear tile index L: computed ONCE per tile (when it is prefetched) and carried to
    // the iteration that computes it; the two divisions are multiplications (fastdiv24, exact for < 2^24 tiles)
    struct TileAt { int frame_i, tyi, txi; };
    auto tile_at = [&](int tile_id) {
        const uint32_t t = (uint32_t)tile_id;
        const uint32_t f = fastdiv24(t, a.div_tpf);
        const uint32_t tile = t - f * (uint32_t)tiles_per_frame;
        const uint32_t ty_ = fastdiv24(tile, a.div_tx);
        return TileAt{(int)f, (int)ty_, (int)(tile - ty_ * (uint32_t)a.tiles_x)};
    };
    auto fetch = [&](const TileAt& at, uint32_t* pre) {
        const int frame_i = at.frame_i, tyi = at.tyi, txi = at.txi;
        const size_t frame = (size_t)frame_i * a.width * a.height;
        // a tile whose halo stays inside the image needs no border reflection (workgroup-uniform); the one tile whose halo
        // ends on the last pixel of the whole batch takes the general path (the 4-byte read of that pixel is split there)
        const bool ends_batch = frame_i == a.n - 1 && txi * kPreTW + kPreTW + R == a.width && tyi * kPreTH + kPreTH + R == a.height;
        const bool inner = txi * kPreTW - R >= 0 && txi * kPreTW + kPreTW + R <= a.width && tyi * kPreTH - R >= 0 &&
                           tyi * kPreTH + kPreTH + R <= a.height && !ends_batch;
        if (inner) {
            const uint8_t* base = a.src + (frame + (size_t)(tyi * kPreTH - R) * a.width + (size_t)(txi * kPreTW - R)) * 3;
#pragma unroll
            for (int k = 0; k < NSLOT; k++) {
                if (stager && sy0 + k * RPP < LH)        // one (unaligned) global_load_dword: scalar base + 32-bit offset
                    __builtin_memcpy(&pre[k], base + (size_t)(k * RPP) * a.width * 3 + rel0, 4);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            if (stager && sy0 + k * RPP < LH) {
                const int gx = reflect101(txi * kPreTW + sx - R, a.width);
                const int gy = reflect101(tyi * kPreTH + sy0 + k * RPP - R, a.height);
                const size_t pix = frame + (size_t)gy * a.width + gx;
                if (pix < last_pix) __builtin_memcpy(&pre[k], a.src + pix * 3, 4);
                else pre[k] = load_bgrx(a.src, pix);
            }
        }
    };

    uint32_t pre[NSLOT];
    TileAt cur{0, 0, 0};
    int cur_id = tile_index(0);
    if (cur_id >= 0) {
        cur = tile_at(cur_id);
        fetch(cur, pre);
    }
    for (int it = 0; cur_id >= 0; it++) {
        const int frame_i = cur.frame_i;
        const int x0 = cur.txi * kPreTW, y0 = cur.tyi * kPreTH;
        const size_t frame = (size_t)frame_i * a.width * a.height;

        __syncthreads();   // previous tile fully consumed (also orders the LUT staging before first use)
#pragma unroll
        for (int k = 0; k < NSLOT; k++)
            if (stager && sy0 + k * RPP < LH) sc[lds0 + k * RPP * LP] = pre[k] & 0x00ffffffu;     // the 4th byte belongs to the next pixel
        __syncthreads();
        cur_id = tile_index(it + 1);
        if (cur_id >= 0) {
            cur = tile_at(cur_id);
            fetch(cur, pre);
        }

        // a tile that lies fully inside the image (workgroup-uniform) needs no per-pixel bounds tests
        const bool full = x0 + kPreTW <= a.width && y0 + kPreTH <= a.height;
        const int xb = x0 + tx * kPrePX, y = y0 + ty;
        if (!full && (xb >= a.width || y >= a.height)) continue;

        // (b, g) accumulate as one packed pair (v_pk_fma_f32, same roundings as two fma).  Measured and rejected (the kernel
        // lives on its 7 resident waves, every extra live register spills): packing (r, weight sum) as well (8 more VGPRs for
        // the {r, 1.0} operands), the weight sums of the two pixels as one pair (the table reads only land in adjacent
        // registers through moves or spills: 0.142 ms), re-using the weight of the mutual tap of the pair (0.105-0.111 ms),
        // converting the six output bytes straight into a dword + a halfword (unaligned dword stores: 0.099 vs 0.098 ms)
        uint32_t cc[kPrePX];
        pre_f2 sbg[kPrePX];
        float sr[kPrePX], sden[kPrePX];
#pragma unroll
        for (int k = 0; k < kPrePX; k++) {
            cc[k] = sc[(ty + R) * LP + tx * kPrePX + R + k];
            sbg[k] = pre_f2{0.0f, 0.0f};
            sr[k] = sden[k] = 0.0f;
        }
#pragma unroll
        for (int dy = -R; dy <= R; dy++) {
            uint32_t v[(SEG + 3) / 4 * 4];
            pre_f2 fbg[SEG];
            float fr[SEG];
            if (PX == 4) {
#pragma unroll
                for (int q4 = 0; q4 < (SEG + 3) / 4; q4++) {
                    const uint4 w = *reinterpret_cast<const uint4*>(&sc[(ty + R + dy) * LP + tx * kPrePX + 4 * q4]);
                    v[4 * q4] = w.x; v[4 * q4 + 1] = w.y; v[4 * q4 + 2] = w.z; v[4 * q4 + 3] = w.w;
                }
            } else {
#pragma unroll
                for (int q2 = 0; q2 < (SEG + 1) / 2; q2++) {
                    const uint2 w = *reinterpret_cast<const uint2*>(&sc[(ty + R + dy) * LP + tx * kPrePX + 2 * q2]);
                    v[2 * q2] = w.x; v[2 * q2 + 1] = w.y;
                }
            }
#pragma unroll
            for (int q = 0; q < SEG; q++) {
                fbg[q] = pre_f2{(float)(v[q] & 0xffu), (float)((v[q] >> 8) & 0xffu)};
                fr[q] = (float)((v[q] >> 16) & 0xffu);
            }
#pragma unroll
            for (int dx = -R; dx <= R; dx++) {
                const int space2 = dx * dx + dy * dy;
                if (space2 > R * R) continue;        // same tap order as the reference kernel: cy outer, cx inner
#pragma unroll
                for (int k = 0; k < kPrePX; k++) {
                    const int q = k + R + dx;
                    float w;
                    if (space2 == 0) {
                        w = w_centre;                // the pixel itself: distance 0, n1 = 0 -- one table entry, no gather
                    } else {
                        const uint32_t n1 = __builtin_amdgcn_sad_u8(v[q], cc[k], 0u);   // |db| + |dg| + |dr|
                        w = lut[space2 * 766 + n1];
                    }
                    sbg[k] = __builtin_elementwise_fma(pre_f2{w, w}, fbg[q], sbg[k]);
                    sr[k] = __builtin_fmaf(w, fr[q], sr[k]);
                    sden[k] += w;
                }
            }
        }
        // The reference divides (IEEE) and rounds half-to-even to u8.  num * rcp(den) is within 1.5 ulp (< 2.3e-5 below
        // 256) of the exact quotient and the IEEE quotient within 0.5 ulp, so both round to the same integer unless the
        // product sits within 1e-4 of k + 0.5; those (rare) lanes take the IEEE division, so the byte is the reference's.
        uint32_t px[kPrePX];
#pragma unroll
        for (int k = 0; k < kPrePX; k++) {
            const float den = sden[k];
            const float r = __builtin_amdgcn_rcpf(den);
            const pre_f2 qbg = sbg[k] * pre_f2{r, r};
            float q[3] = {qbg.x, qbg.y, sr[k] * r};
            const pre_f2 ebg = pre_f2{__builtin_amdgcn_fractf(q[0]), __builtin_amdgcn_fractf(q[1])} - pre_f2{0.5f, 0.5f};
            const float er = __builtin_amdgcn_fractf(q[2]) - 0.5f;
            // distance of the nearest channel to its rounding boundary (one v_min3_f32 with |.| modifiers); a weight sum
            // that is not a positive normal number (NaN / infinite table entries from degenerate sigmas) -> exact path
            const float emin = __builtin_fminf(__builtin_fminf(__builtin_fabsf(ebg.x), __builtin_fabsf(ebg.y)), __builtin_fabsf(er));
            const bool ambiguous = !(emin > 1.0e-4f) || !__builtin_amdgcn_class(den, 0x100);
            if (ambiguous) {
                q[0] = sbg[k].x / den;
                q[1] = sbg[k].y / den;
                q[2] = sr[k] / den;
            }
            // saturate_cast<uchar>: v_cvt_pk_u8_f32 rounds to nearest even, clamps to [0, 255] and packs the byte
            px[k] = cvt_pk_u8(q[2], 2u, cvt_pk_u8(q[1], 1u, cvt_pk_u8(q[0], 0u, 0u)));
        }
        // wave-uniform tile origin + this thread's 32-bit offset inside the tile (out_rel, formed once per kernel)
        uint8_t* o = a.dst + (frame + (size_t)y0 * a.width + x0) * 3 + out_rel;
        if (PX == 4 && (a.width & 3) == 0 && xb + 3 < a.width && (reinterpret_cast<uintptr_t>(a.dst) & 3u) == 0) {
            uint32_t* ow = reinterpret_cast<uint32_t*>(o);                  // 4 pixels = 12 bytes = 3 dwords
            ow[0] = px[0] | (px[1] << 24);
            ow[1] = (px[1] >> 8) | (px[2] << 16);
            ow[2] = (px[2] >> 16) | (px[3 % PX] << 8);
        } else if (PX == 2 && (full || xb + 1 < a.width)) {
            // 2 pixels = 6 bytes = 3 halfword stores (global stores take any alignment: odd widths / odd base addresses)
            const uint16_t h0 = (uint16_t)(px[0] & 0xffffu), h1 = (uint16_t)((px[0] >> 16) | ((px[1] & 0xffu) << 8)),
                           h2 = (uint16_t)(px[1] >> 8);
            __builtin_memcpy(o, &h0, 2);
            __builtin_memcpy(o + 2, &h1, 2);
            __builtin_memcpy(o + 4, &h2, 2);
        } else {
#pragma unroll
            for (int k = 0; k < kPrePX; k++)
                if (xb + k < a.width) {
                    o[3 * k] = (uint8_t)(px[k] & 0xff);
                    o[3 * k + 1] = (uint8_t)((px[k] >> 8) & 0xff);
                    o[3 * k + 2] = (uint8_t)((px[k] >> 16) & 0xff);
                }
        }
    }
}

#ifdef KDE_AB_SWITCHES      // measured slower than the row-pair kernel: compiled for tools/bench_k0.py only (tools/hooks/libkde_hip_ab.so)
// --------------------------------------------------------------------------------------------
// K0, 2 x 2 pixels per thread (r03; opt-in, KDE_K0_2X2=1: measured slower than the row-pair kernel above).  The kernel above is bound by the VALU instructions
// it issues (184 per pixel at radius 2: 45 byte -> float conversions, 65 for the 13 taps, ~50 for the quotient /
// rounding / packing), not by memory.  A thread that owns a 2 x 2 block loads each of its 2R + 2 window rows ONCE for
// both output rows: 27 instead of 45 conversions per pixel; (r, weight) accumulate as one packed pair like (b, g):
// 4 instead of 5 instructions per tap; the epilogue tests "is the quotient within 1e-4 of a rounding boundary" on its
// fractional part and converts with v_cvt_pk_u8_f32 (round-to-nearest-even, saturating, packing: the reference's
// saturate_cast<uchar>).  Same tap order per pixel (dy outer, dx inner), same fma / division semantics: same bytes.
// --------------------------------------------------------------------------------------------
constexpr int kPre22TW = kPreBX * 2, kPre22TH = kPreBY * 2;       // 32 x 32 pixels per workgroup
#ifndef KDE_K0_WAVES
#define KDE_K0_WAVES 5
#endif
constexpr int kPre22Waves = KDE_K0_WAVES;       // waves per SIMD the register allocator must leave room for (latency-bound on LDS gathers)

template <int R, int WAVES>
__global__ __launch_bounds__(kPreBX* kPreBY, WAVES) void presmooth22_kernel(PreDev a)
{
    constexpr int NT = kPreBX * kPreBY;
    constexpr int LW = kPre22TW + 2 * R, LH = kPre22TH + 2 * R;
    constexpr int LUT_N = (R * R + 1) * 766;
    constexpr int SEG = 2 + 2 * R;                       // window columns of a pixel pair (even)
    constexpr int ROWS = 2 + 2 * R;                      // window rows of the two output rows
    constexpr int LP = (LW + 1) / 2 * 2;                 // LDS row pitch (even: a pair's segment starts 8-byte aligned)
    __shared__ float lut[LUT_N];
    __shared__ __attribute__((aligned(16))) uint32_t sc[LH * LP + 4];

    const int tid = threadIdx.x;
    for (int i = tid; i < LUT_N; i += NT) lut[i] = a.lut[i];

    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const int total = tiles_per_frame * a.n;
    const int tx = tid % kPreBX, ty = tid / kPreBX;

    constexpr int NSLOT = (LW * LH + NT - 1) / NT;
    // staging slot k of this thread: element tid + k NT of the (LW x LH) tile; its coordinates are recomputed where they
    // are used (a multiply-shift each) instead of living in 2 NSLOT registers across the tap loop
    auto slot_y = [&](int k) { return (int)(((unsigned)(tid + k * NT) * ((1u << 20) / LW + 1u)) >> 20); };     // (tid + k NT) / LW, exact for < 4096
    auto slot_x = [&](int k) { return tid + k * NT - slot_y(k) * LW; };
    static_assert(LW * LH < 4096, "slot division");
    const size_t last_pix = (size_t)a.n * a.width * a.height - 1;     // its 4-byte read would leave the buffer
    auto fetch = [&](int t, uint32_t* pre) {
        const int frame_i = t / tiles_per_frame;
        const int tile = t - frame_i * tiles_per_frame;
        const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
        const size_t frame = (size_t)frame_i * a.width * a.height;
        const bool inner = txi * kPre22TW - R >= 0 && txi * kPre22TW + kPre22TW + R <= a.width && tyi * kPre22TH - R >= 0 &&
                           tyi * kPre22TH + kPre22TH + R <= a.height;
#pragma unroll
        for (int k = 0; k < NSLOT; k++) {
            if (tid + k * NT < LW * LH) {
                const int ux = txi * kPre22TW + slot_x(k) - R, uy = tyi * kPre22TH + slot_y(k) - R;
                const int gx = inner ? ux : reflect101(ux, a.width);
                const int gy = inner ? uy : reflect101(uy, a.height);
                const size_t pix = frame + (size_t)gy * a.width + gx;
                if (pix < last_pix) {
                    uint32_t v;
                    __builtin_memcpy(&v, a.src + pix * 3, 4);          // one (unaligned) global_load_dword
                    pre[k] = v & 0x00ffffffu;
                } else {
                    pre[k] = load_bgrx(a.src, pix);
                }
            }
        }
    };

    uint32_t pre[NSLOT];
    if ((int)blockIdx.x < total) fetch(blockIdx.x, pre);
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int frame_i = t / tiles_per_frame;
        const int tile = t - frame_i * tiles_per_frame;
        const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
        const int x0 = txi * kPre22TW, y0 = tyi * kPre22TH;
        const size_t frame = (size_t)frame_i * a.width * a.height;

        __syncthreads();   // previous tile fully consumed (also orders the LUT staging before first use)
#pragma unroll
        for (int k = 0; k < NSLOT; k++)
            if (tid + k * NT < LW * LH) sc[slot_y(k) * LP + slot_x(k)] = pre[k];
        __syncthreads();
        if (t + (int)gridDim.x < total) fetch(t + gridDim.x, pre);

        const int xb = x0 + 2 * tx, yb = y0 + 2 * ty;
        if (xb >= a.width || yb >= a.height) continue;

        // [output row][pixel of the pair]: (b, g) and (r, weight sum) as packed pairs
        uint32_t cc[2][2];
        pre_f2 sbg[2][2], srw[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                cc[j][k] = sc[(2 * ty + R + j) * LP + 2 * tx + R + k];
                sbg[j][k] = srw[j][k] = pre_f2{0.0f, 0.0f};
            }
#pragma unroll
        for (int r = 0; r < ROWS; r++) {               // window row r of the block = staged row 2 ty + r
            // one row in flight at a time: without the fence the scheduler hoists the LDS reads and conversions of all
            // 2R + 2 rows to the top (146 VGPRs at radius 2: 3 waves per SIMD, and the kernel lives on resident waves)
            __builtin_amdgcn_sched_barrier(0);
            uint32_t v[SEG];
            pre_f2 fbg[SEG], frw[SEG];
#pragma unroll
            for (int q2 = 0; q2 < SEG / 2; q2++) {
                const uint2 w = *reinterpret_cast<const uint2*>(&sc[(2 * ty + r) * LP + 2 * tx + 2 * q2]);
                v[2 * q2] = w.x;
                v[2 * q2 + 1] = w.y;
            }
#pragma unroll
            for (int q = 0; q < SEG; q++) {
                fbg[q] = pre_f2{(float)(v[q] & 0xffu), (float)((v[q] >> 8) & 0xffu)};
                frw[q] = pre_f2{(float)((v[q] >> 16) & 0xffu), 1.0f};
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {               // output row j sees this row as dy = r - R - j
                const int dy = r - R - j;
                if (dy < -R || dy > R) continue;
#pragma unroll
                for (int dx = -R; dx <= R; dx++) {
                    const int space2 = dx * dx + dy * dy;
                    if (space2 > R * R) continue;        // same tap order as the reference kernel: cy outer, cx inner
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const int q = k + R + dx;
                        const uint32_t n1 = __builtin_amdgcn_sad_u8(v[q], cc[j][k], 0u);   // |db| + |dg| + |dr|
                        const float w = lut[space2 * 766 + n1];
                        const pre_f2 ww = pre_f2{w, w};
                        sbg[j][k] = __builtin_elementwise_fma(ww, fbg[q], sbg[j][k]);
                        srw[j][k] = __builtin_elementwise_fma(ww, frw[q], srw[j][k]);       // (w r + s, w 1 + sum): fma(w, 1, s) == w + s
                    }
                }
            }
        }
        // The reference divides (IEEE) and rounds half-to-even to u8.  rcp + one Newton step is within 1 ulp
        // (< 1.6e-5 below 256) of the quotient, which rounds to the same integer unless the quotient sits within
        // that distance of k + 0.5; those (rare) lanes take the IEEE division, so the byte is the reference's.
        uint32_t px[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const float den = srw[j][k].y;
                const float rc = __builtin_amdgcn_rcpf(den);
                float q[3] = {sbg[j][k].x, sbg[j][k].y, srw[j][k].x};
                bool ambiguous = false;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float num = q[c];
                    const float q0 = num * rc;
                    q[c] = __builtin_fmaf(__builtin_fmaf(-den, q0, num), rc, q0);
                    ambiguous |= !(__builtin_fabsf(__builtin_amdgcn_fractf(q[c]) - 0.5f) > 1.0e-4f);   // NaN -> exact path
                }
                if (ambiguous) {
                    q[0] = sbg[j][k].x / den;
                    q[1] = sbg[j][k].y / den;
                    q[2] = srw[j][k].x / den;
                }
                // saturate_cast<uchar>: v_cvt_pk_u8_f32 rounds to nearest even, clamps to [0, 255] and packs
                px[j][k] = cvt_pk_u8(q[2], 2u, cvt_pk_u8(q[1], 1u, cvt_pk_u8(q[0], 0u, 0u)));
            }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int y = yb + j;
            if (y >= a.height) continue;
            uint8_t* o = a.dst + (frame + (size_t)y * a.width + xb) * 3;
            if (xb + 1 < a.width && (reinterpret_cast<uintptr_t>(o) & 1u) == 0) {
                uint16_t* oh = reinterpret_cast<uint16_t*>(o);                  // 2 pixels = 6 bytes = 3 halfwords
                oh[0] = (uint16_t)(px[j][0] & 0xffffu);
                oh[1] = (uint16_t)((px[j][0] >> 16) | ((px[j][1] & 0xffu) << 8));
                oh[2] = (uint16_t)(px[j][1] >> 8);
            } else {
#pragma unroll
                for (int k = 0; k < 2; k++)
                    if (xb + k < a.width) {
                        o[3 * k] = (uint8_t)(px[j][k] & 0xff);
                        o[3 * k + 1] = (uint8_t)((px[j][k] >> 8) & 0xff);
                        o[3 * k + 2] = (uint8_t)((px[j][k] >> 16) & 0xff);
                    }
            }
        }
    }
}
#endif  // KDE_AB_SWITCHES

// K0 for any radius (OpenCV accepts any kernel size; the reference's call site uses 5): one thread per pixel, taps
// read through the caches with reflect-101 addressing, weights from the same host-computed table (in global memory:
// (radius^2 + 1) x 766 floats do not fit LDS beyond radius 4), same tap order / fma / IEEE division as the oracle.
__global__ __launch_bounds__(kPreBX* kPreBY) void presmooth_generic_kernel(PreDev a, int R)
{
    const int x = blockIdx.x * kPreBX + threadIdx.x % kPreBX, y = blockIdx.y * kPreBY + threadIdx.x / kPreBX;
    if (x >= a.width || y >= a.height) return;
    const size_t frame = (size_t)blockIdx.z * a.width * a.height;
    const uint32_t cc = load_bgrx(a.src, frame + (size_t)y * a.width + x);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, sum2 = 0.0f;
    for (int dy = -R; dy <= R; dy++) {
        const int gy = reflect101(y + dy, a.height);
        for (int dx = -R; dx <= R; dx++) {
            const int space2 = dx * dx + dy * dy;
            if (space2 > R * R) continue;
            const uint32_t v = load_bgrx(a.src, frame + (size_t)gy * a.width + reflect101(x + dx, a.width));
            const uint32_t n1 = __builtin_amdgcn_sad_u8(v, cc, 0u);
            const float w = a.lut[(size_t)space2 * 766 + n1];
            s0 = __builtin_fmaf(w, (float)(v & 0xffu), s0);
            s1 = __builtin_fmaf(w, (float)((v >> 8) & 0xffu), s1);
            s2 = __builtin_fmaf(w, (float)((v >> 16) & 0xffu), s2);
            sum2 += w;
        }
    }
    auto sat = [](float q) -> uint8_t {      // saturate_cast<uchar>: round half to even, clamp
        if (!(q > 0.0f)) return 0;
        if (q >= 255.0f) return 255;
        return (uint8_t)rintf(q);
    };
    uint8_t* o = a.dst + (frame + (size_t)y * a.width + x) * 3;
    o[0] = sat(s0 / sum2);
    o[1] = sat(s1 / sum2);
    o[2] = sat(s2 / sum2);
}

// --------------------------------------------------------------------------------------------
// markov_random_field — MarkovRandomField/MarkovRandomField.cu:4-40
// --------------------------------------------------------------------------------------------
struct MrfDev {
    const float* depth;
    const uint8_t* bgr;
    float* out;
    int width, height, window;
    float color_sigma, smooth_sigma;
};

__global__ __launch_bounds__(kThreads) void mrf_kernel(MrfDev a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = a.window / 2;
    const int LW = kTileX + 2 * R, LH = kTileY + 2 * R;
    float* sd = reinterpret_cast<float*>(smem);
    uint32_t* sc = reinterpret_cast<uint32_t*>(sd + LW * LH);

    const size_t frame = (size_t)blockIdx.z * a.width * a.height;
    const float* __restrict__ depth = a.depth + frame;
    const uint8_t* __restrict__ bgr = a.bgr + frame * 3;
    const int x0 = blockIdx.x * kTileX, y0 = blockIdx.y * kTileY;
    const int tid = threadIdx.x;

    for (int i = tid; i < LW * LH; i += kThreads) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float d = 0.0f;
        uint32_t c = 0;
        if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
            const size_t q = (size_t)gy * a.width + gx;
            d = depth[q];
            if (!(d > 50.0f)) d = 0.0f;
            c = load_bgrx(bgr, q);
        }
        sd[i] = d;
        sc[i] = c;
    }
    __syncthreads();

    const int tx = tid & (kTileX - 1), ty = tid / kTileX;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= a.width || y >= a.height) return;

    const uint32_t cc = sc[(ty + R) * LW + tx + R];
    // the centre enters with weight 1 whatever its value (MarkovRandomField.cu:15)
    float numerator = depth[(size_t)y * a.width + x], denominator = 1.0f;
    for (int i = 0; i < a.window; i++) {
        for (int j = 0; j < a.window; j++) {
            const int li = (ty + i) * LW + tx + j;
            const float dq = sd[li];
            if (dq > 50.0f) {
                const int cd = color_dist2(cc, sc[li]);
                float color_filter = 0.0f;
                if (a.color_sigma != 0.0f) color_filter = exp_denormal(-a.color_sigma * (float)cd);
                float filter = a.smooth_sigma;
                filter *= color_filter;
                numerator += dq * filter;
                denominator += filter;
            }
        }
    }
    a.out[frame + (size_t)y * a.width + x] = (denominator == 0.0f) ? 0.0f : numerator / denominator;
}

}  // namespace

int jbf_variant_count() { return 1 + jbf_fast_variant_count(); }
const char* jbf_variant_name(int v) { return v == 0 ? "generic-32x8-1px" : jbf_fast_variant_name(v - 1); }

// the public variant id launch_jbf() will run for these parameters (0 = the generic kernel)
int jbf_active_variant(const JbfLaunch& a)
{
    if (a.variant > 0) return a.variant;
    if (a.variant < 0 && jbf_fast_supported(a)) return 1 + jbf_fast_default_variant(a);
    return 0;
}

int launch_jbf(const JbfLaunch& a, hipStream_t s)
{
    // variant -1: tuned kernel when one exists for this window and parameter regime, else the generic one
    if (a.variant > 0) {
        if (!jbf_fast_supported(a)) return fail(KDE_ERR_INVALID, "jbf: tuned variants need non-zero sigmas and a built window");
        return launch_jbf_fast(a, a.variant - 1, a.table_host, s);
    }
    if (a.variant < 0 && jbf_fast_supported(a))
        return launch_jbf_fast(a, jbf_fast_default_variant(a), a.table_host, s);
    JbfDev d;
    d.depth = a.depth;
    d.guide = a.guide;
    d.out = a.out;
    d.s_eff = a.s_eff;
    d.width = a.width;
    d.height = a.height;
    d.window = a.window;
    d.color_den = a.color_den;
    d.depth_den = a.depth_den;
    d.color_on = a.color_sigma != 0.0f;
    d.depth_on = a.depth_sigma != 0.0f;
    d.cd_skip = a.cd_skip;
    d.d2_skip = a.d2_skip;
    KDE_STAGE(d.stage_avg = g_stage.jbf_avg;)
    const int R = a.window / 2;
    const size_t lds = (size_t)(kTileX + 2 * R) * (kTileY + 2 * R) * 8 + (size_t)a.window * a.window * 4;
    dim3 grid(ceil_div(a.width, kTileX), ceil_div(a.height, kTileY), a.n);
    hipLaunchKernelGGL(jbf_generic_kernel, grid, dim3(kThreads), lds, s, d);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

// persistent-grid size of the tuned K0 kernels on the CURRENT device: every CU filled to the occupancy the variant
// reaches (LDS table + VGPRs), no second wave.  Asked once per handle (kde_jbf_create) and kept there -- a handle
// belongs to the device it was created on -- so nothing is cached across devices or threads here.
#ifndef KDE_K0_PX
#define KDE_K0_PX 2
#endif
constexpr int kPrePxPerThread = KDE_K0_PX;                  // pixels per thread (A/B: -DKDE_K0_PX=4, tools/ab_k0_build.sh)

// A/B switch of the measurement build (tools/bench_k0.py): KDE_K0_2X2=1 selects the 2 x 2-pixels-per-thread form for radii 1 and 2.  It issues 25 %
// fewer instructions but needs 146 VGPRs (3 waves per SIMD instead of 6) and measured 0.146 vs 0.115 ms on 64 x VGA.
#ifdef KDE_AB_SWITCHES
static bool k0_use_2x2()
{
    static const bool v = KDE_AB_ENV("KDE_K0_2X2") != nullptr;
    return v;
}
#endif

long long presmooth_resident_blocks(int radius)
{
    auto resident = [](auto kernel) -> long long {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kPreBX * kPreBY, 0) != hipSuccess ||
            cus <= 0 || per_cu <= 0) {
            (void)hipGetLastError();
            return 256;
        }
        return (long long)cus * per_cu;
    };
    KDE_AB(if (k0_use_2x2() && radius == 1) return resident(presmooth22_kernel<1, kPre22Waves>);
           if (k0_use_2x2() && radius == 2) return resident(presmooth22_kernel<2, kPre22Waves>);)
    switch (radius) {
        case 1: return resident(presmooth_kernel<1, kPrePxPerThread>);
        case 2: return resident(presmooth_kernel<2, kPrePxPerThread>);
        case 3: return resident(presmooth_kernel<3, kPrePxPerThread>);     // radii 3, 4: the 2 x 2 form needs > 300 VGPRs
        case 4: return resident(presmooth_kernel<4, kPrePxPerThread>);
        default: return 0;                                  // generic kernel: a plain grid
    }
}

// How the row-pair kernel walks its tiles (PreDev::band_walk; tools/ab_k0_walk.sh, profiles/r05_ab_k0_walk.txt):
//   0  linear: tile t runs on XCD t % 8, so the two tiles that share a 128-byte line (a tile row is 96 bytes) always sit in
//      different L2s: K0 read 3.04 x its algorithmic bytes on 64 x 640x480 (total traffic 2.03 x);
//   1  XCD bands: horizontally adjacent tiles meet in one L2 (reads 0.99 x, total 1.02 x) but are staged by DIFFERENT
//      workgroups at the same time; best while a launch's input fits the eight L2s (one 1080p frame 18.1 vs 19.1 us);
//   2  XCD bands in runs of four adjacent tiles (384 bytes = three lines) per workgroup, the remainder after the full runs
//      dealt tile by tile: the shared lines are re-read by the same workgroup a moment later (reads 1.00 x, total 1.04 x); on
//      batches it is as fast as the other two inside the headline step (0.0850 vs 0.0837 linear / 0.0843 bands on 64 x 640x480,
//      within the run-to-run spread; 8 x 1080p: 0.0703 vs 0.0708 / 0.0707) -- at half the HBM traffic of the linear walk.
static int k0_band_walk(int width, int height, int n)
{
    return (long long)width * height * n * 3 <= (32ll << 20) ? 1 : 2;
}

int launch_presmooth(const PresmoothLaunch& a, hipStream_t s)
{
    PreDev d;
    d.src = a.src;
    d.dst = a.dst;
    d.lut = a.lut;
    d.width = a.width;
    d.height = a.height;
    d.n = a.n;
    constexpr int kPX = kPrePxPerThread;
    bool old_form = true;
    d.band_walk = k0_band_walk(a.width, a.height, a.n);
    d.tiles_x = ceil_div(a.width, kPreBX * kPX);
    d.tiles_y = ceil_div(a.height, kPreTH);
#ifdef KDE_AB_SWITCHES
    // measurement build only: KDE_K0_2X2 selects the 2 x 2 form, KDE_K0_BAND_WALK=0/1 forces the walk (tools/ab_k0_band.sh, tools/bench_k0.py)
    old_form = a.radius > 2 || !k0_use_2x2();
    static const int force = [] { const char* e = KDE_AB_ENV("KDE_K0_BAND_WALK"); return e ? (e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1) : -1; }();
    if (force >= 0) d.band_walk = force;
    if (!old_form) {
        d.tiles_x = ceil_div(a.width, kPre22TW);
        d.tiles_y = ceil_div(a.height, kPre22TH);
    }
#endif
    if (a.radius < 1) return fail(KDE_ERR_INVALID, "presmooth: radius %d", a.radius);
    if (a.radius > 4) {
        if (a.n > 65535) return fail(KDE_ERR_INVALID, "presmooth: batch too large for one launch");
        hipLaunchKernelGGL(presmooth_generic_kernel, dim3(ceil_div(a.width, kPreBX), ceil_div(a.height, kPreBY), a.n),
                           dim3(kPreBX * kPreBY), 0, s, d, a.radius);
        KDE_HIP_TRY(hipGetLastError());
        return KDE_OK;
    }
    const long long total = (long long)d.tiles_x * d.tiles_y * a.n;
    if (total > 0x7fffffffLL) return fail(KDE_ERR_INVALID, "presmooth: batch too large for one launch");
    d.div_tpf = make_fastdiv24((uint32_t)(d.tiles_x * d.tiles_y), (uint64_t)total);
    d.div_tx = make_fastdiv24((uint32_t)d.tiles_x, (uint64_t)d.tiles_x * d.tiles_y);
    const long long cap = a.grid_cap > 0 ? a.grid_cap : 256;
    const unsigned grid = (unsigned)(total < cap ? total : cap);
    if (d.band_walk == 2 && grid < 8) d.band_walk = 1;       // the run walk gives every XCD's band to the workgroups ON that XCD: it needs all eight
    if (old_form) {
        switch (a.radius) {
            case 1: hipLaunchKernelGGL((presmooth_kernel<1, kPX>), dim3(grid), dim3(kPreBX * kPreBY), 0, s, d); break;
            case 2: hipLaunchKernelGGL((presmooth_kernel<2, kPX>), dim3(grid), dim3(kPreBX * kPreBY), 0, s, d); break;
            case 3: hipLaunchKernelGGL((presmooth_kernel<3, kPX>), dim3(grid), dim3(kPreBX * kPreBY), 0, s, d); break;
            default: hipLaunchKernelGGL((presmooth_kernel<4, kPX>), dim3(grid), dim3(kPreBX * kPreBY), 0, s, d); break;
        }
    }
#ifdef KDE_AB_SWITCHES
    else {
        if (a.radius == 1) hipLaunchKernelGGL((presmooth22_kernel<1, kPre22Waves>), dim3(grid), dim3(kPreBX * kPreBY), 0, s, d);
        else hipLaunchKernelGGL((presmooth22_kernel<2, kPre22Waves>), dim3(grid), dim3(kPreBX * kPreBY), 0, s, d);
    }
#endif
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

// --------------------------------------------------------------------------------------------
// markov_random_field, tuned form for the reference's 5x5 window (any positive sigmas): the machinery of the
// scalar K1 kernels (jbf_fast.hip) -- colour distance by v_dot4_u32_u8 on packed BGRX against a pre-biased
// -|b|^2 plane whose invalid entries carry -1.7e38, weight = exp2(log2(smooth) - sigma_c*log2(e)*cd) with one
// v_exp_f32 per tap, two horizontally adjacent pixels per thread.  There is no "factor == 0" rule in this
// filter (MarkovRandomField.cu:27-31 multiplies unconditionally), and the centre enters with weight 1, so a
// weight that underflows simply vanishes against a denominator >= 1.
// --------------------------------------------------------------------------------------------
struct MrfFastDev {
    const float* depth;
    const uint8_t* bgr;
    float* out;
    int width, height;
    float kc;       // sigma_c * log2(e)
    float lsm;      // log2(smooth_sigma) + 32 (-inf for smooth_sigma == 0): the taps are summed at 2^32 times their
    float unscale;  // weight and scaled back once (2^-32), so weights in the denormal range are not flushed by v_exp_f32
};

constexpr int kMrfBX = 32, kMrfBY = 8, kMrfPX = 2;

template <int WIN>
__global__ __launch_bounds__(kMrfBX* kMrfBY) void mrf_fast_kernel(MrfFastDev a)
{
    constexpr int R = WIN / 2, NT = kMrfBX * kMrfBY, TW = kMrfBX * kMrfPX, TH = kMrfBY;
    constexpr int LW = TW + 2 * R, LH = TH + 2 * R;
    constexpr uint32_t kMagicM = 0x4B000000u, kOffM = 1u << 18, kInvalidM = 0xFF000000u;
    constexpr float kBiasM = 8388608.0f + 262144.0f;
    __shared__ __attribute__((aligned(8))) float s_d[LH * LW];
    __shared__ __attribute__((aligned(8))) uint32_t s_c[LH * LW];
    __shared__ __attribute__((aligned(8))) uint32_t s_n[LH * LW];

    const size_t frame = (size_t)blockIdx.z * a.width * a.height;
    const float* __restrict__ depth = a.depth + frame;
    const uint8_t* __restrict__ bgr = a.bgr + frame * 3;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int tid = threadIdx.x;
    for (int i = tid; i < LW * LH; i += NT) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float d = 0.0f;
        uint32_t c = 0;
        if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
            const size_t q = (size_t)gy * a.width + gx;
            d = depth[q];
            c = load_bgrx(bgr, q);
        }
        const bool valid = d > 50.0f;
        s_d[i] = valid ? d : 0.0f;
        s_c[i] = c;
        s_n[i] = valid ? (kMagicM + kOffM) - __builtin_amdgcn_udot4(c, c, 0u, false) : kInvalidM;
    }
    __syncthreads();

    const int tx = tid % kMrfBX, ty = tid / kMrfBX;
    const int xb = x0 + tx * kMrfPX, y = y0 + ty;
    if (xb >= a.width || y >= a.height) return;
    // the pair of pixels of this thread in packed math (unit geometry of jbf_fast.hip: straight / swapped / leftover
    // taps out of aligned LDS pairs; LW and the thread's first column are even)
    static_assert(kMrfPX == 2 && LW % 2 == 0, "pair geometry");
    constexpr int HALF = (WIN - 1) / 2, SEGP = 1 + R;
    uint32_t cc[2];
    pre_f2 negC, num = {0.0f, 0.0f}, den = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        cc[k] = s_c[(ty + R) * LW + tx * 2 + R + k];
        negC[k] = -(kBiasM + (float)__builtin_amdgcn_udot4(cc[k], cc[k], 0u, false));
    }
    const pre_f2 kc2 = {a.kc, a.kc}, lsm2 = {a.lsm, a.lsm};
#pragma unroll
    for (int i = 0; i < WIN; i++) {
        pre_f2 dp[SEGP];
        uint2 cp[SEGP], np[SEGP];
#pragma unroll
        for (int m = 0; m < SEGP; m++) {
            const int li = (ty + i) * LW + tx * 2 + 2 * m;
            dp[m] = *reinterpret_cast<const pre_f2*>(&s_d[li]);
            cp[m] = *reinterpret_cast<const uint2*>(&s_c[li]);
            np[m] = *reinterpret_cast<const uint2*>(&s_n[li]);
        }
#pragma unroll
        for (int u = 0; u < WIN; u++) {
            uint32_t c0, c1, n0, n1;
            pre_f2 dq;
            if (u <= HALF) {
                c0 = cp[u].x; c1 = cp[u].y; n0 = np[u].x; n1 = np[u].y; dq = dp[u];
            } else if (u < WIN - 1) {
                const int m = u - HALF;
                c0 = cp[m].y; c1 = cp[m].x; n0 = np[m].y; n1 = np[m].x;
                dq = __builtin_shufflevector(dp[m], dp[m], 1, 0);
            } else {
                c0 = cp[0].y; c1 = cp[HALF].x; n0 = np[0].y; n1 = np[HALF].x;
                dq = pre_f2{dp[0].y, dp[HALF].x};
            }
            // bits of the float 2^23 + 2^18 + 2 a.b - |b|^2; adding negC gives -cd exactly (-1.7e38 if invalid)
            const uint32_t u0 = (__builtin_amdgcn_udot4(c0, cc[0], 0u, false) << 1) + n0;
            const uint32_t u1 = (__builtin_amdgcn_udot4(c1, cc[1], 0u, false) << 1) + n1;
            const pre_f2 ncd = pre_f2{__uint_as_float(u0), __uint_as_float(u1)} + negC;
            const pre_f2 arg = __builtin_elementwise_fma(ncd, kc2, lsm2);
            const pre_f2 f = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
            num = __builtin_elementwise_fma(dq, f, num);
            den = den + f;
        }
    }
    float* __restrict__ o = a.out + frame + (size_t)y * a.width + xb;
#pragma unroll
    for (int k = 0; k < kMrfPX; k++)
        if (xb + k < a.width) {
            // the centre enters with weight 1 whatever its value (MarkovRandomField.cu:15)
            const float n = __builtin_fmaf(num[k], a.unscale, depth[(size_t)y * a.width + xb + k]);
            const float d = __builtin_fmaf(den[k], a.unscale, 1.0f);
            o[k] = (d == 0.0f) ? 0.0f : n / d;
        }
}

int launch_mrf(const MrfLaunch& a, hipStream_t s)
{
    // tuned kernel: the reference's window, both sigmas positive (color_sigma == 0 zeroes every tap in the reference)
    if (a.window == 5 && a.color_sigma > 0.0f && a.smooth_sigma >= 0.0f && a.color_sigma < 1.0e30f && a.smooth_sigma < 1.0e30f) {
        MrfFastDev f;
        f.depth = a.depth; f.bgr = a.bgr; f.out = a.out; f.width = a.width; f.height = a.height;
        f.kc = (float)((double)a.color_sigma * 1.4426950408889634);
        const double lsm = a.smooth_sigma == 0.0f ? -INFINITY : std::log2((double)a.smooth_sigma);
        const int off = lsm + 32.0 < 120.0 ? 32 : 0;
        f.lsm = (float)(lsm + off);
        f.unscale = (float)std::ldexp(1.0, -off);
        dim3 fgrid(ceil_div(a.width, kMrfBX * kMrfPX), ceil_div(a.height, kMrfBY), a.n);
        hipLaunchKernelGGL(mrf_fast_kernel<5>, fgrid, dim3(kMrfBX * kMrfBY), 0, s, f);
        KDE_HIP_TRY(hipGetLastError());
        return KDE_OK;
    }
    MrfDev d{a.depth, a.bgr, a.out, a.width, a.height, a.window, a.color_sigma, a.smooth_sigma};
    const int R = a.window / 2;
    const size_t lds = (size_t)(kTileX + 2 * R) * (kTileY + 2 * R) * 8;
    dim3 grid(ceil_div(a.width, kTileX), ceil_div(a.height, kTileY), a.n);
    hipLaunchKernelGGL(mrf_kernel, grid, dim3(kThreads), lds, s, d);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

}  // namespace kde
