// ers_kernels.hip — EdgeRefinedSuperpixel on gfx950 (K9 edge_refining, K10 depthmap_enhancement).
// Reference: EdgeRefinedSuperpixel/EdgeRefinedSuperpixel.cu:4-102 and :104-205.
//
// K9 is an in-place scatter with cross-thread races in the reference.  It is re-formulated as a
// race-free GATHER with the snapshot semantics D2: in each phase (horizontal, then
// vertical on the horizontal result) every source pixel evaluates its rule on the phase-start
// labels/depth, its depth-zeroing cascade sees only its own writes, and write-sets are applied in
// raster order of the source (later source wins).  A source at s finds its colour edge at distance d <= window/2
// and writes at most [s-(d-1), s+(d-1)] along the scan, so output p inspects sources p+reach ... p-reach with
// reach = window/2 - 1 (descending = last writer first); reach = 2 for the reference's window of 7.
// K10 reads the K9 result and writes a separate buffer (D3).
#include "kde_internal.h"
#include "kde_device_math.h"
#include <type_traits>

namespace kde {
namespace {

// -------------------------------------------------------------------------------------------------
// K9 rule for ONE output position `pos` of a scan line of length `len` (.cu:4-102 in gather form):
// L(k) / C(k) / D(k) read the phase-start depth label, colour label and depth at scan position k.
// Reads stay inside [pos - reach - half, pos + reach + half] and [0, len), reach = half - 1.
// -------------------------------------------------------------------------------------------------
template <typename FL, typename FC, typename FD>
__device__ __forceinline__ void edge_rule(int pos, int len, int half, FL L, FC C, FD D, int32_t& out_label,
                                          float& out_depth)
{
    out_label = L(pos);
    bool have_label = false;
    bool zero = false;
    const int reach = half > 1 ? half - 1 : 0;
    for (int s = pos + reach; s >= pos - reach; s--) {
        if (s < 0 || s + 1 >= len) continue;
        if (L(s) == L(s + 1)) continue;
        const int cur = C(s);
        // search outwards, left candidate first (.cu:25-55)
        int branch = 0, tp = 0;   // 1 = found on the left at tp, 2 = found on the right at tp
        for (int d = 1; d <= half && branch == 0; d++) {
            if (s - d < 0 && s + d >= len) break;
            if (s - d >= 0 && C(s - d) != cur) {
                branch = 1;
                tp = s - d;
            } else if (s + d < len && C(s + d) != cur) {
                branch = 2;
                tp = s + d;
            }
        }
        if (branch == 1) {
            // pixels tp+1 .. s take the label right of the source (.cu:32-38)
            if (pos >= tp + 1 && pos <= s) {
                if (!have_label) {
                    out_label = L(s + 1);
                    have_label = true;
                }
                const float dp = D(pos);
                if (fabsf(dp - D(pos + 1)) > dp * 0.1f) zero = true;
            }
        } else if (branch == 2) {
            // pixels s+1 .. tp-1 take the source's label; the depth test cascades (.cu:45-51)
            if (pos >= s + 1 && pos <= tp - 1) {
                if (!have_label) {
                    out_label = L(s);
                    have_label = true;
                }
                float prev = D(s);
                bool z = false;
                for (int i = s + 1; i <= pos; i++) {
                    float c = D(i);
                    z = fabsf(c - prev) > c * 0.1f;
                    if (z) c = 0.0f;
                    prev = c;
                }
                if (z) zero = true;
            }
        }
    }
    out_depth = zero ? 0.0f : D(pos);
}

// The same rule with every read issued up front and no divergent control flow: the 6 depth labels, 5 + 2*HALF
// colour labels and 4 depths the rule can touch around `pos` are loaded into registers (independent loads instead
// of a chain of ~40 dependent ones), then the five sources are evaluated with selects.  Which source may write
// `pos`, and how far its outward search must reach to do so, is static per source:
//   left branch  (tp = s-d): pos in [tp+1, s]   <=> s >= pos   and d >= s-pos+1
//   right branch (tp = s+d): pos in [s+1, tp-1] <=> s <  pos   and d >= pos-s+1, cascade over D(s..pos)
// INTERIOR: the caller guarantees [pos - 2 - HALF, pos + 3 + HALF] lies inside the line, so no bound is tested.
template <int HALF, bool INTERIOR, typename FL, typename FC, typename FD>
__device__ __forceinline__ void edge_rule_window(int pos, int len, FL L, FC C, FD D, int32_t& out_label, float& out_depth)
{
    auto clampk = [&](int k) {   // out-of-line values are never used
        if constexpr (INTERIOR) return k;
        else return k < 0 ? 0 : (k >= len ? len - 1 : k);
    };
    int32_t Lw[6], Cw[5 + 2 * HALF];
    float Dw[4];
#pragma unroll
    for (int i = 0; i < 6; i++) Lw[i] = L(clampk(pos - 2 + i));
    bool any = false;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int s = pos - 2 + i;
        any |= (INTERIOR || (s >= 0 && s + 1 < len)) && Lw[i] != Lw[i + 1];
    }
    out_label = Lw[2];
    if (!__builtin_amdgcn_ballot_w64(any)) {       // no lane of the wavefront is near a depth-label boundary
        out_depth = D(pos);
        return;
    }
#pragma unroll
    for (int i = 0; i < 5 + 2 * HALF; i++) Cw[i] = C(clampk(pos - 2 - HALF + i));
#pragma unroll
    for (int i = 0; i < 4; i++) Dw[i] = D(clampk(pos - 2 + i));

    bool have_label = false, zero = false;
#pragma unroll
    for (int si = 4; si >= 0; si--) {              // s = pos + 2 ... pos - 2: the last writer is met first
        const int s = pos - 2 + si;
        const bool boundary = (INTERIOR || (s >= 0 && s + 1 < len)) && Lw[si] != Lw[si + 1];
        const int32_t cur = Cw[si + HALF];
        int branch = 0, dist = 0;
#pragma unroll
        for (int d = 1; d <= HALF; d++) {          // left candidate first (.cu:25-55)
            const bool left = (INTERIOR || s - d >= 0) && Cw[si + HALF - d] != cur;
            const bool right = (INTERIOR || s + d < len) && Cw[si + HALF + d] != cur;
            const bool open = branch == 0;
            dist = (open && (left || right)) ? d : dist;
            branch = open ? (left ? 1 : (right ? 2 : 0)) : branch;
        }
        if (si >= 2) {
            // pixels tp+1 .. s take the label right of the source (.cu:32-38)
            const bool hit = boundary && branch == 1 && dist >= si - 1;
            out_label = (hit && !have_label) ? Lw[si + 1] : out_label;
            have_label |= hit;
            zero |= hit && fabsf(Dw[2] - Dw[3]) > Dw[2] * 0.1f;
        } else {
            // pixels s+1 .. tp-1 take the source's label; the depth test cascades from s to pos (.cu:45-51)
            const bool hit = boundary && branch == 2 && dist >= 3 - si;
            out_label = (hit && !have_label) ? Lw[si] : out_label;
            have_label |= hit;
            float prev = Dw[si];
            bool z = false;
#pragma unroll
            for (int i = si + 1; i <= 2; i++) {
                float c = Dw[i];
                z = fabsf(c - prev) > c * 0.1f;
                c = z ? 0.0f : c;
                prev = c;
            }
            zero |= hit && z;
        }
    }
    out_depth = zero ? 0.0f : Dw[2];
}

// K9, one phase on global memory (any window)
template <int DIR>
__global__ __launch_bounds__(256) void edge_phase_kernel(int width, int height, int window,
                                                        const int32_t* __restrict__ color_labels,
                                                        const int32_t* __restrict__ L0, const float* __restrict__ D0,
                                                        int32_t* __restrict__ L1, float* __restrict__ D1)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const int len = DIR == 0 ? width : height;
    const int pos = DIR == 0 ? x : y;
    const size_t stride = DIR == 0 ? 1 : (size_t)width;
    const size_t base = DIR == 0 ? (size_t)y * width : (size_t)x;
    int32_t ol;
    float od;
    edge_rule(pos, len, window / 2,
              [&](int k) { return L0[base + (size_t)k * stride]; },
              [&](int k) { return color_labels[base + (size_t)k * stride]; },
              [&](int k) { return D0[base + (size_t)k * stride]; }, ol, od);
    L1[base + (size_t)pos * stride] = ol;
    D1[base + (size_t)pos * stride] = od;
}

// -------------------------------------------------------------------------------------------------
// K9, both phases in ONE launch.  The rule is a chain of dependent reads along the scan line, so it runs on LDS.
// Around `pos` it reads colour labels at pos - 2 - HALF ... pos + 2 + HALF, depth labels at pos - 2 ... pos + 3 and
// depths at pos - 2 ... pos + 1.  A workgroup owns a 64 x 16 output tile and stages
//   colour labels : tile + (2 + HALF) on every side          (both phases search them),
//   depth labels, depth : tile + 2 before / 3 after          (horizontal phase input along x, its result along y),
// evaluates the horizontal phase for the tile's columns on the 16 + 5 rows the vertical phase will read, in place,
// then the vertical phase from LDS to global memory.  Same snapshot semantics as two launches (D2); the horizontal
// result never travels through HBM.  20 KB of LDS: 8 workgroups per CU, the whole 1080p grid is resident at once.
// -------------------------------------------------------------------------------------------------
constexpr int kEdgeTX = 64, kEdgeTY = 16;

template <int HALF>
__global__ __launch_bounds__(256) void edge_fused_kernel(int width, int height,
                                                        const int32_t* __restrict__ color_labels,
                                                        const int32_t* __restrict__ L0, const float* __restrict__ D0,
                                                        int32_t* __restrict__ L2, float* __restrict__ D2)
{
    constexpr int HC = 2 + HALF;                               // colour-label halo
    constexpr int LA = 2, LB = 3;                              // depth-label / depth halo before / after
    constexpr int CW = kEdgeTX + 2 * HC, CH = kEdgeTY + 2 * HC, CP = 80;       // pitch 80: rows 16 banks apart
    constexpr int LW = kEdgeTX + LA + LB, LH = kEdgeTY + LA + LB, LP = LW + 1;
    static_assert(CP >= CW && kEdgeTX == 64 && 2 * HC <= 16 && LA + LB <= 8 && CH <= 32 && LH <= 32, "staging layout");
    __shared__ int32_t sC[CH * CP];
    __shared__ int32_t sL[LH * LP];              // the horizontal result replaces the tile's columns
    __shared__ float sD[LH * LP];

    const int tid = threadIdx.x;
    const unsigned tiles_x = (unsigned)((width + kEdgeTX - 1) / kEdgeTX);
    const unsigned tiles = tiles_x * (unsigned)((height + kEdgeTY - 1) / kEdgeTY);
    const unsigned gid = xcd_band_id(blockIdx.x, gridDim.x);        // 1-D grid of frames * tiles_x * tiles_y workgroups
    const unsigned frame = gid / tiles, tile = gid - frame * tiles;
    {   // a batch of frames: every plane of frame f starts f * width * height elements further
        const size_t fpx = (size_t)frame * width * height;
        color_labels += fpx; L0 += fpx; D0 += fpx; L2 += fpx; D2 += fpx;
    }
    const int x0 = (int)(tile % tiles_x) * kEdgeTX, y0 = (int)(tile / tiles_x) * kEdgeTY;
    {
        // Staging.  Every load is issued before the first one is consumed: addresses are clamped into the image
        // (always valid) and out-of-image elements are zeroed afterwards.  The 64 tile columns go row by row (one
        // wavefront = one 256-byte row segment), the halo columns many rows at a time; no index division anywhere.
        constexpr int CR = (CH + 3) / 4, LR = (LH + 3) / 4;   // rounds of 4 rows
        const int mc = tid & 63, mr = tid >> 6;
        const int mcx = min(x0 + mc, width - 1);
        const int chi = tid & 15, chrow = tid >> 4;            // colour halo: 16 rows per round, 2*HC of 16 lanes
        const int chcol = chi < HC ? chi : (chi < 2 * HC ? kEdgeTX + chi : CW - 1);
        const int chcx = min(max(x0 - HC + chcol, 0), width - 1);
        const int lhi = tid & 7, lhrow = tid >> 3;             // label / depth halo: 32 rows in one round, LA+LB of 8 lanes
        const int lhcol = lhi < LA ? lhi : (lhi < LA + LB ? kEdgeTX + lhi : LW - 1);
        const int lhcx = min(max(x0 - LA + lhcol, 0), width - 1);
        auto row_c = [&](int r) { return (size_t)min(max(y0 - HC + r, 0), height - 1) * width; };
        auto row_l = [&](int r) { return (size_t)min(max(y0 - LA + r, 0), height - 1) * width; };
        int32_t rc[CR + 2], rl[LR + 1];
        float rd[LR + 1];
#pragma unroll
        for (int k = 0; k < CR; k++) rc[k] = color_labels[row_c(min(mr + 4 * k, CH - 1)) + mcx];
#pragma unroll
        for (int k = 0; k < 2; k++) rc[CR + k] = color_labels[row_c(min(chrow + 16 * k, CH - 1)) + chcx];
#pragma unroll
        for (int k = 0; k < LR; k++) {
            const size_t q = row_l(min(mr + 4 * k, LH - 1)) + mcx;
            rl[k] = L0[q];
            rd[k] = D0[q];
        }
        {
            const size_t q = row_l(min(lhrow, LH - 1)) + lhcx;
            rl[LR] = L0[q];
            rd[LR] = D0[q];
        }
        auto in_x = [&](int gx) { return gx >= 0 && gx < width; };
        auto in_y = [&](int gy) { return gy >= 0 && gy < height; };
#pragma unroll
        for (int k = 0; k < CR; k++) {
            const int r = mr + 4 * k;
            if (r < CH) sC[r * CP + HC + mc] = (in_x(x0 + mc) && in_y(y0 - HC + r)) ? rc[k] : 0;
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int r = chrow + 16 * k;
            if (chi < 2 * HC && r < CH) sC[r * CP + chcol] = (in_x(x0 - HC + chcol) && in_y(y0 - HC + r)) ? rc[CR + k] : 0;
        }
#pragma unroll
        for (int k = 0; k < LR; k++) {
            const int r = mr + 4 * k;
            if (r < LH) {
                const bool in = in_x(x0 + mc) && in_y(y0 - LA + r);
                sL[r * LP + LA + mc] = in ? rl[k] : 0;
                sD[r * LP + LA + mc] = in ? rd[k] : 0.0f;
            }
        }
        if (lhi < LA + LB && lhrow < LH) {
            const bool in = in_x(x0 - LA + lhcol) && in_y(y0 - LA + lhrow);
            sL[lhrow * LP + lhcol] = in ? rl[LR] : 0;
            sD[lhrow * LP + lhcol] = in ? rd[LR] : 0.0f;
        }
    }
    __syncthreads();

    // horizontal phase: scan position = global x, line = staged row r.  A wavefront takes a block of 16 columns x 4
    // rows per round (not 64 x 1): superpixels are wider than 16 pixels, so most blocks see no depth-label boundary
    // along x and leave through the rule's wavefront-uniform early exit.  Results wait in registers until every
    // wavefront has read its inputs, then replace the tile's columns in place.
    constexpr int HROUNDS = (LH + 3) / 4;
    const bool inner_x = x0 - HC >= 0 && x0 + kEdgeTX + HC < width;                 // workgroup-uniform
    const int hc = (tid >> 6) * 16 + (tid & 15), hr = (tid >> 4) & 3;
    int32_t hl[HROUNDS];
    float hd[HROUNDS];
#pragma unroll
    for (int k = 0; k < HROUNDS; k++) {
        const int r = 4 * k + hr;                                  // row of the label / depth planes
        const int gx = x0 + hc, gy = y0 - LA + r;
        hl[k] = 0;
        hd[k] = 0.0f;
        if (r >= LH || gx >= width || gy < 0 || gy >= height) continue;
        const int lb = r * LP + LA - x0, cb = (r + HC - LA) * CP + HC - x0;       // staged index of scan position k is base + k
        auto fl = [&](int k) { return sL[lb + k]; };
        auto fc = [&](int k) { return sC[cb + k]; };
        auto fd = [&](int k) { return sD[lb + k]; };
        if (inner_x) edge_rule_window<HALF, true>(gx, width, fl, fc, fd, hl[k], hd[k]);
        else edge_rule_window<HALF, false>(gx, width, fl, fc, fd, hl[k], hd[k]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HROUNDS; k++) {
        const int r = 4 * k + hr;
        const int gx = x0 + hc, gy = y0 - LA + r;
        if (r >= LH || gx >= width || gy < 0 || gy >= height) continue;
        sL[r * LP + LA + hc] = hl[k];
        sD[r * LP + LA + hc] = hd[k];
    }
    __syncthreads();

    // vertical phase on the horizontal result: scan position = global y, line = column c
    const bool inner_y = y0 - HC >= 0 && y0 + kEdgeTY + HC < height;
#pragma unroll
    for (int i = tid; i < kEdgeTX * kEdgeTY; i += 256) {
        const int ry = i >> 6, c = i & 63;
        const int gx = x0 + c, gy = y0 + ry;
        if (gx >= width || gy >= height) continue;
        auto fl = [&](int k) { return sL[(k - y0 + LA) * LP + LA + c]; };
        auto fc = [&](int k) { return sC[(k - y0 + HC) * CP + HC + c]; };
        auto fd = [&](int k) { return sD[(k - y0 + LA) * LP + LA + c]; };
        int32_t ol;
        float od;
        if (inner_y) edge_rule_window<HALF, true>(gy, height, fl, fc, fd, ol, od);
        else edge_rule_window<HALF, false>(gy, height, fl, fc, fd, ol, od);
        const size_t q = (size_t)gy * width + gx;
        L2[q] = ol;
        D2[q] = od;
    }
}

// -------------------------------------------------------------------------------------------------
// K10 depthmap_enhancement — three passes over a (window x window) neighbourhood staged in LDS.
// -------------------------------------------------------------------------------------------------
constexpr int kTileX = 32, kTileY = 8, kThreads = 256;

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

struct EnhDev {
    const float* rd;
    const uint8_t* bgr;
    const int32_t* labels;
    const float* s_eff;
    float* out;
    int width, height, window;
    float color_sigma, depth_sigma;
    float exp_zero;   // exp(-x) == 0 in binary32 iff x >= exp_zero
    KDE_STAGE(float* stage_avg; float* stage_dev;)      // tools/hooks/libkde_hip_stage.so only
};

__global__ __launch_bounds__(kThreads) void enhance_kernel(EnhDev a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int R = a.window / 2;
    const int LW = kTileX + 2 * R, LH = kTileY + 2 * R;
    float* sd = reinterpret_cast<float*>(smem);
    uint32_t* sc = reinterpret_cast<uint32_t*>(sd + LW * LH);
    int32_t* sl = reinterpret_cast<int32_t*>(sc + LW * LH);
    float* ss = reinterpret_cast<float*>(sl + LW * LH);

    const int x0 = blockIdx.x * kTileX, y0 = blockIdx.y * kTileY;
    const int tid = threadIdx.x;
    for (int i = tid; i < LW * LH; i += kThreads) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float d = 0.0f;
        uint32_t c = 0;
        int32_t l = 0;
        if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
            const size_t q = (size_t)gy * a.width + gx;
            d = a.rd[q];
            if (!(d > 50.0f)) d = 0.0f;
            c = load_bgrx(a.bgr, q);
            l = a.labels[q];
        }
        sd[i] = d;
        sc[i] = c;
        sl[i] = l;
    }
    for (int i = tid; i < a.window * a.window; i += kThreads) ss[i] = a.s_eff[i];
    __syncthreads();

    const int tx = tid & (kTileX - 1), ty = tid / kTileX;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= a.width || y >= a.height) return;

    const int ci = (ty + R) * LW + tx + R;
    const uint32_t cc = sc[ci];
    const int32_t cl = sl[ci];
    float color_sigma = a.color_sigma;   // mutated in pass 3 (.cu:171-176)
    const float den0 = 2 * (color_sigma * color_sigma);
    const float dden = 2.0f * (a.depth_sigma * a.depth_sigma);

    // pass 1: label-restricted weighted average (.cu:116-139)
    float w_average = 0.0f, weight = 0.0f;
    for (int i = 0; i < a.window; i++)
        for (int j = 0; j < a.window; j++) {
            const int li = (ty + i) * LW + tx + j;
            const float dq = sd[li];
            if (dq > 50.0f && sl[li] == cl) {
                float filter = ss[i * a.window + j];
                if (color_sigma != 0.0f) {
                    const float xarg = (float)color_dist2(cc, sc[li]) / den0;
                    if (!(xarg >= a.exp_zero)) filter *= exp_denormal(-xarg);
                }
                w_average += dq * filter;
                weight += filter;
            }
        }
    float result = 0.0f;
    KDE_STAGE(if (a.stage_avg) a.stage_avg[(size_t)y * a.width + x] = weight > 0.0f ? w_average / weight : __builtin_nanf("");)
    if (weight > 0.0f) {
        w_average /= weight;
        // pass 2: mean absolute deviation over the same taps (.cu:143-156)
        int count = 0;
        float deviation = 0.0f;
        for (int i = 0; i < a.window; i++)
            for (int j = 0; j < a.window; j++) {
                const int li = (ty + i) * LW + tx + j;
                const float dq = sd[li];
                if (dq > 50.0f && sl[li] == cl) {
                    deviation += fabsf(dq - w_average);
                    count++;
                }
            }
        if (count != 0) deviation /= (float)count;
        KDE_STAGE(if (a.stage_dev) a.stage_dev[(size_t)y * a.width + x] = deviation;)
        // .cu:171: 5.0 is a double literal, pow(float,float) is float
        const float adaptive_sigma = (float)(5.0 * (double)deviation / (double)(w_average * w_average));
        // pass 3: all valid taps, colour sigma mutating per valid tap in raster order (.cu:158-195)
        float numerator = 0.0f, denominator = 0.0f;
        for (int i = 0; i < a.window; i++)
            for (int j = 0; j < a.window; j++) {
                const int li = (ty + i) * LW + tx + j;
                const float dq = sd[li];
                if (dq > 50.0f) {
                    float filter = ss[i * a.window + j];
                    if (color_sigma != 0.0f) {
                        if (adaptive_sigma > color_sigma * 0.3f) color_sigma = adaptive_sigma;
                        else color_sigma *= 0.3f;
                        // den may underflow to 0: cd/0 = inf -> factor skipped; 0/0 = NaN -> NaN result (Q6)
                        const float xarg = (float)color_dist2(cc, sc[li]) / (2 * (color_sigma * color_sigma));
                        if (!(xarg >= a.exp_zero)) filter *= exp_denormal(-xarg);
                    }
                    if (a.depth_sigma != 0.0f) {
                        const float dd = dq - w_average;
                        const float xd = (dd * dd) / dden;
                        if (!(xd >= a.exp_zero)) filter *= exp_denormal(-xd);
                    }
                    numerator += dq * filter;
                    denominator += filter;
                }
            }
        result = (denominator == 0.0f) ? 0.0f : numerator / denominator;
    }
    a.out[(size_t)y * a.width + x] = result;
}


// -------------------------------------------------------------------------------------------------
// K10, tuned form for the reference's 7x7 window.  Same semantics as enhance_kernel above, restated so
// that no tap needs a division or an expf():
//   * colour distance by v_dot4_u32_u8 on packed BGRX with the pre-biased -|b|^2 plane (see jbf_fast.hip);
//     -cd is exact and stays in registers (49 values) for pass 3;
//   * weights in the log2 domain, one v_exp_f32 per tap in passes 1 and 3;
//   * pass 3's mutating colour sigma is cs_k = max(a, c_k) with c_k = fl(0.3 * c_{k-1}) and k the rank of the
//     tap among the valid taps (raster order), so 1/(2 cs_k^2) = min(1/(2 a^2), 1/(2 c_k^2)): the c_k terms
//     come from a 50-entry table built on the host with the reference's float operations (an entry is +inf
//     where 2 c_k^2 underflows to exactly 0, which reproduces the 0/0 = NaN of Q6);
//   * the "underflowed factor is skipped" decision is taken on the integer colour distance against
//     max(threshold(2 a^2), threshold(2 c_k^2)), both exact (smallest cd with fl(cd/den) >= x0).
// -------------------------------------------------------------------------------------------------
constexpr uint32_t kMagic = 0x4B000000u, kOff = 1u << 18, kInvalidBias = 0xFF000000u;
constexpr float kBiasF = 8388608.0f + 262144.0f;
// Both tuned kernels form every weight at 2^24 times its value (added to the log2(S) table on the host): v_exp_f32
// flushes results below 2^-126, which at this scale is the float32 underflow-to-zero point 2^-150 of the reference's
// product S*cf*df, so weights the reference still holds as denormals keep full precision, weights it rounds to 0
// vanish, and "denominator == 0" (EdgeRefinedSuperpixel.cu:196) falls where the reference's falls (see jbf_fast.hip).
constexpr double kScaleLog2 = 24.0;
constexpr int32_t kInvalidLabel = (int32_t)0x80000000;

struct Enh7Dev {
    const float* rd;
    const uint8_t* bgr;
    const int32_t* labels;
    float* out;
    int width, height;
    float kc1;        // log2(e) / (2 * ColorSigma^2): pass-1 colour term (never underflows at the built sigmas)
    float sd;         // sqrt(log2(e) / (2 * DepthSigma^2))
    float t_skip;     // depth factor skipped when |d_q - avg| * sd >= t_skip
    float exp_zero;   // x0: exp(-x) == 0 in binary32 iff x >= x0
    float ls[49];     // log2 of the spatial table
    float tinv[50];   // [k]: log2(e) / (2 c_k^2), +inf where the denominator underflows to 0
    float tthr[50];   // [k]: -(smallest cd with fl(cd / (2 c_k^2)) >= x0)
    KDE_STAGE(float* stage_avg; float* stage_dev;)      // tools/hooks/libkde_hip_stage.so only
};

__device__ __forceinline__ uint32_t dot4u(uint32_t a, uint32_t b) { return __builtin_amdgcn_udot4(a, b, 0u, false); }

__global__ __launch_bounds__(kThreads) void enhance7_kernel(const Enh7Dev a)
{
    constexpr int WIN = 7, R = 3;
    constexpr int LW = kTileX + 2 * R, LH = kTileY + 2 * R;
    __shared__ float s_d[LH * LW];
    __shared__ uint32_t s_c[LH * LW];
    __shared__ uint32_t s_n[LH * LW];
    __shared__ int32_t s_l[LH * LW];
    __shared__ float2 s_t[50];        // (tinv, tthr) per rank

    const int x0 = blockIdx.x * kTileX, y0 = blockIdx.y * kTileY;
    const int tid = threadIdx.x;
    for (int i = tid; i < LW * LH; i += kThreads) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float d = 0.0f;
        uint32_t c = 0;
        int32_t l = 0;
        if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
            const size_t q = (size_t)gy * a.width + gx;
            d = a.rd[q];
            c = load_bgrx(a.bgr, q);
            l = a.labels[q];
        }
        const bool valid = d > 50.0f;
        s_d[i] = valid ? d : 0.0f;
        s_c[i] = c;
        s_n[i] = valid ? (kMagic + kOff) - dot4u(c, c) : kInvalidBias;
        s_l[i] = valid ? l : kInvalidLabel;
    }
    if (tid < 50) s_t[tid] = make_float2(a.tinv[tid], a.tthr[tid]);
    __syncthreads();

    const int tx = tid & (kTileX - 1), ty = tid / kTileX;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= a.width || y >= a.height) return;
    const size_t p = (size_t)y * a.width + x;
    const uint32_t cc = s_c[(ty + R) * LW + tx + R];
    const int32_t cl = a.labels[p];                       // the centre's label counts even if its depth is invalid
    const float negC = -(kBiasF + (float)dot4u(cc, cc));

    // ---- pass 1: label-restricted weighted average (.cu:116-139) --------------------------------------
    float ncd[WIN * WIN];
    float wsum = 0.0f, wgt = 0.0f;
#pragma unroll
    for (int i = 0; i < WIN; i++)
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            const int li = (ty + i) * LW + tx + j;
            const uint32_t u = (dot4u(s_c[li], cc) << 1) + s_n[li];
            const float nc = __uint_as_float(u) + negC;                // -cd exactly (-1.7e38 for an invalid tap)
            ncd[i * WIN + j] = nc;
            float a1 = __builtin_fmaf(nc, a.kc1, a.ls[i * WIN + j]);
            a1 = (s_l[li] == cl) ? a1 : -3.0e38f;                      // other label or invalid: weight exactly 0
            const float f = __builtin_amdgcn_exp2f(a1);
            wsum = __builtin_fmaf(s_d[li], f, wsum);
            wgt += f;
        }
    float result = 0.0f;
    KDE_STAGE(if (a.stage_avg) a.stage_avg[p] = wsum / wgt;)
    if (wgt > 0.0f) {
        const float wavg = wsum / wgt;
        // ---- pass 2: mean absolute deviation over the same taps (.cu:143-156) ---------------------------
        float deviation = 0.0f;
        int count = 0;
#pragma unroll
        for (int i = 0; i < WIN; i++)
#pragma unroll
            for (int j = 0; j < WIN; j++) {
                const int li = (ty + i) * LW + tx + j;
                const bool m = s_l[li] == cl;
                deviation += m ? fabsf(s_d[li] - wavg) : 0.0f;
                count += m ? 1 : 0;
            }
        if (count != 0) deviation /= (float)count;
        KDE_STAGE(if (a.stage_dev) a.stage_dev[p] = deviation;)
        // .cu:171: 5.0 is a double literal, pow(float,float) is float
        const float asig = (float)(5.0 * (double)deviation / (double)(wavg * wavg));
        const float den_a = 2 * (asig * asig);
        // +inf only where the reference divides by an exact 0 (0/0 = NaN, Q6); a denormal denominator still gives
        // cd/den = 0 for cd == 0, so the scale must stay finite there (cd >= 1 is then decided by the threshold)
        const float inv_a = den_a == 0.0f ? INFINITY : fminf(1.4426950408889634f / den_a, 3.0e38f);
        // smallest integer cd with fl(cd / den_a) >= x0 (exact: candidate from the product, then stepped)
        float thr_a;
        if (den_a == 0.0f) {
            thr_a = 1.0f;                                   // cd/0 = inf for cd >= 1; 0/0 = NaN is not ">= x0"
        } else {
            float c0 = ceilf(a.exp_zero * den_a);
            c0 = fminf(fmaxf(c0, 0.0f), 400000.0f);
            while (c0 > 0.0f && (c0 - 1.0f) / den_a >= a.exp_zero) c0 -= 1.0f;
            while (c0 < 400000.0f && !(c0 / den_a >= a.exp_zero)) c0 += 1.0f;
            thr_a = c0;
        }
        const float nthr_a = -thr_a;
        // ---- pass 3: every valid tap, colour sigma mutating with the tap's rank (.cu:158-195) ------------
        float num = 0.0f, den = 0.0f;
        int k = 0;
#pragma unroll
        for (int i = 0; i < WIN; i++)
#pragma unroll
            for (int j = 0; j < WIN; j++) {
                const int li = (ty + i) * LW + tx + j;
                const float nc = ncd[i * WIN + j];
                const float dq = s_d[li];
                const bool valid = nc > -1.0e37f;
                k += valid ? 1 : 0;
                const float2 tk = s_t[k];
                const float invl = fminf(inv_a, tk.x);
                float ac = __builtin_fmaf(nc, invl, a.ls[i * WIN + j]);        // log2(S) - cd * log2(e)/(2 cs_k^2)
                ac = (valid && nc <= fminf(nthr_a, tk.y)) ? a.ls[i * WIN + j] : ac;   // underflowed colour factor skipped
                const float t = (dq - wavg) * a.sd;
                float a2 = __builtin_fmaf(-t, t, ac);
                a2 = (fabsf(t) >= a.t_skip) ? ac : a2;                            // underflowed depth factor skipped
                const float f = __builtin_amdgcn_exp2f(a2);
                num = __builtin_fmaf(dq, f, num);
                den += f;
            }
        result = (den == 0.0f) ? 0.0f : num / den;
    }
    a.out[p] = result;
}


// -------------------------------------------------------------------------------------------------
// K10, packed-pair form of enhance7_kernel (same semantics, same tolerance): every float add / mul / fma works
// on a PAIR of horizontally adjacent pixels with v_pk_*_f32, and every per-tap decision is a {0,1} mask made by
// the VOP3P clamp bit instead of v_cmp + v_cndmask (see jbf_fast.hip for the unit / pair geometry):
//   * same-label test: labels are staged as floats (superpixel indices, |l| < 2^24; 1e9 for an invalid tap), and
//     m = clamp(1 - (l - l_centre)^2);
//   * invalid taps enter the pass-3 sums with v = clamp(2 d) = 0;
//   * colour rule: m = clamp(thr - cd) with the integer threshold of the tap's rank, arg = fma(-cd, scale*m, ls);
//   * depth rule: m = clamp((T2 - t^2) * 2^100), arg -= (t*m)*t;
//   * the rank of a tap (number of valid taps up to it, scan order) is a running LDS byte address: validity
//     sits in the spare byte of the packed BGRX word as 0 / 8 and is added by one v_add_u32_sdwa per tap; the
//     (scale, threshold) pair of that rank is one ds_read_b64.
// Scales are kept finite (<= 3e38) so that 0 * scale is 0; the reference's 0/0 = NaN case (Q6: flat patch, so
// adaptive sigma == 0, and a rank whose table sigma has underflowed) is re-created by a rare post-pass.
// -------------------------------------------------------------------------------------------------
typedef float e_f2 __attribute__((ext_vector_type(2)));
typedef uint32_t e_u2 __attribute__((ext_vector_type(2)));

constexpr int kE7BX = 32, kE7BY = 8;        // threads; tile = 64 x 8 pixels
constexpr float kLabelScale = 32.0f;        // labels are compared as floats at this scale (see the staging loop)

struct Enh7PkDev {
    const float* rd;
    const uint8_t* bgr;
    const int32_t* labels;
    float* out;
    int width, height;
    float kc1, sd, t2_skip, exp_zero;
    int kinf;                                   // first rank whose table scale is +inf in the reference's arithmetic
    int kfree;                                  // from this rank on the table's threshold is 1: a colour factor is either
                                                // skipped (cd >= 1) or exp(0) = 1 (cd == 0) -- no colour arithmetic needed
    __attribute__((aligned(8))) float lsp[98];  // [(row*7 + unit)*2 + pixel of the pair]: log2 of the spatial table
    float tinv[50];                             // finite
    float tthr[50];
    KDE_STAGE(float* stage_avg; float* stage_dev; unsigned* stage_counters; int stage_force;)   // libkde_hip_stage.so only
};

__device__ __forceinline__ e_f2 e_fma(e_f2 a, e_f2 b, e_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ e_f2 e_bcast(float v) { return e_f2{v, v}; }
__device__ __forceinline__ e_f2 e_add_clamp(e_f2 a, e_f2 b)
{
    e_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ e_f2 e_mul_clamp(e_f2 a, e_f2 b)
{
    e_f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// clamp(c - a*b)
__device__ __forceinline__ e_f2 e_fnma_clamp(e_f2 a, e_f2 b, e_f2 c)
{
    e_f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// v_min_f32 without the canonicalising v_max that fminf() puts in front of it (operands are never signalling NaNs)
__device__ __forceinline__ float min_raw(float x, float y)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// acc + byte 3 of w
__device__ __forceinline__ uint32_t add_byte3(uint32_t w, uint32_t acc)
{
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD"
        : "=v"(r) : "v"(w), "v"(acc));
    return r;
}

// FUSED_LABEL = false keeps r03's form of the pass-1 label test (weight times a {0,1} mask: one packed instruction more per
// unit) for the A/B of tools/bench_chain.py (KDE_K10_MASK_PRODUCT=1); the outputs are bit-identical
// KDE_K10_WAVES = waves per SIMD the register allocator must leave room for (0 = no bound: 80 VGPRs, 6 workgroups per CU).
// A single 1080p frame is 4050 workgroups: on 6 x 256 slots that is 2.64 rounds (the third 64 % full), on 8 x 256 it is
// 1.98 -- tools/ab_k10_waves.sh measures whether the 6 spilled registers of the 64-VGPR build cost less than the tail.
#ifndef KDE_K10_WAVES
#define KDE_K10_WAVES 0
#endif
template <bool FUSED_LABEL>
__global__ __launch_bounds__(kE7BX* kE7BY) __attribute__((amdgpu_waves_per_eu(KDE_K10_WAVES ? KDE_K10_WAVES : 1))) void enhance7_pk_kernel(const Enh7PkDev a)
{
    constexpr int WIN = 7, R = 3, HALF = 3, SEGP = 4;
    constexpr int NT = kE7BX * kE7BY;
    constexpr int TW = kE7BX * 2, TH = kE7BY;
    constexpr int P = TW + 2 * R, LH = TH + 2 * R;          // P is even: a thread's window starts on an even column
    __shared__ __attribute__((aligned(16))) float s_d[LH * P];
    __shared__ __attribute__((aligned(16))) uint32_t s_c[LH * P];   // BGR in bytes 0..2, validity (0 / 8) in byte 3
    __shared__ __attribute__((aligned(16))) uint32_t s_n[LH * P];
    __shared__ __attribute__((aligned(16))) float s_l[LH * P];
    __shared__ __attribute__((aligned(8))) float2 s_t[50];

    __shared__ uint32_t s_rng[2];                 // bit patterns of the smallest / largest valid depth staged (tile + halo)

    const unsigned tiles_x = (unsigned)((a.width + TW - 1) / TW);
    const unsigned tiles = tiles_x * (unsigned)((a.height + TH - 1) / TH);
    const unsigned gid = xcd_band_id(blockIdx.x, gridDim.x);        // 1-D grid of frames * tiles_x * tiles_y workgroups
    const unsigned frame = gid / tiles, tile = gid - frame * tiles;
    const size_t fpx = (size_t)frame * a.width * a.height;          // a batch of frames: planes of frame f start f * W * H further
    const float* __restrict__ in_rd = a.rd + fpx;
    const uint8_t* __restrict__ in_bgr = a.bgr + fpx * 3;
    const int32_t* __restrict__ in_labels = a.labels + fpx;
    float* __restrict__ out_d = a.out + fpx;
    const int x0 = (int)(tile % tiles_x) * TW, y0 = (int)(tile / tiles_x) * TH;
    const int tid = threadIdx.x;
    if (tid < 2) s_rng[tid] = tid == 0 ? 0x7f800000u : 0u;
    __syncthreads();
    uint32_t st_min = 0x7f800000u, st_max = 0u;
    for (int i = tid; i < P * LH; i += NT) {
        const int ly = i / P, lx = i - ly * P;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float d = 0.0f;
        uint32_t c = 0;
        int32_t l = 0;
        if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) {
            const size_t q = (size_t)gy * a.width + gx;
            d = in_rd[q];
            c = load_bgrx(in_bgr, q);
            l = in_labels[q];
        }
        const bool valid = d > 50.0f;
        s_d[i] = valid ? d : 0.0f;
        s_c[i] = c | (valid ? 0x08000000u : 0u);
        s_n[i] = (kMagic + kOff) - dot4u(c, c);
        // labels are staged at kLabelScale (32) times their value -- exact: |label| < 2^24 -- so that two different labels
        // differ by >= 32 and (l_q - l_c)^2 >= 1024 can be SUBTRACTED from a log2-domain weight argument to flush it (r04);
        // an invalid tap never comes near a label (-1 = unassigned included)
        s_l[i] = valid ? kLabelScale * (float)l : 1.0e12f;
        if (valid) {                                  // positive floats order like their bit patterns
            st_min = min(st_min, __float_as_uint(d));
            st_max = max(st_max, __float_as_uint(d));
        }
    }
    if (tid < 50) s_t[tid] = make_float2(a.tinv[tid], a.tthr[tid]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        st_min = min(st_min, (uint32_t)__shfl_xor((int)st_min, m, 64));
        st_max = max(st_max, (uint32_t)__shfl_xor((int)st_max, m, 64));
    }
    if ((tid & 63) == 0) {
        atomicMin(&s_rng[0], st_min);
        atomicMax(&s_rng[1], st_max);
    }
    __syncthreads();

    // Is the adaptive sigma a = 5 dev / avg^2 provably small for EVERY pixel of this tile?  dev is a mean of |d - avg|
    // over taps whose depths (and whose average) lie inside the tile's valid range [dmin, dmax], so a <= 5 (dmax - dmin) /
    // dmin^2.  "Small" means what pass 3 needs to ignore it: its skip threshold is 1 (fl(1 / 2a^2) >= x0) and its scale
    // is not below the table's at rank kfree - 1.  Then the per-pixel deviation pass, the division and the threshold
    // search are skipped; the deviation is only computed for pixels that could hit the 0/0 quirk (below).
    bool tile_small_a = false, tile_no_drule = false;
    {
        const float dmin = __uint_as_float(s_rng[0]), dmax = __uint_as_float(s_rng[1]);
        if (s_rng[1] >= s_rng[0]) {
            // valid depths and every average of them lie in [dmin, dmax]: |d - avg| * sd <= span for every valid tap
            // (+ slack: the float32 average can leave [dmin, dmax] by the rounding of its sums, (taps + 8) / 2 ulps of dmax)
            const float slack = dmax * (float)(49 + 8) * 0x1p-24f;
            const float span = ((dmax - dmin) * 1.001f + slack) * a.sd;
            tile_no_drule = span * span < a.t2_skip * 0.999f;
            const float ab = 5.0f * ((dmax - dmin) + slack) / (dmin * dmin) * 1.001f;
            const float denb = 2.0f * (ab * ab);
            const float tl = a.tinv[a.kfree > 0 ? a.kfree - 1 : 0];
            tile_small_a = a.kfree < 50 && (denb == 0.0f || (1.0f / denb >= a.exp_zero * 1.001f && 1.4426950408889634f / denb >= tl * 1.001f));
        }
    }
    KDE_STAGE(if (a.stage_force) { tile_small_a = false; tile_no_drule = false; })
    KDE_STAGE(if (a.stage_counters && tid == 0) {
        atomicAdd(&a.stage_counters[tile_small_a ? 4 : 5], 1u);
        atomicAdd(&a.stage_counters[tile_no_drule ? 6 : 7], 1u);
    })

    const int tx = tid % kE7BX, ty = tid / kE7BX;
    const int xb = x0 + 2 * tx, y = y0 + ty;
    if (xb >= a.width || y >= a.height) return;
    const bool has1 = xb + 1 < a.width;
    const int sx = 2 * tx;                                   // LDS column of this pair's first window column
    const size_t p = (size_t)y * a.width + xb;

    uint32_t cc[2];
    e_f2 negC, cl;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        cc[h] = s_c[(ty + R) * P + sx + R + h] & 0x00ffffffu;
        negC[h] = -(kBiasF + (float)dot4u(cc[h], cc[h]));
    }
    cl.x = kLabelScale * (float)in_labels[p];                 // the centre's label counts even if its depth is invalid
    cl.y = has1 ? kLabelScale * (float)in_labels[p + 1] : 2.0e12f;

    // unit u of a row: (tap of p0, tap of p1) taken from ONE aligned LDS pair -- straight / swapped / leftover
    auto pick_u = [&](const e_u2* v, int u, uint32_t& v0, uint32_t& v1) {
        if (u <= HALF) { v0 = v[u].x; v1 = v[u].y; }
        else if (u < WIN - 1) { v0 = v[u - HALF].y; v1 = v[u - HALF].x; }
        else { v0 = v[0].y; v1 = v[HALF].x; }
    };
    auto pick_f = [&](const e_f2* v, int u) -> e_f2 {
        if (u <= HALF) return v[u];
        if (u < WIN - 1) return __builtin_shufflevector(v[u - HALF], v[u - HALF], 1, 0);
        return e_f2{v[0].y, v[HALF].x};
    };
    auto load_f = [&](const float* plane, int i, e_f2* out) {
#pragma unroll
        for (int m = 0; m < SEGP; m++) out[m] = *reinterpret_cast<const e_f2*>(&plane[(ty + i) * P + sx + 2 * m]);
    };
    auto load_u = [&](const uint32_t* plane, int i, e_u2* out) {
#pragma unroll
        for (int m = 0; m < SEGP; m++) out[m] = *reinterpret_cast<const e_u2*>(&plane[(ty + i) * P + sx + 2 * m]);
    };
    const e_f2 one = e_bcast(1.0f);
    auto label_mask = [&](const e_f2* lp, int u) -> e_f2 {
        const e_f2 dl = pick_f(lp, u) - cl;
        return e_fnma_clamp(dl, dl, one);                    // 1 for the centre's label, 0 otherwise (and for invalid taps)
    };

    // -cd (exact) of the two taps of unit u
    auto unit_ncd = [&](const e_u2* cp, const e_u2* np, int u) -> e_f2 {
        uint32_t c0, c1, n0, n1;
        pick_u(cp, u, c0, c1);
        pick_u(np, u, n0, n1);
        const uint32_t u0 = (dot4u(c0, cc[0]) << 1) + n0;
        const uint32_t u1 = (dot4u(c1, cc[1]) << 1) + n1;
        return e_f2{__uint_as_float(u0), __uint_as_float(u1)} + negC;
    };

    // ---- pass 1: label-restricted weighted average (.cu:116-139) --------------------------------------
    // (row loops stay rolled: 100 VGPRs / 4 waves per SIMD; -cd is recomputed in pass 3 rather than cached)
    e_f2 wsum = e_bcast(0.0f), wgt = e_bcast(0.0f);
    e_f2 kc1 = e_bcast(a.kc1);
    asm volatile("" : "+v"(kc1));          // VGPR pair: the only scalar operand of the argument fma is the log2(S) pair
#pragma unroll 1
    for (int i = 0; i < WIN; i++) {
        e_f2 dp[SEGP], lp[SEGP];
        e_u2 cp[SEGP], np[SEGP];
        load_f(s_d, i, dp);
        load_f(s_l, i, lp);
        load_u(s_c, i, cp);
        load_u(s_n, i, np);
#pragma unroll
        for (int u = 0; u < WIN; u++) {
            const e_f2 nc = unit_ncd(cp, np, u);
            const e_f2 lsj = *reinterpret_cast<const e_f2*>(&a.lsp[(i * WIN + u) * 2]);
            const e_f2 a1 = e_fma(nc, kc1, lsj);
            // same-label test folded into the argument (r04; was f * clamp(1 - dl^2): one packed instruction more per unit):
            // dl == 0 leaves a1 untouched (-0 * 0 + a1), any other label subtracts >= 1024 and v_exp_f32 returns exactly 0 --
            // the same bits as the product with the {0,1} mask
            e_f2 f;
            if constexpr (FUSED_LABEL) {
                const e_f2 dl = pick_f(lp, u) - cl;
                const e_f2 a1m = e_fma(-dl, dl, a1);
                f = e_f2{__builtin_amdgcn_exp2f(a1m.x), __builtin_amdgcn_exp2f(a1m.y)};
            } else {
                f = e_f2{__builtin_amdgcn_exp2f(a1.x), __builtin_amdgcn_exp2f(a1.y)} * label_mask(lp, u);
            }
            wsum = e_fma(pick_f(dp, u), f, wsum);
            wgt = wgt + f;
        }
    }

    // ---- pass 2: mean absolute deviation over the same taps (.cu:143-156) and, from it, the adaptive sigma's scale,
    // integer skip threshold and "is exactly 0" flag (.cu:171).  Runs before pass 3 when the tile could not prove the
    // adaptive sigma small, otherwise only afterwards and only for wavefronts with a candidate for the 0/0 quirk.
    const e_f2 wavg = e_f2{wsum.x / wgt.x, wsum.y / wgt.y};    // IEEE: the integer skip thresholds below depend on it
    e_f2 inv_a = e_bcast(3.0e38f), nthr_a = e_bcast(-1.0f);     // values of a provably small adaptive sigma
    bool flat[2] = {false, false};
    KDE_STAGE(float stage_dev2[2] = {0.0f, 0.0f}; bool stage_dev_done = false;)
    auto adaptive_sigma = [&]() {
        e_f2 dev = e_bcast(0.0f), cnt = e_bcast(0.0f);
#pragma unroll 1
        for (int i = 0; i < WIN; i++) {
            e_f2 dp[SEGP], lp[SEGP];
            load_f(s_d, i, dp);
            load_f(s_l, i, lp);
#pragma unroll
            for (int u = 0; u < WIN; u++) {
                const e_f2 m = label_mask(lp, u);
                const e_f2 e = pick_f(dp, u) - wavg;
                dev.x = __builtin_fmaf(__builtin_fabsf(e.x), m.x, dev.x);
                dev.y = __builtin_fmaf(__builtin_fabsf(e.y), m.y, dev.y);
                cnt = cnt + m;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            float deviation = dev[h];
            if (cnt[h] != 0.0f) deviation /= cnt[h];
            KDE_STAGE(stage_dev2[h] = deviation; stage_dev_done = true;)
            // 5.0 is a double literal, pow(float,float) is float
            const float asig = (float)(5.0 * (double)deviation / (double)(wavg[h] * wavg[h]));
            const float den_a = 2 * (asig * asig);
            flat[h] = den_a == 0.0f;
            inv_a[h] = flat[h] ? 3.0e38f : fminf(1.4426950408889634f / den_a, 3.0e38f);
            float thr_a;                         // smallest integer cd with fl(cd / den_a) >= x0 (exact)
            if (flat[h]) {
                thr_a = 1.0f;                    // cd/0 = inf for cd >= 1; 0/0 = NaN is not ">= x0" (post-pass below)
            } else if (den_a != den_a) {
                thr_a = 400000.0f;               // NaN sigma (non-finite depths): nothing compares ">= x0"
            } else {
                float c0 = ceilf(a.exp_zero * den_a);
                c0 = fminf(fmaxf(c0, 0.0f), 400000.0f);
                while (c0 > 0.0f && (c0 - 1.0f) / den_a >= a.exp_zero) c0 -= 1.0f;
                while (c0 < 400000.0f && !(c0 / den_a >= a.exp_zero)) c0 += 1.0f;
                thr_a = c0;
            }
            nthr_a[h] = -thr_a;
        }
    };
    if (!tile_small_a) adaptive_sigma();

    // ---- pass 3: every valid tap, colour sigma mutating with the tap's rank (.cu:158-195) ------------
    // Rank cut-off: once the decayed table sigma c_k is so small that 2 c_k^2 * x0 < 1 (rank kfree = 6 at the
    // reference's ColorSigma), a tap's colour factor is exp(-cd / (2 cs^2)) with cs = max(a, c_k): for cd >= 1 it has
    // underflowed to 0 and is SKIPPED, for cd == 0 it is exp(0) = 1 -- either way the weight is S * df, provided the
    // adaptive sigma a is not large itself (thr_a <= 1 <=> 2 a^2 * x0 <= 1; a = 5 dev / avg^2 is ~1e-5 on real depth).
    // So as soon as every lane of the wavefront has passed rank kfree on both of its pixels, the remaining ROWS run a
    // loop without any colour arithmetic (12 instead of 28 issue slots per unit; typically rows 1..6 of 7).  Lanes
    // with holes in their first row simply keep the wave in the general loop one row longer.
    const e_f2 T2 = e_bcast(a.t2_skip), kBig = e_bcast(0x1p100f), sd2 = e_bcast(a.sd);
    e_f2 num = e_bcast(0.0f), den = e_bcast(0.0f);
    uint32_t rank0 = 0, rank1 = 0;           // byte offsets into s_t (8 bytes per rank)
    const char* tbase = reinterpret_cast<const char*>(s_t);
    const float tinv_last = a.tinv[a.kfree > 0 ? a.kfree - 1 : 0];
    const bool a_small = nthr_a.x >= -1.0f && nthr_a.y >= -1.0f && inv_a.x >= tinv_last && inv_a.y >= tinv_last;
    const uint32_t free_off = (uint32_t)a.kfree * 8u;
    int row = 0;
    // general loop: rows whose taps may still be below rank kfree on some lane of the wavefront
#pragma unroll 1
    for (; row < WIN; row++) {
        const int i = row;
        e_f2 dp[SEGP], vp[SEGP];
        e_u2 cp[SEGP], np[SEGP];
        load_f(s_d, i, dp);
        load_u(s_c, i, cp);
        load_u(s_n, i, np);
#pragma unroll
        for (int m = 0; m < SEGP; m++) vp[m] = e_add_clamp(dp[m], dp[m]);
        // scan-order ranks of the row's taps: p0 covers columns 0..6 of the segment, p1 columns 1..7
        uint32_t r0[WIN], r1[WIN];
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            const uint32_t w0 = (j & 1) ? cp[j >> 1].y : cp[j >> 1].x;
            const uint32_t w1 = ((j + 1) & 1) ? cp[(j + 1) >> 1].y : cp[(j + 1) >> 1].x;
            rank0 = add_byte3(w0, rank0);
            rank1 = add_byte3(w1, rank1);
            r0[j] = rank0;
            r1[j] = rank1;
        }
#pragma unroll
        for (int u = 0; u < WIN; u++) {
            const int j0 = u <= HALF ? 2 * u : (u < WIN - 1 ? 2 * (u - HALF) + 1 : 1);
            const int j1 = u <= HALF ? 2 * u : (u < WIN - 1 ? 2 * (u - HALF) - 1 : WIN - 2);
            const float2 t0 = *reinterpret_cast<const float2*>(tbase + r0[j0]);
            const float2 t1 = *reinterpret_cast<const float2*>(tbase + r1[j1]);
            const e_f2 invl = e_f2{min_raw(inv_a.x, t0.x), min_raw(inv_a.y, t1.x)};
            const e_f2 thr = e_f2{min_raw(nthr_a.x, t0.y), min_raw(nthr_a.y, t1.y)};
            const e_f2 nc = unit_ncd(cp, np, u);
            const e_f2 lsj = *reinterpret_cast<const e_f2*>(&a.lsp[(i * WIN + u) * 2]);
            const e_f2 mc = e_add_clamp(nc, -thr);                       // 0 <=> -cd <= thr: underflowed colour factor skipped
            const e_f2 ac = e_fma(nc, mc * invl, lsj);                    // log2(S) - cd * log2(e)/(2 cs_k^2)
            const e_f2 dq = pick_f(dp, u);
            const e_f2 t = (dq - wavg) * sd2;
            const e_f2 md = e_mul_clamp(e_fma(-t, t, T2), kBig);          // 0 <=> underflowed depth factor skipped
            const e_f2 a2 = e_fma(-(t * md), t, ac);
            const e_f2 f = e_f2{__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
            num = e_fma(dq, f, num);
            den = e_fma(pick_f(vp, u), f, den);
        }
        // every (active) lane past rank kfree on both pixels, with a small adaptive sigma?  then no later tap needs colour
        const bool lane_free = a_small && rank0 >= free_off && rank1 >= free_off;
        if (__builtin_amdgcn_ballot_w64(!lane_free) == 0) {       // wave-uniform
            row++;
            break;
        }
    }
    // colour-free loop for the remaining rows.  Rule elision as in K1: when the staged depth range of the tile proves
    // that no valid tap can be further from ANY average than the depth-factor underflow distance (1009 mm at the
    // reference's DepthSigma), the factor never underflows, md == 1 exactly and its three instructions per unit go
    // (75 -> 72 us at 1080p; the same elision in the general rows above measured no further gain).
    auto free_rows = [&](auto depth_rule) {
        constexpr bool DRULE = decltype(depth_rule)::value;
#pragma unroll 1
        for (; row < WIN; row++) {
            const int i = row;
            e_f2 dp[SEGP], vp[SEGP];
            load_f(s_d, i, dp);
#pragma unroll
            for (int m = 0; m < SEGP; m++) vp[m] = e_add_clamp(dp[m], dp[m]);
#pragma unroll
            for (int u = 0; u < WIN; u++) {
                const e_f2 lsj = *reinterpret_cast<const e_f2*>(&a.lsp[(i * WIN + u) * 2]);
                const e_f2 dq = pick_f(dp, u);
                const e_f2 t = (dq - wavg) * sd2;
                e_f2 a2;
                if constexpr (DRULE) {
                    const e_f2 md = e_mul_clamp(e_fma(-t, t, T2), kBig);      // 0 <=> underflowed depth factor skipped
                    a2 = e_fma(-(t * md), t, lsj);
                } else {
                    a2 = e_fma(-t, t, lsj);       // an invalid tap (d = 0) gets some finite weight here; its vp and d are 0
                }
                const e_f2 f = e_f2{__builtin_amdgcn_exp2f(a2.x), __builtin_amdgcn_exp2f(a2.y)};
                num = e_fma(dq, f, num);
                den = e_fma(pick_f(vp, u), f, den);
            }
        }
    };
    if (tile_no_drule) free_rows(std::false_type{});
    else free_rows(std::true_type{});

    float res[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        res[h] = 0.0f;
        if (wgt[h] > 0.0f) res[h] = (den[h] == 0.0f) ? 0.0f : num[h] / den[h];
    }
    // In a tile with a provably small adaptive sigma the deviation has not been computed yet.  It only matters for the
    // 0/0 quirk below, which needs a valid tap of the centre's exact colour at a rank >= kinf (47): such a tap can only
    // be one of the last 50 - kinf taps of the window (rank <= position + 1).  Wavefronts without such a pixel are done.
    if (tile_small_a) {
        bool cand = false;
        if (a.kinf <= WIN * WIN) {
            // ... and the quirk needs a deviation of exactly 0, i.e. EVERY valid tap of the centre's label equal to the
            // average -- the centre itself is such a tap when it is valid, so a valid centre that differs rules it out
            const int lc = (ty + R) * P + sx + R;
            const float dc0 = s_d[lc], dc1 = s_d[lc + 1];
            const bool may0 = wgt.x > 0.0f && (dc0 == 0.0f || dc0 == wavg.x), may1 = wgt.y > 0.0f && (dc1 == 0.0f || dc1 == wavg.y);
#pragma unroll 1
            for (int pos = a.kinf - 1 < 0 ? 0 : a.kinf - 1; pos < WIN * WIN; pos++) {
                const int li = (ty + pos / WIN) * P + sx + pos % WIN;
                cand |= may0 && s_d[li] > 0.0f && (s_c[li] & 0x00ffffffu) == cc[0];
                cand |= may1 && s_d[li + 1] > 0.0f && (s_c[li + 1] & 0x00ffffffu) == cc[1];
            }
        }
        if (__builtin_amdgcn_ballot_w64(cand) != 0) adaptive_sigma();       // wave-uniform; sets flat[]
    }
    // Q6 post-pass (rare): flat patch and a valid tap of the centre's exact colour whose rank has a zero table
    // denominator -> the reference evaluates 0/0 = NaN and the NaN reaches the result
    if ((flat[0] && wgt.x > 0.0f) || (flat[1] && wgt.y > 0.0f)) {
        int k0 = 0, k1 = 0;
        bool nan0 = false, nan1 = false;
#pragma unroll 1
        for (int i = 0; i < WIN; i++)
#pragma unroll 1
            for (int j = 0; j < WIN; j++) {
                const int li = (ty + i) * P + sx + j;
                const bool v0 = s_d[li] > 0.0f, v1 = s_d[li + 1] > 0.0f;
                k0 += v0 ? 1 : 0;
                k1 += v1 ? 1 : 0;
                nan0 |= v0 && k0 >= a.kinf && (s_c[li] & 0x00ffffffu) == cc[0];        // cd == 0
                nan1 |= v1 && k1 >= a.kinf && (s_c[li + 1] & 0x00ffffffu) == cc[1];
            }
        if (flat[0] && wgt.x > 0.0f && nan0) res[0] = __builtin_nanf("");
        if (flat[1] && wgt.y > 0.0f && nan1) res[1] = __builtin_nanf("");
    }
    out_d[p] = res[0];
    if (has1) out_d[p + 1] = res[1];
    KDE_STAGE(if (a.stage_avg) {
        (a.stage_avg + fpx)[p] = wavg.x;
        if (has1) (a.stage_avg + fpx)[p + 1] = wavg.y;
    })
    KDE_STAGE(if (a.stage_dev) {
        if (!stage_dev_done) {        // the product path never needed this pixel's deviation: form it for the dump only
            const e_f2 keep_inv = inv_a, keep_thr = nthr_a;
            const bool keep_flat[2] = {flat[0], flat[1]};
            adaptive_sigma();
            inv_a = keep_inv; nthr_a = keep_thr; flat[0] = keep_flat[0]; flat[1] = keep_flat[1];
        }
        (a.stage_dev + fpx)[p] = stage_dev2[0];
        if (has1) (a.stage_dev + fpx)[p + 1] = stage_dev2[1];
    })
}

}  // namespace

int launch_ers_edge_phase(int width, int height, int dir, int window, const int32_t* color_labels, const int32_t* l0,
                          const float* d0, int32_t* l1, float* d1, hipStream_t s)
{
    dim3 grid(ceil_div(width, 64), ceil_div(height, 4));
    if (dir == 0)
        hipLaunchKernelGGL(edge_phase_kernel<0>, grid, dim3(256), 0, s, width, height, window, color_labels, l0, d0, l1, d1);
    else
        hipLaunchKernelGGL(edge_phase_kernel<1>, grid, dim3(256), 0, s, width, height, window, color_labels, l0, d0, l1, d1);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

// both phases; scratch_l / scratch_d hold the horizontal result only when the window is too wide for the fused kernel
int launch_ers_edge_refining(int width, int height, int n, int window, const int32_t* color_labels, const int32_t* l0,
                             const float* d0, int32_t* scratch_l, float* scratch_d, int32_t* l2, float* d2, bool two_launches,
                             hipStream_t s)
{
    // The fused kernel's halos (colour labels +-(2 + window/2), depth labels / depth -2..+3) and register window are
    // laid out for the reference's window of 7 (EdgeRefinedSuperpixel.cpp:4); any other window runs the two-launch
    // form, whose rule takes any reach.
    const long long tiles = (long long)ceil_div(width, kEdgeTX) * ceil_div(height, kEdgeTY);
    if (!two_launches && window / 2 == 3 && tiles * n <= 0x7fffffffLL) {
        const dim3 grid((unsigned)(tiles * n));          // all frames of the batch in one launch
        hipLaunchKernelGGL(edge_fused_kernel<3>, grid, dim3(256), 0, s, width, height, color_labels, l0, d0, l2, d2);
        KDE_HIP_TRY(hipGetLastError());
        return KDE_OK;
    }
    const size_t px = (size_t)width * height;
    for (int f = 0; f < n; f++) {                        // the two-launch form works frame by frame (one frame of scratch)
        KDE_TRY(launch_ers_edge_phase(width, height, 0, window, color_labels + f * px, l0 + f * px, d0 + f * px, scratch_l, scratch_d, s));
        KDE_TRY(launch_ers_edge_phase(width, height, 1, window, color_labels + f * px, scratch_l, scratch_d, l2 + f * px, d2 + f * px, s));
    }
    return KDE_OK;
}

static int launch_ers_enhance_one(int width, int height, int n, const float* rd, const uint8_t* bgr, const int32_t* labels,
                                  const float* s_eff, const float* table_host, int window, float color_sigma, float depth_sigma,
                                  float exp_zero, float* out, int variant, hipStream_t s);

// n frames back to back: the packed kernel takes the whole batch in one launch, the fall-back kernels go frame by frame
int launch_ers_enhance(int width, int height, int n, const float* rd, const uint8_t* bgr, const int32_t* labels,
                       const float* s_eff, const float* table_host, int window, float color_sigma, float depth_sigma,
                       float exp_zero, float* out, int variant, hipStream_t s)
{
    const float cden = 2 * (color_sigma * color_sigma);
    const bool tuned = window == 7 && color_sigma != 0.0f && depth_sigma != 0.0f && !(195075.0f / cden >= exp_zero);
    const bool can_pk = tuned && (long long)width * height <= (1LL << 24);
    const long long blocks = (long long)ceil_div(width, 64) * ceil_div(height, 8) * n;
    if (n == 1 || (((variant == 0 && can_pk) || variant == 1) && blocks <= 0x7fffffffLL))
        return launch_ers_enhance_one(width, height, n, rd, bgr, labels, s_eff, table_host, window, color_sigma, depth_sigma,
                                      exp_zero, out, variant, s);
    const size_t px = (size_t)width * height;
    for (int f = 0; f < n; f++)
        KDE_TRY(launch_ers_enhance_one(width, height, 1, rd + f * px, bgr + f * px * 3, labels + f * px, s_eff, table_host, window,
                                       color_sigma, depth_sigma, exp_zero, out + f * px, variant, s));
    return KDE_OK;
}

static int launch_ers_enhance_one(int width, int height, int n, const float* rd, const uint8_t* bgr, const int32_t* labels,
                                  const float* s_eff, const float* table_host, int window, float color_sigma, float depth_sigma,
                                  float exp_zero, float* out, int variant, hipStream_t s)
{
    const float cden = 2 * (color_sigma * color_sigma);
    // tuned kernels: 7x7 window, both sigmas on, pass-1 colour factor can never underflow
    const bool tuned = window == 7 && color_sigma != 0.0f && depth_sigma != 0.0f && !(195075.0f / cden >= exp_zero);
    // variant 0 = built-in choice, 1 = packed-pair kernel, 2 = scalar tuned kernel, 3 = generic kernel
    // (the packed kernel compares labels as floats: superpixel indices of a frame of <= 2^24 pixels are exact)
    const bool can_pk = tuned && (long long)width * height <= (1LL << 24);
    if (variant == 1 && !can_pk) return fail(KDE_ERR_INVALID, "ers: the packed kernel does not serve this configuration");
    if (variant == 2 && !tuned) return fail(KDE_ERR_INVALID, "ers: the tuned kernel does not serve this configuration");
    if ((variant == 0 && can_pk) || variant == 1) {
        Enh7PkDev d;
        memset(&d, 0, sizeof(d));
        d.rd = rd; d.bgr = bgr; d.labels = labels; d.out = out; d.width = width; d.height = height;
        const double log2e = 1.4426950408889634;
        d.kc1 = (float)(log2e / (double)cden);
        const float dden = 2.0f * (depth_sigma * depth_sigma);
        d.sd = (float)std::sqrt(log2e / (double)dden);
        {   // smallest q with fl(q / dden) >= x0, as a bound on ((d_q - avg) * sd)^2
            uint32_t lo = 0, hi = 0x7f800000u;
            auto val = [](uint32_t b) { float f; memcpy(&f, &b, 4); return f; };
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (val(mid) / dden >= exp_zero) hi = mid; else lo = mid + 1;
            }
            const double t2 = (double)val(lo) * (log2e / (double)dden);
            d.t2_skip = t2 < 3.0e38 ? (float)t2 : 3.0e38f;
        }
        d.exp_zero = exp_zero;
        auto lg = [&](int i, int j) {   // S == 0 -> factor skipped -> log2 = 0; + the 2^24 scale (kScaleLog2)
            const float sv = table_host[i * 7 + j];
            return (float)((sv == 0.0f ? 0.0 : std::log2((double)sv)) + kScaleLog2);
        };
        for (int i = 0; i < 7; i++)
            for (int u = 0; u < 7; u++) {   // unit -> (tap of p0, tap of p1): see enhance7_pk_kernel
                const int j0 = u <= 3 ? 2 * u : (u < 6 ? 2 * (u - 3) + 1 : 1);
                const int j1 = u <= 3 ? 2 * u : (u < 6 ? 2 * (u - 3) - 1 : 5);
                d.lsp[(i * 7 + u) * 2] = lg(i, j0);
                d.lsp[(i * 7 + u) * 2 + 1] = lg(i, j1);
            }
        // rank table: c_0 = ColorSigma, c_k = c_{k-1} * 0.3f (EdgeRefinedSuperpixel.cu:172-175 while a <= 0.3 c)
        float c = color_sigma;
        d.tinv[0] = 1.0f;    // rank 0 = only invalid taps so far (their weight is multiplied by 0 anyway)
        d.tthr[0] = 0.0f;
        d.kinf = 50;
        d.kfree = 50;
        for (int k = 1; k < 50; k++) {
            c *= 0.3f;
            const float den = 2 * (c * c);
            if (den == 0.0f) {
                if (d.kinf == 50) d.kinf = k;
                d.tinv[k] = 3.0e38f;
                d.tthr[k] = -1.0f;
            } else {
                d.tinv[k] = (float)std::fmin(log2e / (double)den, 3.0e38);
                int lo = 0, hi = 400000;
                while (lo < hi) {
                    const int mid = (lo + hi) / 2;
                    if ((float)mid / den >= exp_zero) hi = mid; else lo = mid + 1;
                }
                d.tthr[k] = -(float)lo;
            }
        }
        // first rank from which every threshold is 1 (the sigmas only shrink with the rank: once 1, always 1)
        for (int k = 49; k >= 1 && d.tthr[k] >= -1.0f; k--) d.kfree = k;
        KDE_STAGE(d.stage_avg = g_stage.ers_avg; d.stage_dev = g_stage.ers_dev; d.stage_counters = g_stage.counters;
                  d.stage_force = g_stage.force_full_rules;)
        const dim3 grid((unsigned)(ceil_div(width, kE7BX * 2) * ceil_div(height, kE7BY) * n));
#ifdef KDE_AB_SWITCHES
        static const bool mask_product = KDE_AB_ENV("KDE_K10_MASK_PRODUCT") != nullptr;  // r03's pass-1 label test (tools/ab_k10.py)
        if (mask_product) hipLaunchKernelGGL(enhance7_pk_kernel<false>, grid, dim3(kE7BX * kE7BY), 0, s, d);
        else
#endif
            hipLaunchKernelGGL(enhance7_pk_kernel<true>, grid, dim3(kE7BX * kE7BY), 0, s, d);
        KDE_HIP_TRY(hipGetLastError());
        return KDE_OK;
    }
    if (tuned && variant != 3) {
        Enh7Dev d;
        memset(&d, 0, sizeof(d));
        d.rd = rd; d.bgr = bgr; d.labels = labels; d.out = out; d.width = width; d.height = height;
        const double log2e = 1.4426950408889634;
        d.kc1 = (float)(log2e / (double)cden);
        const float dden = 2.0f * (depth_sigma * depth_sigma);
        d.sd = (float)std::sqrt(log2e / (double)dden);
        {   // smallest q with fl(q / dden) >= x0, as a bound on |d_q - avg| * sd
            uint32_t lo = 0, hi = 0x7f800000u;
            auto val = [](uint32_t b) { float f; memcpy(&f, &b, 4); return f; };
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2;
                if (val(mid) / dden >= exp_zero) hi = mid; else lo = mid + 1;
            }
            d.t_skip = (float)(std::sqrt((double)val(lo)) * std::sqrt(log2e / (double)dden));
        }
        d.exp_zero = exp_zero;
        for (int i = 0; i < 49; i++)   // S == 0 -> factor skipped -> log2 = 0; + the 2^24 scale (kScaleLog2)
            d.ls[i] = (float)((table_host[i] == 0.0f ? 0.0 : std::log2((double)table_host[i])) + kScaleLog2);
        // rank table: c_0 = ColorSigma, c_k = c_{k-1} * 0.3f (EdgeRefinedSuperpixel.cu:172-175 while a <= 0.3 c)
        float c = color_sigma;
        d.tinv[0] = 1.0f;    // rank 0 = only invalid taps so far: any positive scale keeps -1.7e38 * scale at -huge
        d.tthr[0] = 0.0f;
        for (int k = 1; k < 50; k++) {
            c *= 0.3f;
            const float den = 2 * (c * c);
            if (den == 0.0f) {
                d.tinv[k] = INFINITY;
                d.tthr[k] = -1.0f;
            } else {
                d.tinv[k] = (float)std::fmin(log2e / (double)den, 3.0e38);   // finite while den != 0 (see inv_a)
                int lo = 0, hi = 400000;
                while (lo < hi) {
                    const int mid = (lo + hi) / 2;
                    if ((float)mid / den >= exp_zero) hi = mid; else lo = mid + 1;
                }
                d.tthr[k] = -(float)lo;
            }
        }
        KDE_STAGE(d.stage_avg = g_stage.ers_avg; d.stage_dev = g_stage.ers_dev;)
        hipLaunchKernelGGL(enhance7_kernel, dim3(ceil_div(width, kTileX), ceil_div(height, kTileY)), dim3(kThreads), 0, s, d);
        KDE_HIP_TRY(hipGetLastError());
        return KDE_OK;
    }
    EnhDev d;
    memset(&d, 0, sizeof(d));
    d.rd = rd; d.bgr = bgr; d.labels = labels; d.s_eff = s_eff; d.out = out; d.width = width; d.height = height;
    d.window = window; d.color_sigma = color_sigma; d.depth_sigma = depth_sigma; d.exp_zero = exp_zero;
    KDE_STAGE(d.stage_avg = g_stage.ers_avg; d.stage_dev = g_stage.ers_dev;)
    const int R = window / 2;
    const size_t lds = (size_t)(kTileX + 2 * R) * (kTileY + 2 * R) * 12 + (size_t)window * window * 4;
    hipLaunchKernelGGL(enhance_kernel, dim3(ceil_div(width, kTileX), ceil_div(height, kTileY)), dim3(kThreads), lds, s, d);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

}  // namespace kde
