// dasp_kernels.hip — DepthAdaptiveSuperpixel on gfx950 (K5-K8).
// Reference: SuperpixelSegmentation/DepthAdaptiveSuperpixel.cu:3-588.
//
// Re-architecture vs the reference:
//   K7 calculateLD launches one 16-thread block per PIXEL in the reference (2.07 M blocks at 1080p);
//      here one thread owns one pixel and walks its 16 candidate clusters in registers, with the
//      cluster table (mean + centre, 32 B per cluster) staged in LDS.  The 16-way tree argmin and
//      its tie-break order are kept exactly.
//   K8 analyzeClusters keeps the reference's summation order (per-thread serial, then a 256-way
//      tree) so the float sums agree bit for bit with the CPU restatement; the last six tree levels
//      run as wavefront shuffles.
//   All grids are ceil-div and bounds-guarded (D4); out-of-buffer taps of K6 read colour 0 (D1).
#include "kde_internal.h"
#include "kde_device_math.h"
#include <type_traits>

namespace kde {
namespace {

// ---- K5 init_LD (.cu:3-14): folded into the first calculateLD (calc_ld_kernel<.., FIRST>) --------------

// ---- K6 sampleInitialClusters<16> (.cu:16-165) ---------------------------------------------------
// A workgroup of four wavefronts serves 4 clusters x 16 candidates (one candidate per lane, as the reference's 16-thread
// blocks).  The 121 gradient terms of a candidate are independent; only their SUM has an order (yy outer, xx inner)
// that the float result -- and with it the argmin's ties -- depends on.  So the four wavefronts each evaluate a quarter
// of the terms into LDS, and wavefront 0 then adds them up in the reference's order: the kernel is a latency chain on a
// nearly empty GPU (75 workgroups at 300 clusters), and this cuts the chain from 121 x (distance + sqrt) to 31 x that
// plus 121 additions (11 -> 4 us at 1080p).  The square root's argument is an integer < 2^24: sqrt_int24 is sqrtf there.
// A batch of frames (kde_rgbf_process_batch / kde_spdsr_process_batch): blockIdx.y = frame; every frame has its own
// colour, cloud and cluster table, so a frame's result does not depend on what else is in the launch.
__global__ __launch_bounds__(256) void sample_clusters_kernel(DaspGeom g, const uint8_t* __restrict__ bgr,
                                                             const kde_float3* __restrict__ pts,
                                                             kde_superpixel* __restrict__ mean,
                                                             kde_float3* __restrict__ centers,
                                                             kde_superpixel* __restrict__ mean2,
                                                             kde_float3* __restrict__ centers2)
{
    // (mean2, centers2): a pipeline's second segmenter starts from the same sampled clusters -- written here instead of
    // being copied device-to-device afterwards (two copy launches per frame); null for a single segmenter
    {
        const size_t fpx = (size_t)blockIdx.y * g.width * g.height, fk = (size_t)blockIdx.y * g.rows * g.cols;
        bgr += fpx * 3;
        pts += fpx;
        mean += fk;
        centers += fk;
        if (mean2) {
            mean2 += fk;
            centers2 += fk;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane >> 4, tid = lane & 15;
    const int cluster = blockIdx.x * 4 + sub;
    const int nclusters = g.rows * g.cols;
    const bool active = cluster < nclusters;
    const int bx = active ? cluster % g.cols : 0, by = active ? cluster / g.cols : 0;
    const int tx = tid & 3, ty = tid >> 2;
    const int around_x = bx * g.wx + g.wx / 2 + tx - 2;
    const int around_y = by * g.wy + g.wy / 2 + ty - 2;
    const long npix = (long)g.width * g.height;
    const uint8_t* ca = bgr + ((size_t)around_y * g.width + around_x) * 3;
    const float a0 = (float)ca[0], a1 = (float)ca[1], a2 = (float)ca[2];
    // The 11x11 taps are addressed ABSOLUTELY in the reference (idx = yy*width + xx, .cu:52-54), i.e. they are the
    // same 121 colours for every candidate of every cluster: stage them once per workgroup instead of issuing
    // 363 dependent byte loads per thread.
    __shared__ float taps[121][3];
    __shared__ float grs[121][64];             // [tap][candidate lane]: consecutive lanes, consecutive words
    for (int i = threadIdx.x; i < 121; i += 256) {
        const int yy = i / 11 - 5, xx = i % 11 - 5;
        const long idx = (long)yy * g.width + xx;
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
        if (idx >= 0 && idx < npix) {
            t0 = (float)bgr[idx * 3];
            t1 = (float)bgr[idx * 3 + 1];
            t2 = (float)bgr[idx * 3 + 2];
        }
        taps[i][0] = t0;
        taps[i][1] = t1;
        taps[i][2] = t2;
    }
    __syncthreads();
    for (int i = wv; i < 121; i += 4) {
        const float d0 = a0 - taps[i][0], d1 = a1 - taps[i][1], d2 = a2 - taps[i][2];
        grs[i][lane] = sqrt_int24(d0 * d0 + d1 * d1 + d2 * d2);     // integer-valued, <= 3 * 255^2: exact like sqrtf
    }
    __syncthreads();
    if (wv != 0) return;
    float sumG = 0.0f;
    int count = 0;
    for (int i = 0; i < 121; i++) {               // yy outer, xx inner: the reference's summation order
        const float gr = grs[i][lane];
        count += gr > 0.0f ? 1 : 0;
        sumG += gr;
    }
    float gradient = sumG / (float)count;
    int ax = around_x, ay = around_y;
    // 16-element tree argmin, strict '>' (.cu:106-149); lanes >= step of a sub-group carry garbage
    // that element 0 never reads, exactly like the reference's lock-step warp
#pragma unroll
    for (int step = 8; step >= 1; step >>= 1) {
        const float og = __shfl_down(gradient, step, 16);
        const int ox = __shfl_down(ax, step, 16);
        const int oy = __shfl_down(ay, step, 16);
        if (tid < step && gradient > og) {
            gradient = og;
            ax = ox;
            ay = oy;
        }
    }
    if (active && tid == 0) {
        const int id = by * (g.width / g.wx) + bx;
        const uint8_t* cs = bgr + ((size_t)ay * g.width + ax) * 3;
        kde_superpixel m = mean[id];
        m.x = ax;
        m.y = ay;
        m.r = cs[0];
        m.g = cs[1];
        m.b = (uint8_t)(cs[0] + 2);   // sic, .cu:159
        const kde_float3 c = pts[(size_t)ay * g.width + ax];
        mean[id] = m;
        centers[id] = c;
        if (mean2) {
            mean2[id] = m;
            centers2[id] = c;
        }
    }
}

// ---- K7 calculateLD<16> (.cu:167-313) -------------------------------------------------------------
struct ClusterRec {   // LDS copy of one cluster: 32 B
    float r, g, b;
    float x, y;           // pixel coordinates as floats (exact below 2^24): offsets to a pixel are one v_sub_f32 each
    float cz;
    int xi, yi;           // the same coordinates as the reference holds them: used when one does not fit a float (see big)
};

constexpr int kMaxLdsClusters = 2048;   // 64 KiB of LDS

// One segmenter's view of the assignment step.  RGBF / SPDSR run TWO segmenters (colour-only SP and
// depth-adaptive DASP) on the same colour + cloud: NS = 2 assigns both label maps in one pass over the pixels
// (colour and depth are read once, one launch instead of two); results are identical to two separate launches.
struct CalcSet {
    kde_label_distance* ld;
    const kde_superpixel* mean;
    const kde_float3* centers;
    int32_t* labels;
    float kc, ks, kd;
    int depth_on;
};
template <int NS>
struct CalcSets {
    CalcSet s[NS];
    int write_ld;      // 0: the (distance, label) records are not stored (a pipeline with ONE assignment step never reads them)
};

// FIRST = the assignment step that follows sampleInitialClusters: init_LD (K5) is folded in -- the previous
// assignment is "own grid cell at distance 999999.9" and is formed in registers instead of being written by one
// kernel and read back by the next -- and all NS segmenters still share the sampled clusters, so the colour /
// spatial / depth distances of the 16 candidates are computed ONCE and only the weighted sum and the argmin run
// per segmenter (set 0's cluster table serves all).
struct CalcDivs {          // divisions of the assignment step by multiplication (kde_device_math.h)
    FastDiv24 wx, wy, cols;
};

// "take = d < best; best = take ? d : best; label = take ? id : label" compiles to one v_cmp and TWO back-to-back
// v_cndmask_b32 in the VOP2 encoding (condition implicitly in VCC) -- and on gfx950 the second of two adjacent VOP2 v_cndmask
// stalls the SIMD for ~20 cycles (tools/valu_microbench: "1 v_cmp + 2 v_cndmask (vcc)" 3.2-6.8 ns per instruction against
// 1.3 with one other VALU instruction between them and 1.7 in the VOP3 encoding; profiles/r04_valu_microbench_vcc.txt).
// The argmin update therefore keeps its condition in an SGPR pair and selects with the VOP3 form: same results, no stall.
// (A 64-bit SGPR pair is the condition of a 64-wide wavefront: this library is built for gfx950 only, see the Makefile.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "dasp_kernels.hip: the SGPR-pair conditions below assume gfx950 (wave64); build with --offload-arch=gfx950"
#endif
__device__ __forceinline__ uint64_t lt_mask(float x, float y)
{
    uint64_t m;
    asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(x), "v"(y));
    return m;
}
__device__ __forceinline__ float sel_f(uint64_t m, float a, float b)       // m ? a : b
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ __forceinline__ int sel_i(uint64_t m, int a, int b)
{
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}

template <int NS, bool USE_LDS, bool FIRST>
__global__ __launch_bounds__(256) void calc_ld_kernel(DaspGeom g, const uint8_t* __restrict__ bgr,
                                                     const kde_float3* __restrict__ pts, CalcSets<NS> sets, float win2,
                                                     CalcDivs dv)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ClusterRec* recs = reinterpret_cast<ClusterRec*>(smem);
    const int nclusters = g.rows * g.cols;
    constexpr int NTAB = FIRST ? 1 : NS;
    {   // blockIdx.z = frame of a batch: per-frame colour, cloud, cluster tables and outputs
        const size_t fpx = (size_t)blockIdx.z * g.width * g.height, fk = (size_t)blockIdx.z * nclusters;
        bgr += fpx * 3;
        pts += fpx;
#pragma unroll
        for (int n = 0; n < NS; n++) {
            sets.s[n].ld += fpx;
            sets.s[n].labels += fpx;
            sets.s[n].mean += fk;
            sets.s[n].centers += fk;
        }
    }
    // analyzeClusters keeps a projected centre whose row lies BELOW the image (its test reads pixel.y <= height, .cu:549),
    // so mean.y can be any int above the height, INT_MAX included.  (float)(y - mean.y), which the reference forms
    // (.cu:213), equals (float)y - (float)mean.y only while both fit a float exactly: a table that holds such a
    // coordinate (|v| >= 2^24) switches the workgroup to the integer subtraction and the library sqrtf.
    __shared__ int s_big;
    bool big = !USE_LDS;              // the table-in-global form always subtracts integers
    if (USE_LDS) {
        if (threadIdx.x == 0) s_big = 0;
        __syncthreads();
        bool mine = false;
#pragma unroll
        for (int n = 0; n < NTAB; n++)
            for (int i = threadIdx.x; i < nclusters; i += 256) {
                const kde_superpixel m = sets.s[n].mean[i];
                ClusterRec r;
                r.r = (float)m.r;
                r.g = (float)m.g;
                r.b = (float)m.b;
                r.x = (float)m.x;
                r.y = (float)m.y;
                r.cz = sets.s[n].centers[i].z;
                r.xi = m.x;
                r.yi = m.y;
                mine |= m.x >= (1 << 24) || m.x <= -(1 << 24) || m.y >= (1 << 24) || m.y <= -(1 << 24);
                recs[n * nclusters + i] = r;
            }
        if (mine) s_big = 1;
        __syncthreads();
        big = s_big != 0;             // workgroup-uniform
    }
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.width || y >= g.height) return;
    const size_t p = (size_t)y * g.width + x;
    const float c0 = (float)bgr[p * 3], c1 = (float)bgr[p * 3 + 1], c2 = (float)bgr[p * 3 + 2];
    const float z = pts[p].z;
    const float xf = (float)x, yf = (float)y;

    // the three distances of candidate t against cluster table n (.cu:196-217)
    auto candidate = [&](int n, int id, float& color_distance, float& spatial_distance, float& depth_distance) {
        float mr, mg, mb, cz, mx, my;
        int mxi, myi;
        if (USE_LDS) {
            const ClusterRec r = recs[n * nclusters + id];
            mr = r.r; mg = r.g; mb = r.b; mx = r.x; my = r.y; cz = r.cz; mxi = r.xi; myi = r.yi;
        } else {
            const kde_superpixel m = sets.s[n].mean[id];
            mr = (float)m.r; mg = (float)m.g; mb = (float)m.b; mx = (float)m.x; my = (float)m.y; mxi = m.x; myi = m.y;
            cz = sets.s[n].centers[id].z;
        }
        const float e0 = c0 - mr, e1 = c1 - mg, e2 = c2 - mb;
        color_distance = e0 * e0 + e1 * e1 + e2 * e2;
        if (big) {
            // (x - mean.x wraps for mean.x near INT_MIN exactly as the reference's int subtraction does on the GPU)
            const float px = (float)(int)((unsigned)x - (unsigned)mxi), py = (float)(int)((unsigned)y - (unsigned)myi);
            spatial_distance = sqrtf(px * px + py * py) * win2;
        } else {
            const float px = xf - mx, py = yf - my;       // = (float)(x - mx): integers below 2^24
            spatial_distance = sqrt_int24(px * px + py * py) * win2;
        }
        depth_distance = 0.0f;
        if (z > 50.0f && cz > 50.0f) depth_distance = fabsf(z - cz);
    };
    // The reference reduces the 16 candidate distances with a tree of strict '>' comparisons (.cu:226-305): the
    // minimum wins, and among equal minima the one whose index is first in BIT-REVERSED order (0, 8, 4, 12, 2, ...).
    // A scan in that order with a strict '<' is the same function -- and lets candidates be skipped: the pixel's own
    // grid cell (candidate 10) is evaluated first, and a cluster whose spatial term ALONE exceeds that distance (with
    // a 1e-5 margin; colour and depth terms are >= 0) can neither win nor tie.  The skip is taken when no lane of
    // the wavefront needs the candidate; typically 5 of 16 are evaluated.  Segmenters lo..hi share table `tab` and
    // the previous assignment `cur` (all of them in the FIRST step, one at a time later).
    // inner = every lane's 4 x 4 candidate block lies inside the cluster grid (wavefront-uniform, std::true_type /
    // std::false_type): no candidate can be "outside the grid", so its tests and selects are not even compiled in
    auto assign_impl = [&](auto inner, int tab, int lo, int hi, const kde_label_distance cur, int ccx, int ccy) {
        constexpr bool INNER = decltype(inner)::value;
        auto grid_id = [&](int t) {
            const int rx = ccx - 2 + (t & 3), ry = ccy - 2 + (t >> 2);
            if constexpr (INNER) return ry * g.cols + rx;
            else return (rx >= 0 && rx < g.cols && ry >= 0 && ry < g.rows) ? ry * g.cols + rx : -1;
        };
        float best[NS], thr[NS], own_d[NS];
        int bl[NS];
        const int own_id = grid_id(10);
        {
            float cdv = 0.0f, sdv = 0.0f, ddv = 0.0f;
            candidate(tab, own_id >= 0 ? own_id : 0, cdv, sdv, ddv);
#pragma unroll
            for (int n = 0; n < NS; n++) {
                if (n < lo || n > hi) continue;
                const CalcSet& cs = sets.s[n];
                own_d[n] = cdv * cs.kc + sdv * cs.ks + ddv * cs.kd;                  // .cu:218
                // spatial term of a candidate = sqrt(n2) * win2 * ks > own_d * (1 + 1e-5)  <=>  n2 > thr
                const float q = own_d[n] * 1.00001f / (win2 * cs.ks);
                thr[n] = (own_id >= 0 && cs.ks > 0.0f && own_d[n] >= 0.0f) ? q * q : INFINITY;
                best[n] = INFINITY;
                bl[n] = cur.l;
            }
        }
        float thr_any = -INFINITY;                    // the largest of the thresholds (NaN-free: thr is q*q or +inf)
#pragma unroll
        for (int n = 0; n < NS; n++)
            if (n >= lo && n <= hi) thr_any = fmaxf(thr_any, thr[n]);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int t = ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3);   // bit-reversed scan
            const int id = grid_id(t);
            if (t == 10 || (!INNER && id < 0)) {
#pragma unroll
                for (int n = 0; n < NS; n++) {
                    if (n < lo || n > hi) continue;
                    const float d = (!INNER && id < 0) ? cur.d : own_d[n];        // .cu:221-224 outside the grid
                    const int l = (!INNER && id < 0) ? cur.l : id;
                    const uint64_t take = lt_mask(d, best[n]);
                    best[n] = sel_f(take, d, best[n]);
                    bl[n] = sel_i(take, l, bl[n]);
                }
                continue;
            }
            float px, py;
            if (USE_LDS && !big) {
                px = xf - recs[tab * nclusters + id].x;
                py = yf - recs[tab * nclusters + id].y;
            } else {
                const int mxi = USE_LDS ? recs[tab * nclusters + id].xi : sets.s[tab].mean[id].x;
                const int myi = USE_LDS ? recs[tab * nclusters + id].yi : sets.s[tab].mean[id].y;
                px = (float)(int)((unsigned)x - (unsigned)mxi);
                py = (float)(int)((unsigned)y - (unsigned)myi);
            }
            const float n2 = px * px + py * py;
            const bool need = !(n2 > thr_any);       // some segmenter of this pass may still need the candidate
            if (__builtin_amdgcn_ballot_w64(need) == 0) continue;              // nobody in the wavefront needs it
            float cdv, sdv, ddv;
            candidate(tab, id, cdv, sdv, ddv);
#pragma unroll
            for (int n = 0; n < NS; n++) {
                if (n < lo || n > hi) continue;
                const CalcSet& cs = sets.s[n];
                const float d = cdv * cs.kc + sdv * cs.ks + ddv * cs.kd;          // .cu:218
                const uint64_t take = lt_mask(d, best[n]);
                best[n] = sel_f(take, d, best[n]);
                bl[n] = sel_i(take, id, bl[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < NS; n++) {
            if (n < lo || n > hi) continue;
            const CalcSet& cs = sets.s[n];
            kde_label_distance o;
            o.l = bl[n];
            o.d = best[n];
            if (z < 50.0f && cs.depth_on) {   // .cu:308-312
                o.l = -1;
                o.d = 0.0f;
            }
            if (sets.write_ld) cs.ld[p] = o;      // (a kernel argument: uniform; the last step of a pipeline stores no records)
            cs.labels[p] = o.l;
        }
    };
    // (ccx, ccy) = the grid cell of the previous assignment: cur.l % cols, cur.l / cols
    auto assign = [&](int tab, int lo, int hi, const kde_label_distance cur, int ccx, int ccy) {
        const bool in = cur.l >= 0 && ccx >= 2 && ccx + 1 < g.cols && ccy >= 2 && ccy + 1 < g.rows;
        if (__builtin_amdgcn_ballot_w64(!in) == 0) assign_impl(std::true_type{}, tab, lo, hi, cur, ccx, ccy);
        else assign_impl(std::false_type{}, tab, lo, hi, cur, ccx, ccy);
    };

    if (FIRST) {
        kde_label_distance cur;                                   // init_LD (.cu:3-14)
        const int cx = (int)fastdiv24((uint32_t)x, dv.wx), cy = (int)fastdiv24((uint32_t)y, dv.wy);
        cur.l = cy * g.cols + cx;
        cur.d = 999999.9f;
        // (x / wx may reach cols when the width is not a multiple of the cell: cur.l % cols, cur.l / cols then wrap
        // into the next row, as the reference's arithmetic does)
        const bool wrap = cx >= g.cols;
        assign(0, 0, NS - 1, cur, wrap ? cur.l % g.cols : cx, wrap ? cur.l / g.cols : cy);
        return;
    }
#pragma unroll
    for (int n = 0; n < NS; n++) {
        const kde_label_distance cur = sets.s[n].ld[p];
        int ccx, ccy;
        if (cur.l >= 0) {
            ccy = (int)fastdiv24((uint32_t)cur.l, dv.cols);
            ccx = cur.l - ccy * g.cols;
        } else {                                                  // -1 (unassigned): C's truncating % and /
            ccx = cur.l % g.cols;
            ccy = cur.l / g.cols;
        }
        assign(n, n, n, cur, ccx, ccy);
    }
}

// ---- K8 analyzeClusters<256> (.cu:315-568) ---------------------------------------------------------
__device__ __forceinline__ int f2i_rz(float v) { return (int)v; }   // v_cvt_i32_f32: RZ, saturating, NaN -> 0

// One segmenter's view of analyzeClusters; blockIdx.z picks the segmenter (a pipeline updates both in one launch).
// `labels` is the label map calculateLD writes next to its (distance, label) records -- the same labels at 4 bytes
// per pixel, so one 16-byte access fetches four of them.
struct AnalyzeSet {
    const int32_t* labels;
    kde_superpixel* mean;
    kde_float3* centers;
};
struct AnalyzeSets {
    AnalyzeSet s[2];
    int nsets;          // grid = frames x nsets x clusters
    int band;
};

// The tail of analyzeClusters (.cu:425-566): the reference's 256-way tree over the threads' partial sums (levels +128, +64
// through LDS, +32 ... +1 inside the first wavefront) and the new cluster record.  si / sf: 7 x 256 ints and 3 x 256 floats of LDS.
__device__ __forceinline__ void analyze_reduce_and_store(const DaspGeom& g, int tid, int cluster_id, int (*si)[256], float (*sf)[256],
                                                         int r_, int g_, int b_, int x_, int y_, int s_, int n_, float xf, float yf,
                                                         float zf, kde_superpixel* __restrict__ mean, kde_float3* __restrict__ centers,
                                                         const float* __restrict__ intr)
{
    si[0][tid] = r_; si[1][tid] = g_; si[2][tid] = b_; si[3][tid] = x_; si[4][tid] = y_;
    si[5][tid] = s_; si[6][tid] = n_;
    sf[0][tid] = xf; sf[1][tid] = yf; sf[2][tid] = zf;
    __syncthreads();
    // tree levels +128, +64 through LDS (.cu:425-454)
    if (tid < 128) {
#pragma unroll
        for (int k = 0; k < 7; k++) si[k][tid] += si[k][tid + 128];
#pragma unroll
        for (int k = 0; k < 3; k++) sf[k][tid] += sf[k][tid + 128];
    }
    __syncthreads();
    if (tid < 64) {
        int iv[7];
        float fv[3];
#pragma unroll
        for (int k = 0; k < 7; k++) iv[k] = si[k][tid] + si[k][tid + 64];
#pragma unroll
        for (int k = 0; k < 3; k++) fv[k] = sf[k][tid] + sf[k][tid + 64];
        // levels +32 ... +1 inside the wavefront (.cu:455-528); lane 0 sees the clean tree
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
#pragma unroll
            for (int k = 0; k < 7; k++) iv[k] += __shfl_down(iv[k], step, 64);
#pragma unroll
            for (int k = 0; k < 3; k++) fv[k] += __shfl_down(fv[k], step, 64);
        }
        if (tid == 0 && iv[5] != 0) {   // .cu:530-566
            const int size = iv[5], np = iv[6];
            int r = iv[0] / size > 255 ? 255 : iv[0] / size;
            int gg = iv[1] / size > 255 ? 255 : iv[1] / size;
            int b = iv[2] / size > 255 ? 255 : iv[2] / size;
            r = r < 0 ? 0 : r;
            gg = gg < 0 ? 0 : gg;
            b = b < 0 ? 0 : b;
            int pix_x, pix_y;
            if (np != 0) {
                kde_float3 c;
                c.x = fv[0] / (float)np;
                c.y = fv[1] / (float)np;
                c.z = fv[2] / (float)np;
                centers[cluster_id] = c;
                const float nx = c.x / c.z, ny = c.y / c.z;
                pix_x = f2i_rz(nx * intr[0] + intr[2]);
                pix_y = f2i_rz(intr[5] - ny * intr[4]);
                if (pix_x < 0 || pix_x >= g.width || pix_y < 0 || pix_y <= g.height) {   // sic, .cu:549
                    pix_x = iv[3] / size;
                    pix_y = iv[4] / size;
                }
            } else {
                pix_x = iv[3] / size;
                pix_y = iv[4] / size;
            }
            kde_superpixel m;
            m.r = (uint8_t)r;
            m.g = (uint8_t)gg;
            m.b = (uint8_t)b;
            m.pad_ = 0;
            m.x = pix_x;
            m.y = pix_y;
            m.size = size;
            mean[cluster_id] = m;
        }
    }
}

__global__ __launch_bounds__(256) void analyze_clusters_kernel(DaspGeom g, const uint8_t* __restrict__ bgr,
                                                              const kde_float3* __restrict__ pts, AnalyzeSets sets,
                                                              const float* __restrict__ intr)
{
    // 1-D grid of (frame, segmenter, cluster row, cluster column) walked in XCD bands: a cluster scans a 2w x 2h window, i.e.
    // it overlaps its eight neighbours' windows by half -- neighbours now share one XCD's L2 (KDE_K8_NO_BAND_WALK: A/B switch)
    const unsigned ncl = (unsigned)(g.rows * g.cols);
    const unsigned gid = sets.band ? xcd_band_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const unsigned zz = gid / ncl, cid = gid - zz * ncl;
    const unsigned frame = zz / (unsigned)sets.nsets, seg = zz - frame * (unsigned)sets.nsets;
    const size_t fpx = (size_t)frame * g.width * g.height, fk = (size_t)frame * g.rows * g.cols;
    bgr += fpx * 3;
    pts += fpx;
    const int32_t* __restrict__ labels = sets.s[seg].labels + fpx;
    kde_superpixel* __restrict__ mean = sets.s[seg].mean + fk;
    kde_float3* __restrict__ centers = sets.s[seg].centers + fk;
    __shared__ int si[7][256];     // r g b x y size npoints
    __shared__ float sf[3][256];   // X Y Z
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int cluster_id = (int)cid;
    const int rpx = g.wx * 2 / 16 + 1, rpy = g.wy * 2 / 16 + 1;
    const kde_superpixel m0 = mean[cluster_id];

    int r_ = 0, g_ = 0, b_ = 0, x_ = 0, y_ = 0, s_ = 0, n_ = 0;
    float xf = 0.0f, yf = 0.0f, zf = 0.0f;
    // The thread's rpx x rpy sub-window is walked in the reference's order (yy outer, xx inner; the float sums
    // depend on it), but in chunks of CH positions whose loads are issued together: first the CH labels, then
    // colour + point of the positions that belong to this cluster, then the accumulation: 2 load round trips per
    // chunk instead of 2 per pixel.  (Staging the label test through LDS with coalesced row reads was measured
    // slower: the kernel is bound by the strided gathers of the matched pixels, one cache line per lane.)
    constexpr int CH = 8;
    const size_t last_pix = (size_t)g.width * g.height - 1;    // its 4-byte colour read would leave the buffer
    // 32-bit wrapping adds, as the reference's: a centre kept below the image (y up to INT_MAX) scans rows that wrap to negative
    const int ax0 = (int)((unsigned)m0.x + (unsigned)((tx - 8) * rpx)), ay0 = (int)((unsigned)m0.y + (unsigned)((ty - 8) * rpy));
    // rows of the sub-window in the reference's order, CH positions at a time; labels are fetched four positions per
    // 16-byte load (the kernel is bound by the NUMBER of per-lane cache-line accesses of its strided gathers, so
    // bytes per access is what counts)
    for (int yy = 0; yy < rpy; yy++) {
      const int ary = (int)((unsigned)ay0 + (unsigned)yy);
      const bool row_in = ary >= 0 && ary < g.height;
      for (int xc = 0; xc < rpx; xc += CH) {
        bool hit[CH];
        size_t q[CH];
        int px_[CH], py_[CH];
#pragma unroll
        for (int k = 0; k < CH; k += 4) {
            const int arx = (int)((unsigned)ax0 + (unsigned)(xc + k));
            bool in[4];
            bool all_in = true;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                in[j] = row_in && xc + k + j < rpx && arx + j >= 0 && arx + j < g.width;
                all_in = all_in && in[j];
                q[k + j] = in[j] ? (size_t)ary * g.width + arx + j : 0;
                px_[k + j] = arx + j;
                py_[k + j] = ary;
            }
            int l[4] = {-2, -2, -2, -2};               // never a cluster id
            if (all_in) {
                int4 w;
                __builtin_memcpy(&w, &labels[q[k]], 16);
                l[0] = w.x; l[1] = w.y; l[2] = w.z; l[3] = w.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (in[j]) l[j] = labels[q[k + j]];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) hit[k + j] = l[j] == cluster_id;
        }
        uint32_t col[CH];
        kde_float3 pt[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            col[k] = 0;
            pt[k] = kde_float3{0.0f, 0.0f, 0.0f};
            if (hit[k]) {
                if (q[k] < last_pix) {
                    __builtin_memcpy(&col[k], bgr + q[k] * 3, 4);            // one (unaligned) dword load
                } else {
                    col[k] = (uint32_t)bgr[q[k] * 3] | ((uint32_t)bgr[q[k] * 3 + 1] << 8) | ((uint32_t)bgr[q[k] * 3 + 2] << 16);
                }
                pt[k] = pts[q[k]];
            }
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            if (hit[k]) {
                r_ += (int)(col[k] & 0xffu);
                g_ += (int)((col[k] >> 8) & 0xffu);
                b_ += (int)((col[k] >> 16) & 0xffu);
                x_ += px_[k];
                y_ += py_[k];
                s_ += 1;
                xf += pt[k].x;
                yf += pt[k].y;
                zf += pt[k].z;
                n_ += pt[k].z > 50.0f ? 1 : 0;
            }
        }
      }
    }
    analyze_reduce_and_store(g, tid, cluster_id, si, sf, r_, g_, b_, x_, y_, s_, n_, xf, yf, zf, mean, centers, intr);
}

#ifdef KDE_AB_SWITCHES
// ---- K8, row-coalesced form (r05) --------------------------------------------------------------------------------------
// The kernel above is bound by the request queue of the L1 / texture-address path: a thread's sub-window is rpx pixels wide,
// so at every step the 16 threads of a row touch pixels rpx apart -- 16 different cache lines per quarter-wave for the points
// (12 B at a stride of 12 x rpx), once per MATCHED pixel and load.  But the 16 sub-windows of one thread row are one
// contiguous run of 16 x rpx pixels of one image row, and only the float sums X, Y, Z depend on who adds what in which
// order (per thread serially, then the 256-way tree); the integer sums r, g, b, x, y, size are associative.  So per
// wavefront (4 thread rows) and sub-window row yy:
//   loader phase -- the 64 lanes walk the 4 row segments in 4-pixel chunks, lane after lane (a chunk's 4 labels are one
//     16-byte load, 64 consecutive chunks one contiguous kilobyte); a lane whose chunk holds a pixel of this cluster loads the
//     chunk's 12 colour bytes and 48 point bytes (contiguous across lanes as well), adds the matched pixels to ITS integer
//     sums, and leaves the 4 match flags and the 4 points in the wavefront's LDS stage;
//   chain phase  -- every lane walks its own rpx pixels of the row in the reference's order and adds the staged points of the
//     matched ones to its float sums (and counts z > 50).
// The float chains are the reference's, add for add; the tree and the record are shared with the kernel above.
// The stage is private to a wavefront (LDS operations of one wavefront execute in order: no workgroup barrier inside the loop).
//
// MEASURED AND NOT USED (measurement build only, KDE_K8_ROWS=1; tools/k8_rows_check.py, tools/bench_spdsr.py): labels, cluster
// records and float centres are bit-identical to the kernel above in every configuration tried, but it is SLOWER -- per launch
// in SPDSR at 1080p: 82 us in its first form (one chunk after the other: four dependent label -> point round trips per row),
// 71 us with all loads of a row issued together and the labels one row ahead (130 VGPRs, 3 workgroups per CU by its 43 KB
// stage), 88 us with the chain's LDS reads batched as well (197 VGPRs, 2 per CU) -- against 47.5 us.  The request-bound
// kernel keeps 8 workgroups per CU and hundreds of independent gathers in flight; the staged form trades that for fewer,
// wider requests and two dependent phases per row, and loses.  What would pay is reading each label once instead of 4.7
// times (the windows of neighbouring clusters overlap by half), which needs several clusters per workgroup.
__global__ __launch_bounds__(256) void analyze_clusters_rows_kernel(DaspGeom g, const uint8_t* __restrict__ bgr,
                                                                   const kde_float3* __restrict__ pts, AnalyzeSets sets,
                                                                   const float* __restrict__ intr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char k8_lds[];
    const unsigned ncl = (unsigned)(g.rows * g.cols);
    const unsigned gid = sets.band ? xcd_band_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const unsigned zz = gid / ncl, cid = gid - zz * ncl;
    const unsigned frame = zz / (unsigned)sets.nsets, seg_i = zz - frame * (unsigned)sets.nsets;
    const size_t fpx = (size_t)frame * g.width * g.height, fk = (size_t)frame * g.rows * g.cols;
    bgr += fpx * 3;
    pts += fpx;
    const int32_t* __restrict__ labels = sets.s[seg_i].labels + fpx;
    kde_superpixel* __restrict__ mean = sets.s[seg_i].mean + fk;
    kde_float3* __restrict__ centers = sets.s[seg_i].centers + fk;
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4, tyl = ty & 3, wave = tid >> 6, ln = tid & 63;
    const int cluster_id = (int)cid;
    const int rpx = g.wx * 2 / 16 + 1, rpy = g.wy * 2 / 16 + 1;
    const int RW = 16 * rpx;                          // pixels of a row segment (all 16 sub-windows of a thread row)
    const int CPS = RW / 4, NCH = 4 * CPS;            // 4-pixel chunks per segment / per wavefront step
    const kde_superpixel m0 = mean[cluster_id];
    // this wavefront's stage: 4 segments x RW match flags (bytes), then 4 x RW points (float3)
    unsigned char* mt = k8_lds + (size_t)wave * ((size_t)4 * RW * 13);
    float* pp = reinterpret_cast<float*>(mt + 4 * RW);
    const int xs = (int)((unsigned)m0.x - (unsigned)(8 * rpx));          // 32-bit wrapping adds, as the reference's
    const size_t last_pix = (size_t)g.width * g.height - 1;

    int r_ = 0, g_ = 0, b_ = 0, x_ = 0, y_ = 0, s_ = 0, n_ = 0;
    float xf = 0.0f, yf = 0.0f, zf = 0.0f;
    // this lane's chunks (at most kMaxC: 16 rpx x 4 / 4 <= 256 chunks per step), fixed for the whole window: segment, first
    // pixel, image column, which of the 4 columns lie inside the image, and the image row of sub-window row 0
    constexpr int kMaxC = 4;
    int c_off[kMaxC], c_ax[kMaxC], c_row0[kMaxC];
    unsigned c_inx[kMaxC];                            // bit j: column ax + j is inside the image; bit 4: the chunk exists
#pragma unroll
    for (int k = 0; k < kMaxC; k++) {
        const int c = ln + 64 * k;
        const int sg = c / CPS, px0 = (c - sg * CPS) * 4;
        c_off[k] = sg * RW + px0;
        c_ax[k] = (int)((unsigned)xs + (unsigned)px0);
        c_row0[k] = (int)((unsigned)m0.y + (unsigned)((4 * wave + sg - 8) * rpy));
        c_inx[k] = c < NCH ? 16u : 0u;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (c_ax[k] + j >= 0 && c_ax[k] + j < g.width) c_inx[k] |= 1u << j;
    }
    // the labels of one sub-window row: issued for ALL chunks of the lane at once, and one row ahead of their use
    auto load_labels = [&](int yy, int4 (&lab)[kMaxC]) {
#pragma unroll
        for (int k = 0; k < kMaxC; k++) {
            lab[k] = make_int4(-2, -2, -2, -2);                          // never a cluster id
            const int row = (int)((unsigned)c_row0[k] + (unsigned)yy);
            if (!(c_inx[k] & 16u) || row < 0 || row >= g.height) continue;
            const size_t q = (size_t)row * g.width + c_ax[k];
            if ((c_inx[k] & 15u) == 15u) {
                __builtin_memcpy(&lab[k], &labels[q], 16);
            } else {
                if (c_inx[k] & 1u) lab[k].x = labels[q];
                if (c_inx[k] & 2u) lab[k].y = labels[q + 1];
                if (c_inx[k] & 4u) lab[k].z = labels[q + 2];
                if (c_inx[k] & 8u) lab[k].w = labels[q + 3];
            }
        }
    };
    int4 lab[kMaxC];
    load_labels(0, lab);
    for (int yy = 0; yy < rpy; yy++) {
        // ---- loader phase: flags of every chunk, then the colour / point loads of ALL matched chunks before any is consumed ----
        uint32_t flags[kMaxC], cb[kMaxC][3];
        float pf[kMaxC][12];
#pragma unroll
        for (int k = 0; k < kMaxC; k++) {
            flags[k] = (lab[k].x == cluster_id ? 1u : 0u) | (lab[k].y == cluster_id ? 1u << 8 : 0u) | (lab[k].z == cluster_id ? 1u << 16 : 0u) |
                       (lab[k].w == cluster_id ? 1u << 24 : 0u);
            if (c_inx[k] & 16u) *reinterpret_cast<uint32_t*>(mt + c_off[k]) = flags[k];
            cb[k][0] = cb[k][1] = cb[k][2] = 0u;
#pragma unroll
            for (int i = 0; i < 12; i++) pf[k][i] = 0.0f;
            if (!flags[k]) continue;
            const int row = (int)((unsigned)c_row0[k] + (unsigned)yy);
            const size_t q0 = (size_t)row * g.width + c_ax[k];
            if ((c_inx[k] & 15u) == 15u) {
                __builtin_memcpy(cb[k], bgr + q0 * 3, 12);                                   // 4 packed BGR pixels
                __builtin_memcpy(pf[k], &pts[q0], 48);                                       // 4 points
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((flags[k] >> (8 * j)) & 1u) {
                        const size_t q = q0 + j;
                        const uint32_t v = (uint32_t)bgr[q * 3] | ((uint32_t)bgr[q * 3 + 1] << 8) | ((uint32_t)bgr[q * 3 + 2] << 16);
                        // the same packed layout as the 12-byte load gives
                        if (j == 0) cb[k][0] |= v;
                        else if (j == 1) { cb[k][0] |= v << 24; cb[k][1] |= v >> 8; }
                        else if (j == 2) { cb[k][1] |= v << 16; cb[k][2] |= v >> 16; }
                        else cb[k][2] |= v << 8;
                        const kde_float3 t = pts[q];
                        pf[k][3 * j] = t.x; pf[k][3 * j + 1] = t.y; pf[k][3 * j + 2] = t.z;
                    }
            }
        }
        int4 nxt[kMaxC];
        if (yy + 1 < rpy) load_labels(yy + 1, nxt);                   // in flight while this row is summed
#pragma unroll
        for (int k = 0; k < kMaxC; k++) {
            if (!flags[k]) continue;
            const int row = (int)((unsigned)c_row0[k] + (unsigned)yy);
            const uint32_t px[4] = {cb[k][0] & 0xffffffu, (cb[k][0] >> 24) | ((cb[k][1] & 0xffffu) << 8),
                                    (cb[k][1] >> 16) | ((cb[k][2] & 0xffu) << 16), cb[k][2] >> 8};
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((flags[k] >> (8 * j)) & 1u) {
                    r_ += (int)(px[j] & 0xffu);
                    g_ += (int)((px[j] >> 8) & 0xffu);
                    b_ += (int)((px[j] >> 16) & 0xffu);
                    x_ += c_ax[k] + j;
                    y_ += row;
                    s_ += 1;
                }
            float4* dst = reinterpret_cast<float4*>(pp + (size_t)c_off[k] * 3);
            dst[0] = make_float4(pf[k][0], pf[k][1], pf[k][2], pf[k][3]);
            dst[1] = make_float4(pf[k][4], pf[k][5], pf[k][6], pf[k][7]);
            dst[2] = make_float4(pf[k][8], pf[k][9], pf[k][10], pf[k][11]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the stage is written; the chains below read other lanes' chunks
        __builtin_amdgcn_wave_barrier();
        // ---- chain phase: this thread's rpx pixels of the row, in the reference's order ----
        {
            const int base = tyl * RW + tx * rpx;
            for (int xx = 0; xx < rpx; xx++) {
                if (mt[base + xx]) {
                    const float* P = pp + (size_t)(base + xx) * 3;
                    xf += P[0];
                    yf += P[1];
                    zf += P[2];
                    n_ += P[2] > 50.0f ? 1 : 0;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the next row overwrites the stage
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < kMaxC; k++) lab[k] = nxt[k];
    }
    (void)last_pix;
    __syncthreads();                                                  // every wavefront is done with its stage: reuse it for the tree
    int (*si)[256] = reinterpret_cast<int (*)[256]>(k8_lds);
    float (*sf)[256] = reinterpret_cast<float (*)[256]>(k8_lds + 7 * 256 * sizeof(int));
    analyze_reduce_and_store(g, tid, cluster_id, si, sf, r_, g_, b_, x_, y_, s_, n_, xf, yf, zf, mean, centers, intr);
}

// LDS of the row-coalesced form: 4 wavefronts x 4 segments x 16 rpx pixels x (1 flag byte + 12 point bytes), at least the
// 10 KB of the tree; 0 = the geometry does not fit (the request-bound kernel serves it)
static size_t analyze_rows_lds(const DaspGeom& g)
{
    const int rpx = g.wx * 2 / 16 + 1;
    const size_t need = (size_t)4 * 4 * 16 * rpx * 13;
    const size_t tree = 7 * 256 * sizeof(int) + 3 * 256 * sizeof(float);
    return need <= 48 * 1024 && rpx <= 16 ? (need > tree ? need : tree) : 0;     // (rpx <= 16: at most 4 chunks per lane and row)
}
#endif  // KDE_AB_SWITCHES (row-coalesced K8)

}  // namespace

int launch_dasp_sample(const DaspGeom& g, int n, const uint8_t* bgr, const kde_float3* pts, kde_superpixel* mean,
                       kde_float3* centers, kde_superpixel* mean2, kde_float3* centers2, hipStream_t s)
{
    hipLaunchKernelGGL(sample_clusters_kernel, dim3(ceil_div(g.rows * g.cols, 4), n), dim3(256), 0, s, g, bgr, pts, mean,
                       centers, mean2, centers2);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

static CalcSet make_set(const DaspGeom& g, kde_label_distance* ld, const kde_superpixel* mean, const kde_float3* centers,
                        int32_t* labels, float color_sigma, float spatial_sigma, float depth_sigma)
{
    // host-side scalars formed exactly as .cu:209-218 does per thread
    const float sum_sigma = spatial_sigma + color_sigma + depth_sigma;
    const float rc = color_sigma / sum_sigma, rs = spatial_sigma / sum_sigma, rd = depth_sigma / sum_sigma;
    CalcSet c;
    c.ld = ld; c.mean = mean; c.centers = centers; c.labels = labels;
    c.kc = rc * rc; c.ks = rs * rs; c.kd = rd * rd;
    c.depth_on = depth_sigma != 0.0f ? 1 : 0;
    return c;
}

template <int NS>
static int launch_calc_sets(const DaspGeom& g, int n, const uint8_t* bgr, const kde_float3* pts, const CalcSets<NS>& sets,
                            bool first, hipStream_t s)
{
    const float half = (float)(g.wx + g.wy) / 2.0f;
    const float win2 = half * half;
    const int nclusters = g.rows * g.cols;
    const int tables = first ? 1 : NS;
    const bool lds = tables * nclusters <= kMaxLdsClusters;
    const size_t bytes = lds ? (size_t)tables * nclusters * sizeof(ClusterRec) : 0;
    dim3 grid(ceil_div(g.width, 64), ceil_div(g.height, 4), n);
    CalcDivs dv;
    dv.wx = make_fastdiv24((uint32_t)g.wx, (uint64_t)g.width);
    dv.wy = make_fastdiv24((uint32_t)g.wy, (uint64_t)g.height);
    dv.cols = make_fastdiv24((uint32_t)g.cols, (uint64_t)g.width * g.height);      // labels are < rows * cols <= pixels
    if (first) {
        if (lds) hipLaunchKernelGGL((calc_ld_kernel<NS, true, true>), grid, dim3(256), bytes, s, g, bgr, pts, sets, win2, dv);
        else hipLaunchKernelGGL((calc_ld_kernel<NS, false, true>), grid, dim3(256), 0, s, g, bgr, pts, sets, win2, dv);
    } else {
        if (lds) hipLaunchKernelGGL((calc_ld_kernel<NS, true, false>), grid, dim3(256), bytes, s, g, bgr, pts, sets, win2, dv);
        else hipLaunchKernelGGL((calc_ld_kernel<NS, false, false>), grid, dim3(256), 0, s, g, bgr, pts, sets, win2, dv);
    }
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

// first = the step right after sampleInitialClusters (init_LD is folded in, see calc_ld_kernel)
int launch_dasp_calc_ld(const DaspGeom& g, const uint8_t* bgr, const kde_float3* pts, kde_label_distance* ld,
                        const kde_superpixel* mean, const kde_float3* centers, int32_t* labels, float color_sigma,
                        float spatial_sigma, float depth_sigma, bool first, hipStream_t s)
{
    CalcSets<1> sets;
    sets.s[0] = make_set(g, ld, mean, centers, labels, color_sigma, spatial_sigma, depth_sigma);
    sets.write_ld = 1;
    return launch_calc_sets<1>(g, 1, bgr, pts, sets, first, s);
}

// both segmenters of a pipeline in one pass (see CalcSet); with first = true both read set A's sampled clusters
int launch_dasp_calc_ld_dual(const DaspGeom& g, int n, const uint8_t* bgr, const kde_float3* pts, kde_label_distance* ld_a,
                             const kde_superpixel* mean_a, const kde_float3* centers_a, int32_t* labels_a,
                             const float sig_a[3], kde_label_distance* ld_b, const kde_superpixel* mean_b,
                             const kde_float3* centers_b, int32_t* labels_b, const float sig_b[3], bool first, bool write_ld,
                             hipStream_t s)
{
    CalcSets<2> sets;
    sets.s[0] = make_set(g, ld_a, mean_a, centers_a, labels_a, sig_a[0], sig_a[1], sig_a[2]);
    sets.s[1] = make_set(g, ld_b, mean_b, centers_b, labels_b, sig_b[0], sig_b[1], sig_b[2]);
    sets.write_ld = write_ld ? 1 : 0;
    return launch_calc_sets<2>(g, n, bgr, pts, sets, first, s);
}

static int analyze_band_walk()
{
    static const int v = KDE_AB_ENV("KDE_K8_NO_BAND_WALK") == nullptr ? 1 : 0;      // (measurement build only)
    return v;
}

#ifdef KDE_AB_SWITCHES
// measurement build: KDE_K8_ROWS=1 selects the row-coalesced form of K8 where its stage fits (measured slower, see above)
static bool analyze_rows_form()
{
    static const bool v = [] { const char* e = KDE_AB_ENV("KDE_K8_ROWS"); return e && e[0] != '0'; }();
    return v;
}
#endif

int launch_dasp_analyze(const DaspGeom& g, const uint8_t* bgr, const kde_float3* pts, const int32_t* labels,
                        kde_superpixel* mean, kde_float3* centers, const float* intr_dev, hipStream_t s)
{
    AnalyzeSets sets;
    sets.s[0] = sets.s[1] = AnalyzeSet{labels, mean, centers};
    sets.nsets = 1;
    sets.band = analyze_band_walk();
#ifdef KDE_AB_SWITCHES
    const size_t rows_lds = analyze_rows_form() ? analyze_rows_lds(g) : 0;
    if (rows_lds) hipLaunchKernelGGL(analyze_clusters_rows_kernel, dim3(g.cols * g.rows), dim3(256), rows_lds, s, g, bgr, pts, sets, intr_dev);
    else
#endif
        hipLaunchKernelGGL(analyze_clusters_kernel, dim3(g.cols * g.rows), dim3(256), 0, s, g, bgr, pts, sets, intr_dev);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

// both segmenters of a pipeline in one launch (same geometry, colour, cloud and intrinsics)
int launch_dasp_analyze_dual(const DaspGeom& g, int n, const uint8_t* bgr, const kde_float3* pts, const int32_t* labels_a,
                             kde_superpixel* mean_a, kde_float3* centers_a, const int32_t* labels_b, kde_superpixel* mean_b,
                             kde_float3* centers_b, const float* intr_dev, hipStream_t s)
{
    AnalyzeSets sets;
    sets.s[0] = AnalyzeSet{labels_a, mean_a, centers_a};
    sets.s[1] = AnalyzeSet{labels_b, mean_b, centers_b};
    sets.nsets = 2;
    sets.band = analyze_band_walk();
    const dim3 grid((unsigned)(g.cols * g.rows * 2 * n));
#ifdef KDE_AB_SWITCHES
    const size_t rows_lds = analyze_rows_form() ? analyze_rows_lds(g) : 0;
    if (rows_lds) hipLaunchKernelGGL(analyze_clusters_rows_kernel, grid, dim3(256), rows_lds, s, g, bgr, pts, sets, intr_dev);
    else
#endif
        hipLaunchKernelGGL(analyze_clusters_kernel, grid, dim3(256), 0, s, g, bgr, pts, sets, intr_dev);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

}  // namespace kde
