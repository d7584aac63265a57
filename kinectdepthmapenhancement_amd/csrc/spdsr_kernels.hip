// spdsr_kernels.hip — the tail of SPDepthSuperResolution::Process on the device:
//   * per-superpixel plane fit: the reference copies the cloud to the host, pushes every labelled point into
//     a cv::Mat per cluster (W*H heap allocations per frame) and runs cv::PCA per cluster
//     (SPDepthSuperResolution.cpp:65-142).  Here: two passes of double-precision moment accumulation
//     (LDS atomics per workgroup, then one global atomic per touched cluster), and a 3x3 symmetric Jacobi
//     eigen-solve per cluster — no host round trip;
//   * Projection_GPU::PlaneProjection(nd, labels, points): setPsuedoDepth [sic] + copy
//     (Projection_GPU.cu:55-81, 277-281) and 20 mrf_optimization sweeps (.cu:148-187, 282-285).  The sweeps
//     are in place and racy in the reference; deviation D5: every sweep reads the previous sweep's result.
// The accumulation order of the double sums is not fixed (atomics); results agree with the serial restatement
// to ~1e-15 relative before the cast to float.
#include "kde_internal.h"
#include "kde_device_math.h"

#include <type_traits>

namespace kde {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxLdsClusters = 1024;   // 4 or 6 doubles per cluster in LDS

// ---- moments ---------------------------------------------------------------------------------------------
template <bool COV>
__global__ __launch_bounds__(kThreads) void cluster_moments_kernel(int npix, int nclusters, int use_lds,
                                                                   const int32_t* __restrict__ labels,
                                                                   const kde_float3* __restrict__ pts,
                                                                   double* __restrict__ sums,   // [k][4]: n, sx, sy, sz
                                                                   double* __restrict__ cov)    // [k][6]
{
    constexpr int NV = COV ? 6 : 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* acc = reinterpret_cast<double*>(smem);
    {   // blockIdx.y = frame of a batch: own labels, cloud and moment tables
        const size_t f = blockIdx.y;
        labels += f * npix;
        pts += f * npix;
        sums += f * nclusters * 4;
        cov += f * nclusters * 6;
    }
    if (use_lds) {
        for (int i = threadIdx.x; i < nclusters * NV; i += kThreads) acc[i] = 0.0;
        __syncthreads();
    }
    double* out = COV ? cov : sums;
    // a thread walks RUN consecutive pixels: superpixel labels change rarely along a row, so the moments of a run
    // with one label are summed in registers and cost one set of atomics instead of one per pixel
    constexpr int RUN = 8;
    const int per_block = kThreads * RUN;
    const int first = blockIdx.x * per_block + threadIdx.x * RUN;
    double v[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = 0.0;
    int cur = -1;                                      // label of the open run (-1: none)
    auto flush = [&]() {
        if (cur >= 0) {
            double* dst = use_lds ? acc + cur * NV : out + cur * NV;
#pragma unroll
            for (int k = 0; k < NV; k++) atomicAdd(dst + k, v[k]);
#pragma unroll
            for (int k = 0; k < NV; k++) v[k] = 0.0;
        }
    };
    double mx = 0.0, my = 0.0, mz = 0.0;               // COV: mean of the open run's cluster
#pragma unroll
    for (int r = 0; r < RUN; r++) {
        const int i = first + r;
        if (i >= npix) break;
        const int l = labels[i];
        if (l < 0 || l >= nclusters) continue;         // label != -1 (SPDepthSuperResolution.cpp:70)
        if (l != cur) {
            flush();
            cur = l;
            if (COV) {
                const double n = sums[l * 4];
                mx = sums[l * 4 + 1] / n;
                my = sums[l * 4 + 2] / n;
                mz = sums[l * 4 + 3] / n;
            }
        }
        const kde_float3 p = pts[i];
        if (!COV) {
            v[0] += 1.0; v[1] += (double)p.x; v[2] += (double)p.y; v[3] += (double)p.z;
        } else {
            const double dx = (double)p.x - mx, dy = (double)p.y - my, dz = (double)p.z - mz;
            v[0] += dx * dx; v[1] += dx * dy; v[2] += dx * dz; v[3] += dy * dy; v[4] += dy * dz; v[5] += dz * dz;
        }
    }
    flush();
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nclusters * NV; i += kThreads)
            if (acc[i] != 0.0) atomicAdd(out + i, acc[i]);
    }
}

// ---- plane per cluster (cv::PCA + sign convention, SPDepthSuperResolution.cpp:84-138) -----------------------
__device__ void jacobi_eigen3(double A[3][3], double evals[3], double evecs[3][3])
{
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        // converged to double precision (cyclic Jacobi converges quadratically: ~6 sweeps); waiting for an exact 0
        // often runs all 60 sweeps on rounding noise and changes nothing a float normal can show
        const double diag = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1.0e-34 * diag) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int order[3] = {0, 1, 2};
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (A[order[j]][order[j]] > A[order[i]][order[i]]) {
                const int t = order[i];
                order[i] = order[j];
                order[j] = t;
            }
    for (int i = 0; i < 3; i++) {
        evals[i] = A[order[i]][order[i]];
        for (int k = 0; k < 3; k++) evecs[i][k] = V[k][order[i]];
    }
}

// The moment tables are accumulated with atomics into zeroed memory; this kernel is their only reader and clears the
// entries it has consumed, so that the next call finds them zero again (instead of two memset launches per call).
__global__ __launch_bounds__(64) void cluster_planes_kernel(int nclusters, double* __restrict__ sums,
                                                           double* __restrict__ cov, float4* __restrict__ nd)
{
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= nclusters) return;
    sums += (size_t)blockIdx.y * nclusters * 4;        // blockIdx.y = frame of a batch
    cov += (size_t)blockIdx.y * nclusters * 6;
    nd += (size_t)blockIdx.y * nclusters;
    const double cnt = sums[l * 4];
    if (cnt >= 3.0) {
        double A[3][3] = {{cov[l * 6] / cnt, cov[l * 6 + 1] / cnt, cov[l * 6 + 2] / cnt},
                          {cov[l * 6 + 1] / cnt, cov[l * 6 + 3] / cnt, cov[l * 6 + 4] / cnt},
                          {cov[l * 6 + 2] / cnt, cov[l * 6 + 4] / cnt, cov[l * 6 + 5] / cnt}};
        double ev[3], evec[3][3];
        jacobi_eigen3(A, ev, evec);
        float nx = (float)evec[2][0], ny = (float)evec[2][1], nz = (float)evec[2][2];
        const double gx = sums[l * 4 + 1] / cnt, gy = sums[l * 4 + 2] / cnt, gz = sums[l * 4 + 3] / cnt;
        const double plane_d_tmp = nx * gx + ny * gy + nz * gz;
        if (plane_d_tmp < 0) {
            nx = (float)(nx * -1.0);
            ny = (float)(ny * -1.0);
            nz = (float)(nz * -1.0);
        }
        nd[l] = make_float4(nx, ny, nz, (float)fabs(plane_d_tmp));
    } else {
        float4 v = nd[l];                 // w keeps its previous value, like the reference's pinned ClusterND_Host
        v.x = v.y = v.z = 5.0f;
        nd[l] = v;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) sums[l * 4 + k] = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) cov[l * 6 + k] = 0.0;
}

// ---- Projection_GPU ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void init_normalized_kernel(Camera c, float2* __restrict__ nxy)
{
    // initTemp, Projection_GPU.cu:3-19 (D4: every pixel is covered)
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= c.width || y >= c.height) return;
    float tx = (float)x, ty = (float)y;
    ty = (float)c.cy - ty;
    tx = tx - (float)c.cx;
    tx /= c.fx;
    ty /= c.fy;
    nxy[(size_t)y * c.width + x] = make_float2(tx * 1.0f, ty * 1.0f);
}

__global__ __launch_bounds__(kThreads) void set_pseudo_depth_kernel(int npix, int nclusters, const float4* __restrict__ nd,
                                                                   const int32_t* __restrict__ labels,
                                                                   const kde_float3* __restrict__ pts,
                                                                   const float2* __restrict__ nxy,
                                                                   kde_float3* __restrict__ plane_fitted,
                                                                   float* __restrict__ z0, float* __restrict__ pfz)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= npix) return;
    {   // blockIdx.y = frame of a batch (the normalized rays nxy belong to the camera and are shared)
        const size_t f = blockIdx.y;
        nd += f * nclusters; labels += f * npix; pts += f * npix; plane_fitted += f * npix; z0 += f * npix; pfz += f * npix;
    }
    const int l = labels[i];
    const kde_float3 p = pts[i];
    kde_float3 pf = p;
    if (l > -1 && l < nclusters) {
        const float4 n = nd[l];
        if (fabsf(n.x) < 1.0f) {                     // Projection_GPU.cu:70 (normal (5,5,5) marks "no plane")
            const float2 r = nxy[i];
            const float z = fabsf(n.w / (n.x * r.x + n.y * r.y + n.z));
            pf.z = z;
            pf.x = z * r.x;
            pf.y = z * r.y;
        }
    }
    plane_fitted[i] = pf;
    // state of the mrf sweeps (see mrf_sweep_kernel): only z evolves, so the sweeps run on compact planes
    z0[i] = p.z > 0.0f ? p.z : 0.0f;                  // optimized = input cloud (the cudaMemcpy of .cu:281); a z <= 0
                                                      // or NaN can neither be a tap nor pass the centre test, like 0
    pfz[i] = pf.z;
}

// mrf_optimization, Projection_GPU.cu:148-187 with (window 5, K 0.5, smooth_sigma 1.0), D5 (each sweep reads the
// previous sweep's buffer).  Only the z of a point evolves (x, y = normalized ray * z whenever z is rewritten), so a
// sweep reads and writes ONE float per pixel instead of two float3 (44 -> 12 B/px): the planes hold z, negated once
// the pixel has been rewritten at least once (a rewritten z is a weighted mean of values > 50, never 0), and
// mrf_expand_kernel turns the final plane back into points -- rewritten pixels as ray * z, the others untouched
// from the input cloud, exactly what 20 in-place float3 sweeps leave behind.
// (Fusing several sweeps per launch through LDS with a halo was measured slower: the sweep is VALU-bound, and the
// halo recomputation costs more than the saved traffic.  What pays is packed math: a thread owns two horizontally
// adjacent pixels and every float op of the 25 taps is a v_pk_*_f32 on the pair -- see jbf_fast.hip for the unit
// geometry -- with the "tap is valid" select folded into a {0, 0.5} factor prepared once per loaded LDS pair.)
typedef float s_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s_f2 s_add_clamp(s_f2 a, s_f2 b)
{
    s_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

constexpr int kSwBX = 32, kSwBY = 8;                  // threads; tile = 64 x 8 pixels
__global__ __launch_bounds__(kSwBX* kSwBY) void mrf_sweep_kernel(int width, int height, const float* __restrict__ zin,
                                                                const float* __restrict__ pfz, float* __restrict__ zout, int band)
{
    constexpr int R = 2, WIN = 5, HALF = 2, TW = kSwBX * 2, TH = kSwBY, LW = TW + 2 * R, LH = TH + 2 * R;
    __shared__ __attribute__((aligned(8))) float sz[LH * LW];
    // 1-D grid of frames x tiles walked in XCD bands (kde_device_math.h): vertically adjacent tiles share 4 of their 12
    // staged rows, and a sweep reads what the previous sweep's neighbours wrote -- within one XCD's L2 instead of across the fabric
    const unsigned tiles_x = (unsigned)((width + TW - 1) / TW), tiles = tiles_x * (unsigned)((height + TH - 1) / TH);
    const unsigned gid = band ? xcd_band_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const unsigned frame = gid / tiles, tile = gid - frame * tiles;
    {
        const size_t fpx = (size_t)frame * width * height;        // frame of a batch
        zin += fpx; pfz += fpx; zout += fpx;
    }
    const int x0 = (int)(tile % tiles_x) * TW, y0 = (int)(tile / tiles_x) * TH;
    const int tx = threadIdx.x % kSwBX, ty = threadIdx.x / kSwBX;
    const int x = x0 + 2 * tx, y = y0 + ty;
    const bool own = x < width && y < height, has1 = own && x + 1 < width;
    const size_t p = (size_t)y * width + x;
    // this thread's own pixels first: their loads are in flight while the tile is staged (one round trip, not two)
    const float zs0 = own ? zin[p] : 0.0f, zs1 = has1 ? zin[p + 1] : 0.0f;
    const s_f2 pf = {own ? pfz[p] : 0.0f, has1 ? pfz[p + 1] : 0.0f};
    for (int i = threadIdx.x; i < LW * LH; i += kSwBX * kSwBY) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = x0 + lx - R, gy = y0 + ly - R;
        float z = 0.0f;
        if (gx >= 0 && gx < width && gy >= 0 && gy < height) z = fabsf(zin[(size_t)gy * width + gx]);
        sz[i] = z > 50.0f ? z : 0.0f;                 // taps need optimized.z > 50 (.cu:167)
    }
    __syncthreads();
    if (!own) return;
    s_f2 oz = {fabsf(zs0), fabsf(zs1)};
    bool rw0 = zs0 < 0.0f, rw1 = zs1 < 0.0f;
    const bool c0 = pf.x > 50.0f && fabsf(oz.x - pf.x) < oz.x * 0.01f;
    const bool c1 = pf.y > 50.0f && fabsf(oz.y - pf.y) < oz.y * 0.01f;
    if (c0 || c1) {
        s_f2 num = pf, den = {1.0f, 1.0f};
        const s_f2 one = {1.0f, 1.0f}, half = {0.5f, 0.5f};
#pragma unroll
        for (int i = 0; i < WIN; i++) {
            s_f2 q[3], hv[3];
#pragma unroll
            for (int m = 0; m < 3; m++) {
                q[m] = *reinterpret_cast<const s_f2*>(&sz[(ty + i) * LW + 2 * tx + 2 * m]);
                hv[m] = s_add_clamp(q[m], q[m]) * half;             // 0.5 for a valid tap (z > 50), 0 otherwise
            }
#pragma unroll
            for (int u = 0; u < WIN; u++) {
                // unit u = (tap of p0, tap of p1) out of ONE aligned pair: straight / swapped / leftover
                s_f2 oq, h;
                if (u <= HALF) {
                    oq = q[u];
                    h = hv[u];
                } else if (u < WIN - 1) {
                    oq = __builtin_shufflevector(q[u - HALF], q[u - HALF], 1, 0);
                    h = __builtin_shufflevector(hv[u - HALF], hv[u - HALF], 1, 0);
                } else {
                    oq = s_f2{q[0].y, q[HALF].x};
                    h = s_f2{hv[0].y, hv[HALF].x};
                }
                const s_f2 diff = oz - oq;
                const s_f2 t1 = one + diff * diff;
                // K / (1 + diff^2) * smooth_sigma with K = 0.5, smooth_sigma = 1; v_rcp_f32 (1 ulp) for the division
                const s_f2 filter = s_f2{__builtin_amdgcn_rcpf(t1.x), __builtin_amdgcn_rcpf(t1.y)} * h;
                num = __builtin_elementwise_fma(oq, filter, num);
                den = den + filter;
            }
        }
        if (c0 && den.x != 0.0f) {
            oz.x = num.x / den.x;
            rw0 = true;
        }
        if (c1 && den.y != 0.0f) {
            oz.y = num.y / den.y;
            rw1 = true;
        }
    }
    zout[p] = rw0 ? -oz.x : oz.x;
    if (has1) zout[p + 1] = rw1 ? -oz.y : oz.y;
}

#ifdef KDE_AB_SWITCHES      // measured slower: compiled for tools/bench_spdsr.py and the bit-identity test only
// Two sweeps per launch (temporal blocking) -- MEASURED AND NOT USED BY DEFAULT (KDE_SPDSR_TWO_SWEEPS=1 selects it).
// A single sweep moves 12 B per pixel and is half latency (VALU busy 0.43 at 1080p, 20 launches of 16 us each); here a
// workgroup that owns a 64 x 32 tile stages the z plane with a halo of 4, evaluates sweep k on the tile + a halo of 2
// (68 x 36 = 1.2 x the tile) into LDS, and sweep k + 1 on the tile from there.  Per pixel the arithmetic is exactly
// mrf_sweep_kernel's (tests/test_gpu_dasp_ers.py: same bits), but it is 3 % slower at 1080p and 15 % at 640x480: fewer,
// longer workgroups expose the same latency chain with less of the GPU occupied.  (r02 had measured the same loss on a
// 64 x 8 tile, where the recomputed halo is 1.6 x the tile.)
constexpr int kSw2TW = 64, kSw2TH = 32;
__global__ __launch_bounds__(256) void mrf_sweep2_kernel(int width, int height, const float* __restrict__ zin,
                                                         const float* __restrict__ pfz, float* __restrict__ zout)
{
    constexpr int WIN = 5, HALF = 2;
    constexpr int AW = kSw2TW + 8, AH = kSw2TH + 8;        // staged input: halo 4
    constexpr int BW = kSw2TW + 4, BH = kSw2TH + 4;        // first sweep's result: halo 2
    __shared__ __attribute__((aligned(8))) float sa[AH * AW];       // |z| if > 50 else 0 (what a tap sees)
    __shared__ __attribute__((aligned(8))) float sb[BH * BW];       // signed state after the first sweep
    __shared__ __attribute__((aligned(8))) float sp[BH * BW];       // plane-fitted z of the same region
    {
        const size_t fpx = (size_t)blockIdx.z * width * height;    // blockIdx.z = frame of a batch
        zin += fpx; pfz += fpx; zout += fpx;
    }
    const int x0 = blockIdx.x * kSw2TW, y0 = blockIdx.y * kSw2TH;
    const int tid = threadIdx.x;
    for (int i = tid; i < AW * AH; i += 256) {
        const int ly = i / AW, lx = i - ly * AW;
        const int gx = x0 + lx - 4, gy = y0 + ly - 4;
        float z = 0.0f;
        if (gx >= 0 && gx < width && gy >= 0 && gy < height) z = fabsf(zin[(size_t)gy * width + gx]);
        sa[i] = z > 50.0f ? z : 0.0f;
    }
    __syncthreads();
    const s_f2 one = {1.0f, 1.0f}, half = {0.5f, 0.5f};
    // one pixel pair of one sweep: taps from `plane` (pitch P, the pair's window starts at plane[row0 * P + col0], col0
    // even), own signed state zs, plane-fitted z pf -> new signed state.  Identical to the body of mrf_sweep_kernel.
    auto sweep_pair = [&](const float* plane, int P, int row0, int col0, s_f2 zs, s_f2 pf, auto absval) -> s_f2 {
        s_f2 oz = {fabsf(zs.x), fabsf(zs.y)};
        bool rw0 = zs.x < 0.0f, rw1 = zs.y < 0.0f;
        const bool c0 = pf.x > 50.0f && fabsf(oz.x - pf.x) < oz.x * 0.01f;
        const bool c1 = pf.y > 50.0f && fabsf(oz.y - pf.y) < oz.y * 0.01f;
        if (c0 || c1) {
            s_f2 num = pf, den = {1.0f, 1.0f};
#pragma unroll
            for (int i = 0; i < WIN; i++) {
                s_f2 q[3], hv[3];
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    q[m] = *reinterpret_cast<const s_f2*>(&plane[(row0 + i) * P + col0 + 2 * m]);
                    if constexpr (decltype(absval)::value) {         // the intermediate plane holds signed state: taps see |z| > 50
                        q[m] = s_f2{fabsf(q[m].x), fabsf(q[m].y)};
                        q[m] = s_f2{q[m].x > 50.0f ? q[m].x : 0.0f, q[m].y > 50.0f ? q[m].y : 0.0f};
                    }
                    hv[m] = s_add_clamp(q[m], q[m]) * half;
                }
#pragma unroll
                for (int u = 0; u < WIN; u++) {
                    s_f2 oq, h;
                    if (u <= HALF) {
                        oq = q[u];
                        h = hv[u];
                    } else if (u < WIN - 1) {
                        oq = __builtin_shufflevector(q[u - HALF], q[u - HALF], 1, 0);
                        h = __builtin_shufflevector(hv[u - HALF], hv[u - HALF], 1, 0);
                    } else {
                        oq = s_f2{q[0].y, q[HALF].x};
                        h = s_f2{hv[0].y, hv[HALF].x};
                    }
                    const s_f2 diff = oz - oq;
                    const s_f2 t1 = one + diff * diff;
                    const s_f2 filter = s_f2{__builtin_amdgcn_rcpf(t1.x), __builtin_amdgcn_rcpf(t1.y)} * h;
                    num = __builtin_elementwise_fma(oq, filter, num);
                    den = den + filter;
                }
            }
            if (c0 && den.x != 0.0f) {
                oz.x = num.x / den.x;
                rw0 = true;
            }
            if (c1 && den.y != 0.0f) {
                oz.y = num.y / den.y;
                rw1 = true;
            }
        }
        return s_f2{rw0 ? -oz.x : oz.x, rw1 ? -oz.y : oz.y};
    };
    // ---- first sweep on the tile + halo 2 (BW x BH, pairs) ----
    for (int i = tid; i < (BW / 2) * BH; i += 256) {
        const int by = i / (BW / 2), bx = 2 * (i - by * (BW / 2));
        const int gx = x0 + bx - 2, gy = y0 + by - 2;
        s_f2 zs = {0.0f, 0.0f}, pf = {0.0f, 0.0f};
        const bool in_y = gy >= 0 && gy < height;
        const bool in0 = in_y && gx >= 0 && gx < width, in1 = in_y && gx + 1 >= 0 && gx + 1 < width;
        if (in0) {
            zs.x = zin[(size_t)gy * width + gx];
            pf.x = pfz[(size_t)gy * width + gx];
        }
        if (in1) {
            zs.y = zin[(size_t)gy * width + gx + 1];
            pf.y = pfz[(size_t)gy * width + gx + 1];
        }
        s_f2 r = sweep_pair(sa, AW, by, bx, zs, pf, std::false_type{});
        // a pixel outside the image never is a tap (it stages as 0) and never is written
        if (!in0) r.x = 0.0f;
        if (!in1) r.y = 0.0f;
        *reinterpret_cast<s_f2*>(&sb[by * BW + bx]) = r;
        *reinterpret_cast<s_f2*>(&sp[by * BW + bx]) = pf;
    }
    __syncthreads();
    // ---- second sweep on the tile, taps and own state from the first sweep's result ----
    for (int i = tid; i < (kSw2TW / 2) * kSw2TH; i += 256) {
        const int ty = i / (kSw2TW / 2), tx = 2 * (i - ty * (kSw2TW / 2));
        const int x = x0 + tx, y = y0 + ty;
        if (x >= width || y >= height) continue;
        const s_f2 zs = *reinterpret_cast<const s_f2*>(&sb[(ty + 2) * BW + tx + 2]);
        const s_f2 pf = *reinterpret_cast<const s_f2*>(&sp[(ty + 2) * BW + tx + 2]);
        const s_f2 r = sweep_pair(sb, BW, ty, tx, zs, pf, std::true_type{});
        const size_t p = (size_t)y * width + x;
        zout[p] = r.x;
        if (x + 1 < width) zout[p + 1] = r.y;
    }
}
#endif  // KDE_AB_SWITCHES

__global__ __launch_bounds__(kThreads) void mrf_expand_kernel(int npix, const float* __restrict__ zfinal,
                                                             const kde_float3* __restrict__ pts,
                                                             const float2* __restrict__ nxy, kde_float3* __restrict__ out)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= npix) return;
    {
        const size_t f = blockIdx.y;                    // frame of a batch
        zfinal += f * npix; pts += f * npix; out += f * npix;
    }
    const float zs = zfinal[i];
    kde_float3 o = pts[i];
    if (zs < 0.0f) {
        const float2 r = nxy[i];
        const float depth = -zs;
        o.z = depth;
        o.x = r.x * depth;
        o.y = r.y * depth;
    }
    out[i] = o;
}

}  // namespace

int launch_spdsr_init_normalized(const Camera& c, float* nxy, hipStream_t s)
{
    hipLaunchKernelGGL(init_normalized_kernel, dim3(ceil_div(c.width, 64), ceil_div(c.height, 4)), dim3(kThreads), 0, s, c,
                       reinterpret_cast<float2*>(nxy));
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

// n frames back to back (labels, cloud; sums / cov / nd hold one table per frame)
int launch_spdsr_cluster_planes(int width, int height, int n, int nclusters, int table_frames, const int32_t* labels,
                                const kde_float3* pts, double* sums, double* cov, float* nd, int* moments_dirty, hipStream_t s)
{
    const int npix = width * height;
    // sums / cov are zero on entry: cleared at creation and by cluster_planes_kernel after every use.  *moments_dirty (a host
    // flag of the handle) is raised before the moments are accumulated and lowered once the kernel that consumes and clears
    // them has been enqueued: a call that finds it raised -- an earlier call failed between the two -- clears the tables
    // itself, so a failed launch cannot make every later Process on the handle return wrong planes.
    if (*moments_dirty) {
        KDE_HIP_TRY(hipMemsetAsync(sums, 0, (size_t)table_frames * nclusters * 4 * sizeof(double), s));
        KDE_HIP_TRY(hipMemsetAsync(cov, 0, (size_t)table_frames * nclusters * 6 * sizeof(double), s));
    }
    *moments_dirty = 1;
    const int use_lds = nclusters <= kMaxLdsClusters;
    const int blocks = ceil_div(npix, kThreads * 8);
    hipLaunchKernelGGL(cluster_moments_kernel<false>, dim3(blocks, n), dim3(kThreads), use_lds ? (size_t)nclusters * 4 * 8 : 0, s,
                       npix, nclusters, use_lds, labels, pts, sums, cov);
    KDE_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(cluster_moments_kernel<true>, dim3(blocks, n), dim3(kThreads), use_lds ? (size_t)nclusters * 6 * 8 : 0, s,
                       npix, nclusters, use_lds, labels, pts, sums, cov);
    KDE_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(cluster_planes_kernel, dim3(ceil_div(nclusters, 64), n), dim3(64), 0, s, nclusters, sums, cov,
                       reinterpret_cast<float4*>(nd));
    KDE_HIP_TRY(hipGetLastError());
    *moments_dirty = 0;
    return KDE_OK;
}

int launch_spdsr_plane_projection(int width, int height, int n, int nclusters, const float* nd, const int32_t* labels,
                                  const kde_float3* pts, const float* nxy, kde_float3* plane_fitted, kde_float3* opt_a,
                                  kde_float3* opt_b, int sweeps, kde_float3** result, hipStream_t s)
{
    const int npix = width * height;
    // opt_b (12 B/px per frame) is carved into the three compact planes of the sweeps: z ping, z pong, plane-fitted z
    // (each n frames back to back)
    float* zping = reinterpret_cast<float*>(opt_b);
    float* zpong = zping + (size_t)n * npix;
    float* pfz = zpong + (size_t)n * npix;
    hipLaunchKernelGGL(set_pseudo_depth_kernel, dim3(ceil_div(npix, kThreads), n), dim3(kThreads), 0, s, npix, nclusters,
                       reinterpret_cast<const float4*>(nd), labels, pts, reinterpret_cast<const float2*>(nxy), plane_fitted,
                       zping, pfz);
    float *in = zping, *out = zpong;
    dim3 grid((unsigned)(ceil_div(width, kSwBX * 2) * ceil_div(height, kSwBY) * n));
    static const int band = KDE_AB_ENV("KDE_SWEEP_NO_BAND_WALK") == nullptr ? 1 : 0;      // (measurement build only)
    int i = 0;
#ifdef KDE_AB_SWITCHES
    // A/B switch for tools/bench_spdsr.py.  Measured on MI355X (r03): two sweeps per launch are bit-identical and SLOWER --
    // 0.978 vs 0.946 ms per 1080p frame, 0.383 vs 0.333 at 640x480 (150 workgroups there) -- so one sweep per launch stays.
    static const bool two_sweeps = KDE_AB_ENV("KDE_SPDSR_TWO_SWEEPS") != nullptr;
    dim3 grid2(ceil_div(width, kSw2TW), ceil_div(height, kSw2TH), n);
    for (; two_sweeps && i + 1 < sweeps; i += 2) {      // two sweeps per launch (mrf_sweep2_kernel)
        hipLaunchKernelGGL(mrf_sweep2_kernel, grid2, dim3(256), 0, s, width, height, in, pfz, out);
        float* t = in;
        in = out;
        out = t;
    }
#endif
    for (; i < sweeps; i++) {
        hipLaunchKernelGGL(mrf_sweep_kernel, grid, dim3(kSwBX * kSwBY), 0, s, width, height, in, pfz, out, band);
        float* t = in;
        in = out;
        out = t;
    }
    hipLaunchKernelGGL(mrf_expand_kernel, dim3(ceil_div(npix, kThreads), n), dim3(kThreads), 0, s, npix, in, pts,
                       reinterpret_cast<const float2*>(nxy), opt_a);
    KDE_HIP_TRY(hipGetLastError());
    *result = opt_a;
    return KDE_OK;
}

}  // namespace kde
