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

#ifdef KDE_AB_SWITCHES
// ---------------------------------------------------------------------------------------------------------------------
// All 20 sweeps of ONE frame in one launch, the state resident in LDS (r05).  The 20 launches above spend more than half of
// their 15 us each on fill, drain and the staging round trip (VALU busy 0.42), and a frame's z plane is small: 1080p is
// 8.3 MB against 40 MB of LDS on the chip.  So the frame is cut into one block per workgroup (a GX x GY grid of at most one
// workgroup per CU, 1024 threads each); a workgroup keeps the tap-visible z of its block + a halo of 2 in two LDS planes
// (ping-pong) and its own pixels' state (|z|, "rewritten" flag, plane-fitted z) in registers for all sweeps.  Per sweep
// only the 2-pixel rim of the block goes through memory: written to the global z plane of the sweep's parity with sc1
// (write-through) stores, announced by one agent-scope flag store per workgroup, and read by the (up to 8) neighbours
// with sc1 loads once their poll of that flag has matched (the hand-off form of MI355X_MICROARCH.md's table, first row:
// every storing wave drains its stores, workgroup barrier, ONE lane signals; ONE wave polls, workgroup barrier, sc1
// loads).  Two parities suffice: a workgroup can only write rim k + 2 after every neighbour has announced k + 1, i.e.
// has finished reading rim k.
// Per pixel the arithmetic is mrf_sweep_kernel's, instruction for instruction (same pairs: even global column first, same
// unit order), so the result is bit-identical to the 20 launches (tests/test_gpu_dasp_ers.py).
// Progress needs every workgroup resident: the launch is cooperative (the runtime checks the grid against the occupancy
// and runs cooperative launches of a device one after the other), and every spin is bounded -- a workgroup that waits longer
// than kResidentTimeoutTicks raises `status` (host-visible), which ends every other spin as well; the caller sees
// KDE_ERR_HIP at its next call instead of a hang.
//
// MEASURED AND NOT USED (measurement build only, KDE_SPDSR_RESIDENT=1 cooperative / 2 plain launch; tools/bench_spdsr.py,
// profiles/r05_ab_spdsr_resident.txt): bit-identical, the kernel takes 284 us per 1080p frame against 20 x 15.2 = 304 us
// plus 19 boundaries, but SPDepthSuperResolution::Process gains only 0.899 -> 0.873 ms (-2.9 %) with a PLAIN launch -- which
// is not safe when two handles run on one device (two partially resident grids wait for each other until the spin bound) --
// and LOSES with the cooperative launch that makes it safe: 0.927 ms (the runtime's cooperative queue costs ~50 us per
// launch here; 640x480: 0.309 / 0.350 / 0.305 ms).  The sweep is bound by its own arithmetic (25 units x (5 packed + 2
// v_rcp_f32) per pair: ~11 of the 14 us per sweep), not by the launch structure the resident form removes.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kResThreads = 1024;
constexpr int kResMaxGrid = 16;                       // blocks per frame side: 16 x 16 = one workgroup per CU
constexpr unsigned long long kResidentTimeoutTicks = 100000000ull;     // 1 s of the 100 MHz wall clock

struct ResidentArgs {
    int width, height, sweeps;
    int gx, gy, bw, bh;                               // grid of blocks, block size in pixels (bw even)
    float* zplane[2];                                 // global z planes: [0] holds the initial state and receives rims of even parity
    const float* pfz;
    int* flags;                                       // one int per workgroup, 32 ints apart (a 128-byte line each)
    int gen;                                          // flags count up across calls: sweep k of this call is announced as gen + k + 1
    int* status;                                      // host-visible: 0 = fine, else (workgroup + 1) that gave up
};

__device__ __forceinline__ void st_sc1(float* p, float v)
{
    __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p)
{
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

template <int MAXP>     // pairs of pixels a thread owns at most
__global__ __launch_bounds__(kResThreads) void mrf_sweeps_resident_kernel(const ResidentArgs a)
{
    constexpr int R = 2, WIN = 5, HALF = 2;
    extern __shared__ __attribute__((aligned(16))) float res_lds[];
    __shared__ int published;                         // waves that have drained their rim stores, counted across sweeps
    const int LP = a.bw + 2 * R, LH = a.bh + 2 * R;   // LDS plane: block + halo, pitch even
    float* plane[2] = {res_lds, res_lds + LP * LH};
    const int tid = threadIdx.x;
    const int bxi = (int)blockIdx.x % a.gx, byi = (int)blockIdx.x / a.gx;
    const int X0 = bxi * a.bw, Y0 = byi * a.bh;
    const int bw_real = min(a.bw, a.width - X0), bh_real = min(a.bh, a.height - Y0);     // >= 1 by construction of the grid
    const int pw = a.bw / 2;                          // pairs per block row

    // ---- initial state: tap-visible z of block + halo from the global plane (written by the previous kernel) ----
    if (tid == 0) published = 0;
    for (int i = tid; i < LP * LH; i += kResThreads) {
        const int ly = i / LP, lx = i - ly * LP;
        const int gx = X0 + lx - R, gy = Y0 + ly - R;
        float z = 0.0f;
        if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) z = fabsf(a.zplane[0][(size_t)gy * a.width + gx]);
        plane[0][i] = z > 50.0f ? z : 0.0f;
        plane[1][i] = 0.0f;                           // out-of-image halo cells stay 0 in both planes
    }
    // ---- this thread's pairs.  Pair list of the block: the RIM first -- the pairs of rows 0, 1, bh - 2, bh - 1 and the first
    // and last pair of the rows between: what the neighbours read, and the only pairs whose window reaches into the halo --
    // then the interior.  Slot 0 (pairs 0 .. 1023) holds the whole rim (resident_geometry): it is computed LAST in a sweep,
    // once the neighbours' rims have arrived, and the interior slots before it need nothing from outside the block.
    int off[MAXP];                                    // (row * LP + 2 * pair column) = top-left cell of the 5 x 6 window; -1 = no pair
    s_f2 oz[MAXP], pf[MAXP];
    unsigned rw = 0, has = 0;                         // bit 2k: pixel 0 of pair k (rewritten / exists), bit 2k + 1: pixel 1
    const int n_rim = 4 * pw + 2 * (a.bh - 4), n_pairs = pw * a.bh;
#pragma unroll
    for (int k = 0; k < MAXP; k++) {
        const int pi = tid + k * kResThreads;
        off[k] = -1;
        oz[k] = pf[k] = s_f2{0.0f, 0.0f};
        if (pi < n_pairs) {
            int row, pc;
            if (pi < 2 * pw) {
                row = pi / pw; pc = pi - row * pw;
            } else if (pi < 4 * pw) {
                const int j = pi - 2 * pw;
                row = j / pw; pc = j - row * pw; row += a.bh - 2;
            } else if (pi < n_rim) {
                const int j = pi - 4 * pw;
                row = 2 + (j >> 1); pc = (j & 1) ? pw - 1 : 0;
            } else {
                const int j = pi - n_rim;                 // interior: rows 2 .. bh - 3, pairs 1 .. pw - 2 (pw > 2 or none is left)
                row = j / (pw - 2); pc = 1 + j - row * (pw - 2); row += 2;
            }
            const int x = X0 + 2 * pc, y = Y0 + row;
            if (x < a.width && y < a.height) {
                off[k] = row * LP + 2 * pc;
                const size_t p = (size_t)y * a.width + x;
                const bool two = x + 1 < a.width;
                const float z0 = a.zplane[0][p], z1 = two ? a.zplane[0][p + 1] : 0.0f;
                oz[k] = s_f2{fabsf(z0), fabsf(z1)};
                pf[k] = s_f2{a.pfz[p], two ? a.pfz[p + 1] : 0.0f};
                has |= (1u << (2 * k)) | (two ? (2u << (2 * k)) : 0u);
                rw |= (z0 < 0.0f ? (1u << (2 * k)) : 0u) | (z1 < 0.0f ? (2u << (2 * k)) : 0u);
            }
        }
    }
    __syncthreads();

    const unsigned long long t_start = wall_clock64();
    const float* cur = plane[0];
    float* nxt = plane[1];
    // one pair: mrf_sweep_kernel's arithmetic on the planes in LDS
    auto sweep_pair = [&](int k) {
        if (off[k] < 0) return;
        s_f2 z = oz[k];
        const bool p1 = (has >> (2 * k + 1)) & 1u;
        const bool c0 = pf[k].x > 50.0f && fabsf(z.x - pf[k].x) < z.x * 0.01f;
        const bool c1 = p1 && pf[k].y > 50.0f && fabsf(z.y - pf[k].y) < z.y * 0.01f;
        if (c0 || c1) {
            s_f2 num = pf[k], den = {1.0f, 1.0f};
            const s_f2 one = {1.0f, 1.0f}, half = {0.5f, 0.5f};
#pragma unroll
            for (int i = 0; i < WIN; i++) {
                s_f2 q[3], hv[3];
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    q[m] = *reinterpret_cast<const s_f2*>(&cur[off[k] + i * LP + 2 * m]);
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
                    const s_f2 diff = z - oq;
                    const s_f2 t1 = one + diff * diff;
                    const s_f2 filter = s_f2{__builtin_amdgcn_rcpf(t1.x), __builtin_amdgcn_rcpf(t1.y)} * h;
                    num = __builtin_elementwise_fma(oq, filter, num);
                    den = den + filter;
                }
            }
            if (c0 && den.x != 0.0f) {
                z.x = num.x / den.x;
                rw |= 1u << (2 * k);
            }
            if (c1 && den.y != 0.0f) {
                z.y = num.y / den.y;
                rw |= 2u << (2 * k);
            }
            oz[k] = z;
        }
        // what the next sweep's taps see of this pair (a pixel beyond the image edge stays 0)
        const s_f2 vis = {z.x > 50.0f ? z.x : 0.0f, (p1 && z.y > 50.0f) ? z.y : 0.0f};
        *reinterpret_cast<s_f2*>(&nxt[off[k] + R * LP + R]) = vis;
    };

    for (int sw = 0; sw < a.sweeps; sw++) {
        cur = plane[sw & 1];
        nxt = plane[(sw + 1) & 1];
        // ---- A: the interior slots (their windows stay inside the block, complete since the last barrier) ----
#pragma unroll
        for (int k = 1; k < MAXP; k++) sweep_pair(k);
        // ---- H: the neighbours' rims of sweep sw - 1 -> the halo ring of `cur` (the initial load covered sweep 0) ----
        if (sw > 0) {
            bool gave_up = false;
            const int target = a.gen + sw;                        // announcement of rim sw - 1
            if (tid < 64) {                                       // one wave polls, lane d = direction d
                bool waiting = false;
                int nb = 0;
                if (tid < 8) {
                    const int d = tid < 4 ? tid : tid + 1;        // 3 x 3 neighbourhood without the centre
                    const int nx = bxi + d % 3 - 1, ny = byi + d / 3 - 1;
                    if (nx >= 0 && nx < a.gx && ny >= 0 && ny < a.gy) {
                        waiting = true;
                        nb = ny * a.gx + nx;
                    }
                }
                int spins = 0;
                while (true) {
                    if (waiting && __hip_atomic_load(&a.flags[nb * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target >= 0) waiting = false;
                    if (!__any(waiting)) break;
                    if ((++spins & 63) == 0) {                    // bounded: nobody waits for a workgroup that never became resident
                        if (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0 ||
                            wall_clock64() - t_start > kResidentTimeoutTicks) {
                            gave_up = true;
                            break;
                        }
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (gave_up && tid == 0) __hip_atomic_store(a.status, (int)blockIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (__syncthreads_or(gave_up)) return;                // every workgroup leaves: the others see `status` in their own spin
            const float* __restrict__ g = a.zplane[sw & 1];       // rim sw - 1 went to the plane of parity sw
            float* ring = plane[sw & 1];
            const int ring_rows = 2 * R * LP;                     // top R + bottom R rows of the plane, full pitch
            const int ring_cols = 2 * R * a.bh;                   // left / right R columns of the rows between
            for (int i = tid; i < ring_rows + ring_cols; i += kResThreads) {
                int lx, ly;
                if (i < ring_rows) {
                    const int r = i / LP;
                    lx = i - r * LP;
                    ly = r < R ? r : a.bh + r;                    // plane rows 0, 1, bh + 2, bh + 3
                } else {
                    const int j = i - ring_rows, r = j / (2 * R), c = j - r * (2 * R);
                    ly = R + r;
                    lx = c < R ? c : a.bw + c;                    // plane columns 0, 1, bw + 2, bw + 3
                }
                const int gx = X0 + lx - R, gy = Y0 + ly - R;
                if (gx >= 0 && gx < a.width && gy >= 0 && gy < a.height) ring[ly * LP + lx] = ld_sc1(&g[(size_t)gy * a.width + gx]);
            }
        }
        __syncthreads();                                          // ring in place (and, for sw == 0, nothing to wait for)
        // ---- B: slot 0 = the rim (+ the first interior pairs) ----
        sweep_pair(0);
        if (sw + 1 == a.sweeps) break;                            // the last sweep's result leaves through the registers below
        __syncthreads();                                          // `nxt` complete: rim cells for the stores below, all cells for the next A
        // ---- publish this block's rim of sweep sw to the global plane of parity sw + 1: sc1 stores.  Every wave drains its own
        // stores and counts itself in LDS; the wave that counts last announces (no workgroup barrier: the others go on) ----
        {
            float* __restrict__ g = a.zplane[(sw + 1) & 1];
            const int rim_rows = 2 * R * bw_real;                         // top R rows + bottom R rows, full width
            const int rim_cols = 2 * R * max(bh_real - 2 * R, 0);         // left / right R columns of the rows between
            for (int i = tid; i < rim_rows + rim_cols; i += kResThreads) {
                int lx, ly;
                if (i < rim_rows) {
                    const int r = i / bw_real;
                    lx = i - r * bw_real;
                    ly = r < R ? r : bh_real - 2 * R + r;                 // rows 0, 1, bh - 2, bh - 1 (they coincide for bh < 4: harmless)
                    if (ly < 0) ly = r;
                } else {
                    const int j = i - rim_rows, r = j / (2 * R), c = j - r * (2 * R);
                    ly = R + r;
                    lx = c < R ? c : bw_real - 2 * R + c;
                    if (lx < 0) lx = c;
                }
                if (lx < bw_real && ly < bh_real)
                    st_sc1(&g[(size_t)(Y0 + ly) * a.width + X0 + lx], nxt[(ly + R) * LP + lx + R]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((tid & 63) == 0) {
                const int waves = kResThreads / 64;
                if (atomicAdd(&published, 1) + 1 == waves * (sw + 1))
                    __hip_atomic_store(&a.flags[blockIdx.x * 32], a.gen + sw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- final state: signed z (negative = rewritten at least once) for mrf_expand_kernel ----
    float* __restrict__ zout = a.zplane[a.sweeps & 1];
#pragma unroll
    for (int k = 0; k < MAXP; k++) {
        if (off[k] < 0) continue;
        const int row = off[k] / LP, col = off[k] - row * LP;
        const size_t p = (size_t)(Y0 + row) * a.width + X0 + col;
        zout[p] = ((rw >> (2 * k)) & 1u) ? -oz[k].x : oz[k].x;
        if ((has >> (2 * k + 1)) & 1u) zout[p + 1] = ((rw >> (2 * k + 1)) & 1u) ? -oz[k].y : oz[k].y;
    }
}
#endif  // KDE_AB_SWITCHES (resident sweeps)

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

#ifdef KDE_AB_SWITCHES
// Geometry of the resident form for a frame, or false when it does not apply: one block per workgroup on a grid of at most
// kResMaxGrid x kResMaxGrid, block width even (pairs start on even global columns, as in mrf_sweep_kernel), blocks at least
// 4 x 4 (a halo of 2 must not reach past the direct neighbour), both LDS planes within the CU's 160 KB, <= 8 pairs per thread
static bool resident_geometry(int width, int height, int cus, ResidentArgs& a, size_t& lds, int& maxp)
{
    const int g = cus >= kResMaxGrid * kResMaxGrid ? kResMaxGrid : 0;
    if (!g) return false;
    a.bw = (ceil_div(width, g) + 1) / 2 * 2;
    a.bh = ceil_div(height, g);
    if (a.bw < 4 || a.bh < 4) return false;
    a.gx = ceil_div(width, a.bw);
    a.gy = ceil_div(height, a.bh);
    lds = (size_t)2 * (a.bw + 4) * (a.bh + 4) * sizeof(float);
    const int pairs = a.bw / 2 * a.bh;
    maxp = ceil_div(pairs, kResThreads) <= 4 ? 4 : 8;
    const int rim = 4 * (a.bw / 2) + 2 * (a.bh - 4);                // the rim pairs are one thread's slot 0 each
    return lds <= 150 * 1024 && ceil_div(pairs, kResThreads) <= 8 && rim <= kResThreads;
}
#endif

int spdsr_resident_init(SpdsrResident* r)
{
#ifdef KDE_AB_SWITCHES       // the resident sweep launch exists in the measurement build only (see mrf_sweeps_resident_kernel)
    int dev = 0;
    KDE_HIP_TRY(hipGetDevice(&dev));
    KDE_HIP_TRY(hipDeviceGetAttribute(&r->cus, hipDeviceAttributeMultiprocessorCount, dev));
    int coop = 0;
    KDE_HIP_TRY(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    r->cooperative = coop;
    KDE_HIP_TRY(hipMalloc(&r->flags, (size_t)kResMaxGrid * kResMaxGrid * 32 * sizeof(int)));
    KDE_HIP_TRY(hipMemset(r->flags, 0, (size_t)kResMaxGrid * kResMaxGrid * 32 * sizeof(int)));
    KDE_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&r->status), sizeof(int), hipHostMallocDefault));
    *r->status = 0;
    r->gen = 0;
    KDE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mrf_sweeps_resident_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    KDE_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&mrf_sweeps_resident_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#else
    (void)r;
#endif
    return KDE_OK;
}

void spdsr_resident_release(SpdsrResident* r)
{
    if (r->flags) (void)hipFree(r->flags);
    if (r->status) (void)hipHostFree(r->status);
    r->flags = nullptr;
    r->status = nullptr;
}

int launch_spdsr_plane_projection(int width, int height, int n, int nclusters, const float* nd, const int32_t* labels,
                                  const kde_float3* pts, const float* nxy, kde_float3* plane_fitted, kde_float3* opt_a,
                                  kde_float3* opt_b, int sweeps, kde_float3** result, SpdsrResident* res, hipStream_t s)
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
#ifdef KDE_AB_SWITCHES
    // ---- one frame: all sweeps in one launch, state resident in LDS (mrf_sweeps_resident_kernel; measured, not used) ----
    static const int resident_mode = [] { const char* e = KDE_AB_ENV("KDE_SPDSR_RESIDENT"); return e ? e[0] - '0' : 0; }();   // 1 cooperative, 2 plain
    ResidentArgs ra;
    size_t lds = 0;
    int maxp = 4;
    if (resident_mode && n == 1 && sweeps >= 2 && res && res->flags && (res->cooperative || resident_mode == 2) &&
        resident_geometry(width, height, res->cus, ra, lds, maxp)) {
        if (*res->status != 0)
            return fail(KDE_ERR_HIP, "SPDSR: a resident sweep launch gave up waiting for workgroup %d (device shared with another "
                                     "persistent kernel?)", *res->status - 1);
        ra.width = width; ra.height = height; ra.sweeps = sweeps;
        ra.zplane[0] = zping; ra.zplane[1] = zpong; ra.pfz = pfz;
        ra.flags = res->flags; ra.gen = res->gen; ra.status = res->status;
        res->gen += sweeps + 1;
        void* args[] = {&ra};
        const void* fn = maxp == 4 ? reinterpret_cast<const void*>(&mrf_sweeps_resident_kernel<4>)
                                   : reinterpret_cast<const void*>(&mrf_sweeps_resident_kernel<8>);
        if (resident_mode == 2) KDE_HIP_TRY(hipLaunchKernel(fn, dim3(ra.gx * ra.gy), dim3(kResThreads), args, lds, s));
        else KDE_HIP_TRY(hipLaunchCooperativeKernel(fn, dim3(ra.gx * ra.gy), dim3(kResThreads), args, (unsigned)lds, s));
        hipLaunchKernelGGL(mrf_expand_kernel, dim3(ceil_div(npix, kThreads), n), dim3(kThreads), 0, s, npix, ra.zplane[sweeps & 1], pts,
                           reinterpret_cast<const float2*>(nxy), opt_a);
        KDE_HIP_TRY(hipGetLastError());
        *result = opt_a;
        return KDE_OK;
    }
#else
    (void)res;
#endif
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
