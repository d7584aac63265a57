// stream_kernels.hip — the HBM-bound element-wise feeders of the hot path:
//   DimensionConvertor (K2/K3): DimensionConvertor/DimensionConvertor.h:19-148, .cu:3-77
//   Buffer2D (K4):              ArrayBuffer/ArrayBuffer.cu:9-22, ArrayBuffer/Buffer2D.cu:13-147
// One element group per thread and NO grid-stride loop: on this chip a streaming copy written that way reaches
// 6.2 TB/s where the same copy as a capped persistent grid reaches 4.6-5.0 (profiles/r02_hbm_microbench.txt), so
// every launcher sizes its grid to cover the data once (the loops below run one trip and only guard the tail).
// 16-byte accesses where the record layout and the caller's pointers allow it; any other pointer alignment or
// frame size (the reference takes any float*) goes through the scalar kernels at the end of each section.
// The reference's redundant thrust::fill before every projectiveToReal (DimensionConvertor.cu:5-13) is not
// reproduced (every output element is overwritten).
#include "kde_internal.h"

namespace kde {
namespace {

constexpr int kThreads = 256;
constexpr size_t kMaxBlocks = 0x7fffffffull;   // one trip per thread; beyond 2^31 workgroups the loops take over

inline unsigned grid_for(size_t items)
{
    size_t b = (items + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (unsigned)b;
}

constexpr size_t kCacheBytes = (size_t)256 << 20;   // Infinity Cache: calls that move more than this stream past it

inline bool aligned(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// ---- DimensionConvertor ---------------------------------------------------------------------
__device__ __forceinline__ void convert_ptr(float& x, float& y, float z, const Camera& c)
{
    // DimensionConvertor.h:34-48 — subtract, divide, multiply (in that order)
    y = (float)c.cy - y;
    x = x - (float)c.cx;
    x /= c.fx;
    y /= c.fy;
    x *= z;
    y *= z;
}

// Four pixels per thread: one float4 depth load, three float4 stores (48 B = 4 packed float3).
// blockIdx.y = frame; the in-frame index is 32-bit so one division serves four pixels.
__device__ __forceinline__ void p2r_one(const Camera& c, int interp, unsigned x, unsigned y, float z, float* r)
{
    float fx_, fy_;
    if (!interp) {
        // convert_ptr(tuple<float,int>), DimensionConvertor.h:51-62
        fy_ = (float)(int)y;
        fx_ = (float)(int)x;
        convert_ptr(fx_, fy_, z, c);
    } else {
        // convert_ptr_int, DimensionConvertor.h:80-103 (half-pixel grid, index decomposed over 2*width)
        fy_ = (float)(int)y;
        fx_ = (float)(int)x;
        fy_ = (float)c.cy - fy_ / 2.0f;
        fx_ = fx_ / 2.0f - (float)c.cx;
        fx_ /= c.fx;
        fy_ /= c.fy;
        fx_ *= z;
        fy_ *= z;
    }
    r[0] = fx_;
    r[1] = fy_;
    r[2] = z;
}

// A thread owns 4 pixels = 48 B of packed float3 output, which it could only store as three 16-byte
// pieces 48 B apart.  Each wave therefore passes its 3 KB through LDS once, so that every global store
// (and, for the float3 -> float3 maps, every load) instruction moves 64 consecutive float4 = 1 KB.
// LDS instructions of one wave execute in issue order, so a wave-level compiler barrier is all that is
// needed between the write and the transposed read.
__device__ __forceinline__ void wave_exchange_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// NT: streaming (non-temporal) loads and stores, used when the call moves more than the 256 MB Infinity Cache can
// hold anyway (a batch); single frames keep default caching so that the next kernel of the chain reads them on-chip.
typedef float s_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float4* p, bool nt)
{
    if (!nt) return *p;
    const s_v4 v = __builtin_nontemporal_load(reinterpret_cast<const s_v4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4(float4* p, float4 v, bool nt)
{
    if (nt) __builtin_nontemporal_store(s_v4{v.x, v.y, v.z, v.w}, reinterpret_cast<s_v4*>(p));
    else *p = v;
}

template <bool NT>
__global__ __launch_bounds__(kThreads) void p2r_depth_kernel(Camera c, const float* __restrict__ depth_all,
                                                            float* __restrict__ out_all, int interp)
{
    __shared__ float4 xch[kThreads / 64][3 * 64];
    const unsigned frame_px = (unsigned)c.width * (unsigned)c.height;
    const unsigned row = interp ? (unsigned)c.width * 2u : (unsigned)c.width;
    const float* __restrict__ depth = depth_all + (size_t)blockIdx.y * frame_px;
    float* __restrict__ out = out_all + (size_t)blockIdx.y * frame_px * 3;
    const unsigned groups = frame_px / 4;
    const unsigned stride = gridDim.x * kThreads;
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    float4* w = xch[wv];
    for (unsigned g0 = blockIdx.x * kThreads + wv * 64u; g0 < groups; g0 += stride) {   // wave-uniform
        const unsigned g = g0 + lane;
        float r[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (g < groups) {
            const float4 z4 = ld4(reinterpret_cast<const float4*>(depth) + g, NT);
            const float z[4] = {z4.x, z4.y, z4.z, z4.w};
            unsigned y = (g * 4) / row, x = (g * 4) - y * row;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                p2r_one(c, interp, x, y, z[k], r + 3 * k);
                if (++x == row) {
                    x = 0;
                    ++y;
                }
            }
        }
        w[lane * 3 + 0] = make_float4(r[0], r[1], r[2], r[3]);
        w[lane * 3 + 1] = make_float4(r[4], r[5], r[6], r[7]);
        w[lane * 3 + 2] = make_float4(r[8], r[9], r[10], r[11]);
        wave_exchange_fence();
        float4* o = reinterpret_cast<float4*>(out) + (size_t)g0 * 3;
        const unsigned nvec = (groups - g0 < 64u ? groups - g0 : 64u) * 3u;
#pragma unroll
        for (unsigned j = 0; j < 3; j++) {
            const unsigned idx = j * 64u + lane;
            if (idx < nvec) st4(o + idx, w[idx], NT);
        }
        wave_exchange_fence();
    }
    for (unsigned i = groups * 4 + blockIdx.x * kThreads + threadIdx.x; i < frame_px; i += stride) {
        float r[3];
        p2r_one(c, interp, i % row, i / row, depth[i], r);
        out[3 * (size_t)i] = r[0];
        out[3 * (size_t)i + 1] = r[1];
        out[3 * (size_t)i + 2] = r[2];
    }
}

// float3 -> float3 maps; 4 points (48 B) per thread, loads and stores both transposed through LDS.
template <bool NT>
__global__ __launch_bounds__(kThreads) void points_map_kernel(Camera c, size_t total, const float* __restrict__ in,
                                                             float* __restrict__ out, int to_projective)
{
    __shared__ float4 xch[kThreads / 64][3 * 64];
    const size_t groups = total / 4;
    const size_t stride = (size_t)gridDim.x * kThreads;
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    float4* w = xch[wv];
    auto map = [&](float& x, float& y, float& z) {
        if (!to_projective) {
            convert_ptr(x, y, z, c);          // DimensionConvertor.cu:25-33
        } else {
            // convert_rtp, DimensionConvertor.h:128-147
            if (fabsf(z) < 1.0f) {
                x = -1.0f;
                y = -1.0f;
            } else {
                float ox = x / z, oy = y / z;
                ox *= c.fx;
                oy *= c.fy;
                x = ox + (float)c.cx;
                y = (float)c.cy - oy;
            }
        }
    };
    for (size_t g0 = (size_t)blockIdx.x * kThreads + wv * 64u; g0 < groups; g0 += stride) {   // wave-uniform
        const unsigned nvec = (unsigned)(groups - g0 < 64 ? groups - g0 : 64) * 3u;
        const float4* p = reinterpret_cast<const float4*>(in) + g0 * 3;
#pragma unroll
        for (unsigned j = 0; j < 3; j++) {
            const unsigned idx = j * 64u + lane;
            w[idx] = idx < nvec ? ld4(p + idx, NT) : make_float4(0.f, 0.f, 1.f, 0.f);
        }
        wave_exchange_fence();
        const float4 a = w[lane * 3], b = w[lane * 3 + 1], d = w[lane * 3 + 2];
        float r[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 4; k++) map(r[3 * k], r[3 * k + 1], r[3 * k + 2]);
        wave_exchange_fence();
        w[lane * 3 + 0] = make_float4(r[0], r[1], r[2], r[3]);
        w[lane * 3 + 1] = make_float4(r[4], r[5], r[6], r[7]);
        w[lane * 3 + 2] = make_float4(r[8], r[9], r[10], r[11]);
        wave_exchange_fence();
        float4* o = reinterpret_cast<float4*>(out) + g0 * 3;
#pragma unroll
        for (unsigned j = 0; j < 3; j++) {
            const unsigned idx = j * 64u + lane;
            if (idx < nvec) st4(o + idx, w[idx], NT);
        }
        wave_exchange_fence();
    }
    for (size_t i = groups * 4 + (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
        float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        map(x, y, z);
        out[3 * i] = x;
        out[3 * i + 1] = y;
        out[3 * i + 2] = z;
    }
}

// scalar forms: one pixel per thread, dword accesses -- any pointer alignment, any frame size
__global__ __launch_bounds__(kThreads) void p2r_depth_scalar_kernel(Camera c, size_t total, const float* __restrict__ depth,
                                                                   float* __restrict__ out, int interp)
{
    const size_t frame_px = (size_t)c.width * c.height;
    const unsigned row = interp ? (unsigned)c.width * 2u : (unsigned)c.width;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
        const unsigned q = (unsigned)(i % frame_px);
        float r[3];
        p2r_one(c, interp, q % row, q / row, depth[i], r);
        out[3 * i] = r[0];
        out[3 * i + 1] = r[1];
        out[3 * i + 2] = r[2];
    }
}

__global__ __launch_bounds__(kThreads) void points_map_scalar_kernel(Camera c, size_t total, const float* __restrict__ in,
                                                                    float* __restrict__ out, int to_projective)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
        float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
        if (!to_projective) {
            convert_ptr(x, y, z, c);
        } else if (fabsf(z) < 1.0f) {
            x = -1.0f;
            y = -1.0f;
        } else {
            float ox = x / z, oy = y / z;
            ox *= c.fx;
            oy *= c.fy;
            x = ox + (float)c.cx;
            y = (float)c.cy - oy;
        }
        out[3 * i] = x;
        out[3 * i + 1] = y;
        out[3 * i + 2] = z;
    }
}

// ---- Buffer2D -------------------------------------------------------------------------------
__device__ __forceinline__ int f2i_rz(float v) { return (int)v; }   // v_cvt_i32_f32: RZ, saturating, NaN -> 0

__device__ __forceinline__ void update_weighted_depth(float& rd, float& rw, float d)
{
    // updateWaitedDepth, ArrayBuffer/Buffer2D.cu:13-30
    if (d > 50.0f) {
        if (rd != 0.0f) {
            int diff = f2i_rz(rd) - f2i_rz(d);
            if (diff < 0) diff = -diff;
            if ((float)diff < d * 0.01f) {
                rd = ((rd * (rw + 1.0f)) + (d * rw)) / (rw * 2.0f + 1.0f);
                rw = rw + 1.0f;
            }
        } else {
            rd = d;
            rw = 1.0f;
        }
    }
}

__global__ __launch_bounds__(kThreads) void buf_fill_kernel(float4* __restrict__ buf, size_t n4, float2* tail, size_t ntail)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += stride)
        buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < ntail; i += stride) tail[i] = make_float2(0.f, 0.f);
}

// two records (16 B) per thread
__global__ __launch_bounds__(kThreads) void buf_insert_depth_kernel(kde_weighted_d* __restrict__ buf,
                                                                   const float* __restrict__ d, size_t n)
{
    const size_t pairs = n / 2;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < pairs; g += stride) {
        const float2 v = reinterpret_cast<const float2*>(d)[g];
        reinterpret_cast<float4*>(buf)[g] = make_float4(v.x, 1.0f, v.y, 1.0f);   // Buffer2D.cu:45-48
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) {
        buf[n - 1].d = d[n - 1];
        buf[n - 1].w = 1.0f;
    }
}

__global__ __launch_bounds__(kThreads) void buf_insert_float2_kernel(kde_weighted_d* __restrict__ buf,
                                                                    const float2* __restrict__ xy, int width, int height)
{
    const size_t n = (size_t)width * height;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / width);
        kde_weighted_d r;
        r.d = xy[i].x;
        r.w = (float)y;   // sic: Buffer2D.cu:137 stores the row index
        buf[i] = r;
    }
}

__global__ __launch_bounds__(kThreads) void buf_get_kernel(const kde_weighted_d* __restrict__ buf, float* __restrict__ out,
                                                          size_t n, int which)
{
    const size_t pairs = n / 2;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < pairs; g += stride) {
        const float4 v = reinterpret_cast<const float4*>(buf)[g];
        reinterpret_cast<float2*>(out)[g] = which ? make_float2(v.y, v.w) : make_float2(v.x, v.z);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) out[n - 1] = which ? buf[n - 1].w : buf[n - 1].d;
}

// updateData over n_frames consecutive frames fused into one read-modify-write of the buffer.
// Four records (32 B) per thread, depth frames read as float4, two frames in flight.  The rule is evaluated
// branch-free on pairs of records: the float arithmetic is the reference's own operations in its own order
// (v_pk_mul / v_pk_add / IEEE division; rw*2+1 as one fma is exact because rw*2 is), selected at the end.
typedef float s_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void update_pair(s_f2& rd, s_f2& rw, s_f2 d)
{
    const s_f2 one = {1.0f, 1.0f};
    const s_f2 rw1 = rw + one;
    const s_f2 sum = (rd * rw1) + (d * rw);                                     // ((ref.d*(w+1)) + (d*w))
    const s_f2 den = __builtin_elementwise_fma(rw, s_f2{2.0f, 2.0f}, one);     // (w*2+1), exact as an fma
    const s_f2 lim = d * s_f2{0.01f, 0.01f};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float q = sum[k] / den[k];
        const unsigned diff = (unsigned)f2i_rz(rd[k]) - (unsigned)f2i_rz(d[k]);
        const int ad = (int)diff < 0 ? (int)(0u - diff) : (int)diff;             // abs() as the reference's, INT_MIN stays
        const bool valid = d[k] > 50.0f;
        const bool has = rd[k] != 0.0f;
        const bool avg = valid && has && ((float)ad < lim[k]);
        const bool set = valid && !has;
        rd[k] = avg ? q : (set ? d[k] : rd[k]);
        rw[k] = avg ? rw1[k] : (set ? 1.0f : rw[k]);
    }
}

template <bool NT>
__global__ __launch_bounds__(kThreads) void buf_update4_kernel(kde_weighted_d* __restrict__ buf, const float* __restrict__ d,
                                                              size_t n, int n_frames)
{
    const size_t quads = n / 4;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < quads; g += stride) {
        float4* bp = reinterpret_cast<float4*>(buf) + 2 * g;
        const float4 r0 = bp[0], r1 = bp[1];
        s_f2 rdA = {r0.x, r0.z}, rwA = {r0.y, r0.w}, rdB = {r1.x, r1.z}, rwB = {r1.y, r1.w};
        const float4* dp = reinterpret_cast<const float4*>(d) + g;
        const size_t fstride = n / 4;                      // frames are n floats apart (n % 4 == 0 on this path)
        float4 cur = ld4(dp, NT);
        for (int f = 0; f < n_frames; f++) {
            const float4 nxt = f + 1 < n_frames ? ld4(dp + (size_t)(f + 1) * fstride, NT) : cur;
            update_pair(rdA, rwA, s_f2{cur.x, cur.y});
            update_pair(rdB, rwB, s_f2{cur.z, cur.w});
            cur = nxt;
        }
        bp[0] = make_float4(rdA.x, rwA.x, rdA.y, rwA.y);
        bp[1] = make_float4(rdB.x, rwB.x, rdB.y, rwB.y);
    }
}

// scalar form: one record per thread, any alignment / size (also the tail of the vector form)
__global__ __launch_bounds__(kThreads) void buf_update_scalar_kernel(kde_weighted_d* __restrict__ buf, const float* __restrict__ d,
                                                                    size_t first, size_t n, int n_frames)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = first + (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        float rd = buf[i].d, rw = buf[i].w;
        for (int f = 0; f < n_frames; f++) update_weighted_depth(rd, rw, d[(size_t)f * n + i]);
        buf[i].d = rd;
        buf[i].w = rw;
    }
}

__global__ __launch_bounds__(kThreads) void buf_insert_depth_scalar_kernel(kde_weighted_d* __restrict__ buf, const float* __restrict__ d,
                                                                          size_t n)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        buf[i].d = d[i];
        buf[i].w = 1.0f;
    }
}

__global__ __launch_bounds__(kThreads) void buf_insert_float2_scalar_kernel(kde_weighted_d* __restrict__ buf, const float* __restrict__ xy,
                                                                           int width, int height)
{
    const size_t n = (size_t)width * height;
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
        buf[i].d = xy[2 * i];
        buf[i].w = (float)(int)(i / width);   // sic: Buffer2D.cu:137 stores the row index
    }
}

__global__ __launch_bounds__(kThreads) void buf_get_scalar_kernel(const kde_weighted_d* __restrict__ buf, float* __restrict__ out,
                                                                 size_t n, int which)
{
    const size_t stride = (size_t)gridDim.x * kThreads;
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) out[i] = which ? buf[i].w : buf[i].d;
}

}  // namespace

// vector forms need 16-byte aligned pointers, and for a batch frames that start on 16-byte boundaries (W*H % 4 == 0)
static bool p2r_vector_ok(const Camera& c, int n, const void* in, const void* out)
{
    return aligned(in, 16) && aligned(out, 16) && (n == 1 || ((size_t)c.width * c.height) % 4 == 0);
}

static int launch_p2r_any(const Camera& c, int n, const float* depth, kde_float3* out, int interp, hipStream_t s)
{
    const size_t frame_px = (size_t)c.width * c.height;
    const bool streaming = frame_px * n * 16 > kCacheBytes;
    if (p2r_vector_ok(c, n, depth, out) && streaming)
        hipLaunchKernelGGL(p2r_depth_kernel<true>, dim3(grid_for(frame_px / 4 + 1), n), dim3(kThreads), 0, s, c, depth,
                           reinterpret_cast<float*>(out), interp);
    else if (p2r_vector_ok(c, n, depth, out))
        hipLaunchKernelGGL(p2r_depth_kernel<false>, dim3(grid_for(frame_px / 4 + 1), n), dim3(kThreads), 0, s, c, depth,
                           reinterpret_cast<float*>(out), interp);
    else
        hipLaunchKernelGGL(p2r_depth_scalar_kernel, dim3(grid_for(frame_px * n)), dim3(kThreads), 0, s, c, frame_px * n, depth,
                           reinterpret_cast<float*>(out), interp);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

int launch_p2r_depth(const Camera& c, int n, const float* depth, kde_float3* out, hipStream_t s)
{
    return launch_p2r_any(c, n, depth, out, 0, s);
}

int launch_p2r_interp(const Camera& c, int n, const float* depth, kde_float3* out, hipStream_t s)
{
    return launch_p2r_any(c, n, depth, out, 1, s);
}

static int launch_points_any(const Camera& c, int n, const kde_float3* in, kde_float3* out, int to_projective, hipStream_t s)
{
    const size_t total = (size_t)c.width * c.height * n;
    const bool streaming = total * 24 > kCacheBytes;
    if (aligned(in, 16) && aligned(out, 16) && streaming)
        hipLaunchKernelGGL(points_map_kernel<true>, dim3(grid_for(total / 4 + 1)), dim3(kThreads), 0, s, c, total,
                           reinterpret_cast<const float*>(in), reinterpret_cast<float*>(out), to_projective);
    else if (aligned(in, 16) && aligned(out, 16))
        hipLaunchKernelGGL(points_map_kernel<false>, dim3(grid_for(total / 4 + 1)), dim3(kThreads), 0, s, c, total,
                           reinterpret_cast<const float*>(in), reinterpret_cast<float*>(out), to_projective);
    else
        hipLaunchKernelGGL(points_map_scalar_kernel, dim3(grid_for(total)), dim3(kThreads), 0, s, c, total,
                           reinterpret_cast<const float*>(in), reinterpret_cast<float*>(out), to_projective);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

int launch_p2r_points(const Camera& c, int n, const kde_float3* in, kde_float3* out, hipStream_t s)
{
    return launch_points_any(c, n, in, out, 0, s);
}

int launch_r2p(const Camera& c, int n, const kde_float3* in, kde_float3* out, hipStream_t s)
{
    return launch_points_any(c, n, in, out, 1, s);
}

int launch_buf_init(kde_weighted_d* buf, size_t n, hipStream_t s)
{
    const size_t n4 = n / 2;
    hipLaunchKernelGGL(buf_fill_kernel, dim3(grid_for(n4 + 1)), dim3(kThreads), 0, s, reinterpret_cast<float4*>(buf), n4,
                       reinterpret_cast<float2*>(buf) + n4 * 2, n - n4 * 2);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

int launch_buf_insert_depth(kde_weighted_d* buf, const float* d, size_t n, hipStream_t s)
{
    if (aligned(d, 8))
        hipLaunchKernelGGL(buf_insert_depth_kernel, dim3(grid_for(n / 2 + 1)), dim3(kThreads), 0, s, buf, d, n);
    else
        hipLaunchKernelGGL(buf_insert_depth_scalar_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, buf, d, n);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

int launch_buf_insert_float2(kde_weighted_d* buf, const float* xy, int width, int height, hipStream_t s)
{
    if (aligned(xy, 8))
        hipLaunchKernelGGL(buf_insert_float2_kernel, dim3(grid_for((size_t)width * height)), dim3(kThreads), 0, s, buf,
                           reinterpret_cast<const float2*>(xy), width, height);
    else
        hipLaunchKernelGGL(buf_insert_float2_scalar_kernel, dim3(grid_for((size_t)width * height)), dim3(kThreads), 0, s, buf, xy,
                           width, height);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

int launch_buf_get(const kde_weighted_d* buf, float* out, size_t n, int which, hipStream_t s)
{
    if (aligned(out, 8))
        hipLaunchKernelGGL(buf_get_kernel, dim3(grid_for(n / 2 + 1)), dim3(kThreads), 0, s, buf, out, n, which);
    else
        hipLaunchKernelGGL(buf_get_scalar_kernel, dim3(grid_for(n)), dim3(kThreads), 0, s, buf, out, n, which);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

int launch_buf_update(kde_weighted_d* buf, const float* d, size_t n, int n_frames, hipStream_t s)
{
    // vector form: float4 depth loads need a 16-byte aligned pointer and, for a sequence, frames n floats apart that
    // stay aligned (n % 4 == 0); its last n % 4 records, or everything otherwise, go through the scalar form
    const bool vec = aligned(d, 16) && (n_frames == 1 || n % 4 == 0) && n >= 4;
    const size_t done = vec ? (n / 4) * 4 : 0;
    // (non-temporal depth loads were measured on the 32-frame 1080p sequence: 5.62 vs 5.77 TB/s with default caching)
    if (vec) hipLaunchKernelGGL(buf_update4_kernel<false>, dim3(grid_for(n / 4)), dim3(kThreads), 0, s, buf, d, n, n_frames);
    if (done < n)
        hipLaunchKernelGGL(buf_update_scalar_kernel, dim3(grid_for(n - done)), dim3(kThreads), 0, s, buf, d, done, n, n_frames);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

}  // namespace kde
