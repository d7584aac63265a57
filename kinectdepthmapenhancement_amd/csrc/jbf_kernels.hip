// jbf_kernels.hip — K0 (colour pre-smoothing), K1 (joint bilateral depth filter) and the sibling
// MRF filter for gfx950.  Reference: JointBilateralFilter/JointBilateralFilter.cu:4-83, :283-290;
// cv::gpu::bilateralFilter (OpenCV 2.4.3 gpu) at the call site :285; MarkovRandomField.cu:4-40.
//
// Layout: one workgroup = one output tile of one frame (blockIdx.z = frame); depth and guide
// tiles (+halo) are staged once in LDS, out-of-image and invalid (<= 50 mm) taps are stored as
// depth 0 so the inner loops carry no bounds tests.  No MFMA: this is a stencil.
#include "kde_internal.h"

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
                if (a.color_on && cd < a.cd_skip) filter *= expf(-(float)cd / a.color_den);
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
                    if (a.color_on && cd < a.cd_skip) filter *= expf(-(float)cd / a.color_den);
                    const float dd = dq - w_average;
                    const float d2 = dd * dd;
                    // NaN d2 fails the '<' test on purpose: the reference multiplies the NaN in
                    if (a.depth_on && !(d2 >= a.d2_skip)) filter *= expf(-d2 / a.depth_den);
                    numerator += dq * filter;
                    denominator += filter;
                }
            }
        }
        result = (denominator == 0.0f) ? 0.0f : numerator / denominator;
    }
    a.out[frame + (size_t)y * a.width + x] = result;
}

// --------------------------------------------------------------------------------------------
// K0 — cv::gpu::bilateralFilter on packed 8UC3.  The weight exp(space2*ss + n1^2*sc) depends only
// on (space2, n1) with n1 = L1 colour distance in [0,765]; the host tabulates it with the same
// float expression the CPU restatement uses, so the u8 result is bit-identical by construction.
// --------------------------------------------------------------------------------------------
struct PreDev {
    const uint8_t* src;
    uint8_t* dst;
    const float* lut;
    int width, height, radius;
};

__global__ __launch_bounds__(kThreads) void presmooth_kernel(PreDev a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int r = a.radius;
    const int LW = kTileX + 2 * r, LH = kTileY + 2 * r;
    const int lut_n = (r * r + 1) * 766;
    float* lut = reinterpret_cast<float*>(smem);
    uint32_t* sc = reinterpret_cast<uint32_t*>(lut + lut_n);

    const size_t frame = (size_t)blockIdx.z * a.width * a.height;
    const uint8_t* __restrict__ src = a.src + frame * 3;
    const int x0 = blockIdx.x * kTileX, y0 = blockIdx.y * kTileY;
    const int tid = threadIdx.x;

    for (int i = tid; i < lut_n; i += kThreads) lut[i] = a.lut[i];
    for (int i = tid; i < LW * LH; i += kThreads) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = reflect101(x0 + lx - r, a.width);
        const int gy = reflect101(y0 + ly - r, a.height);
        sc[i] = load_bgrx(src, (size_t)gy * a.width + gx);
    }
    __syncthreads();

    const int tx = tid & (kTileX - 1), ty = tid / kTileX;
    const int x = x0 + tx, y = y0 + ty;
    if (x >= a.width || y >= a.height) return;

    const uint32_t cc = sc[(ty + r) * LW + tx + r];
    const int c0 = cc & 0xff, c1 = (cc >> 8) & 0xff, c2 = (cc >> 16) & 0xff;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, sum2 = 0.0f;
    for (int dy = -r; dy <= r; dy++) {
        for (int dx = -r; dx <= r; dx++) {
            const int space2 = dx * dx + dy * dy;
            if (space2 > r * r) continue;
            const uint32_t v = sc[(ty + r + dy) * LW + tx + r + dx];
            const int v0 = v & 0xff, v1 = (v >> 8) & 0xff, v2 = (v >> 16) & 0xff;
            const int n1 = abs(v0 - c0) + abs(v1 - c1) + abs(v2 - c2);
            const float w = lut[space2 * 766 + n1];
            s0 = s0 + w * (float)v0;
            s1 = s1 + w * (float)v1;
            s2 = s2 + w * (float)v2;
            sum2 = sum2 + w;
        }
    }
    auto sat = [](float v) -> uint8_t {
        if (!(v > 0.0f)) return 0;
        if (v >= 255.0f) return 255;
        return (uint8_t)rintf(v);
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
                if (a.color_sigma != 0.0f) color_filter = expf(-a.color_sigma * (float)cd);
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
    const int R = a.window / 2;
    const size_t lds = (size_t)(kTileX + 2 * R) * (kTileY + 2 * R) * 8 + (size_t)a.window * a.window * 4;
    dim3 grid(ceil_div(a.width, kTileX), ceil_div(a.height, kTileY), a.n);
    hipLaunchKernelGGL(jbf_generic_kernel, grid, dim3(kThreads), lds, s, d);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

int launch_presmooth(const PresmoothLaunch& a, hipStream_t s)
{
    PreDev d{a.src, a.dst, a.lut, a.width, a.height, a.radius};
    const int r = a.radius;
    const size_t lds = (size_t)(r * r + 1) * 766 * 4 + (size_t)(kTileX + 2 * r) * (kTileY + 2 * r) * 4;
    dim3 grid(ceil_div(a.width, kTileX), ceil_div(a.height, kTileY), a.n);
    hipLaunchKernelGGL(presmooth_kernel, grid, dim3(kThreads), lds, s, d);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

int launch_mrf(const MrfLaunch& a, hipStream_t s)
{
    MrfDev d{a.depth, a.bgr, a.out, a.width, a.height, a.window, a.color_sigma, a.smooth_sigma};
    const int R = a.window / 2;
    const size_t lds = (size_t)(kTileX + 2 * R) * (kTileY + 2 * R) * 8;
    dim3 grid(ceil_div(a.width, kTileX), ceil_div(a.height, kTileY), a.n);
    hipLaunchKernelGGL(mrf_kernel, grid, dim3(kThreads), lds, s, d);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

}  // namespace kde
