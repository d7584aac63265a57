// ers_kernels.hip — EdgeRefinedSuperpixel on gfx950 (K9 edge_refining, K10 depthmap_enhancement).
// Reference: EdgeRefinedSuperpixel/EdgeRefinedSuperpixel.cu:4-102 and :104-205.
//
// K9 is an in-place scatter with cross-thread races in the reference.  It is re-formulated as a
// race-free GATHER with the snapshot semantics D2: in each phase (horizontal, then
// vertical on the horizontal result) every source pixel evaluates its rule on the phase-start
// labels/depth, its depth-zeroing cascade sees only its own writes, and write-sets are applied in
// raster order of the source (later source wins).  A source at s writes at most [s-2, s+2] along the
// scan, so output p inspects sources p+2 ... p-2 (descending = last writer first).
// K10 reads the K9 result and writes a separate buffer (D3).
#include "kde_internal.h"

namespace kde {
namespace {

// -------------------------------------------------------------------------------------------------
// K9, one phase.  AT(k) addresses coordinate k along the scan line of this thread.
// -------------------------------------------------------------------------------------------------
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
#define AT(arr, k) ((arr)[base + (size_t)(k) * stride])
    const int half = window / 2;

    int out_label = AT(L0, pos);
    bool have_label = false;
    bool zero = false;
    for (int s = pos + 2; s >= pos - 2; s--) {
        if (s < 0 || s + 1 >= len) continue;
        if (AT(L0, s) == AT(L0, s + 1)) continue;
        const int cur = AT(color_labels, s);
        // search outwards, left candidate first (.cu:25-55)
        int branch = 0, tp = 0;   // 1 = found on the left at tp, 2 = found on the right at tp
        for (int d = 1; d <= half && branch == 0; d++) {
            if (s - d < 0 && s + d >= len) break;
            if (s - d >= 0 && AT(color_labels, s - d) != cur) {
                branch = 1;
                tp = s - d;
            } else if (s + d < len && AT(color_labels, s + d) != cur) {
                branch = 2;
                tp = s + d;
            }
        }
        if (branch == 1) {
            // pixels tp+1 .. s take the label right of the source (.cu:32-38)
            if (pos >= tp + 1 && pos <= s) {
                if (!have_label) {
                    out_label = AT(L0, s + 1);
                    have_label = true;
                }
                const float dp = AT(D0, pos);
                if (fabsf(dp - AT(D0, pos + 1)) > dp * 0.1f) zero = true;
            }
        } else if (branch == 2) {
            // pixels s+1 .. tp-1 take the source's label; the depth test cascades (.cu:45-51)
            if (pos >= s + 1 && pos <= tp - 1) {
                if (!have_label) {
                    out_label = AT(L0, s);
                    have_label = true;
                }
                float prev = AT(D0, s);
                bool z = false;
                for (int i = s + 1; i <= pos; i++) {
                    float c = AT(D0, i);
                    z = fabsf(c - prev) > c * 0.1f;
                    if (z) c = 0.0f;
                    prev = c;
                }
                if (z) zero = true;
            }
        }
    }
    AT(L1, pos) = out_label;
    AT(D1, pos) = zero ? 0.0f : AT(D0, pos);
#undef AT
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
                    if (!(xarg >= a.exp_zero)) filter *= expf(-xarg);
                }
                w_average += dq * filter;
                weight += filter;
            }
        }
    float result = 0.0f;
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
                        if (!(xarg >= a.exp_zero)) filter *= expf(-xarg);
                    }
                    if (a.depth_sigma != 0.0f) {
                        const float dd = dq - w_average;
                        const float xd = (dd * dd) / dden;
                        if (!(xd >= a.exp_zero)) filter *= expf(-xd);
                    }
                    numerator += dq * filter;
                    denominator += filter;
                }
            }
        result = (denominator == 0.0f) ? 0.0f : numerator / denominator;
    }
    a.out[(size_t)y * a.width + x] = result;
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

int launch_ers_enhance(int width, int height, const float* rd, const uint8_t* bgr, const int32_t* labels,
                       const float* s_eff, int window, float color_sigma, float depth_sigma, float exp_zero,
                       float* out, hipStream_t s)
{
    EnhDev d{rd, bgr, labels, s_eff, out, width, height, window, color_sigma, depth_sigma, exp_zero};
    const int R = window / 2;
    const size_t lds = (size_t)(kTileX + 2 * R) * (kTileY + 2 * R) * 12 + (size_t)window * window * 4;
    hipLaunchKernelGGL(enhance_kernel, dim3(ceil_div(width, kTileX), ceil_div(height, kTileY)), dim3(kThreads), lds, s, d);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

}  // namespace kde
