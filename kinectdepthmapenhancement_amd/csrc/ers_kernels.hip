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
                       const float* s_eff, const float* table_host, int window, float color_sigma, float depth_sigma,
                       float exp_zero, float* out, hipStream_t s)
{
    const float cden = 2 * (color_sigma * color_sigma);
    // tuned kernel: 7x7 window, both sigmas on, pass-1 colour factor can never underflow
    const bool tuned = window == 7 && color_sigma != 0.0f && depth_sigma != 0.0f && !(195075.0f / cden >= exp_zero);
    if (tuned) {
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
        for (int i = 0; i < 49; i++)   // S == 0 -> factor skipped -> log2 = 0
            d.ls[i] = table_host[i] == 0.0f ? 0.0f : (float)std::log2((double)table_host[i]);
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
        hipLaunchKernelGGL(enhance7_kernel, dim3(ceil_div(width, kTileX), ceil_div(height, kTileY)), dim3(kThreads), 0, s, d);
        KDE_HIP_TRY(hipGetLastError());
        return KDE_OK;
    }
    EnhDev d{rd, bgr, labels, s_eff, out, width, height, window, color_sigma, depth_sigma, exp_zero};
    const int R = window / 2;
    const size_t lds = (size_t)(kTileX + 2 * R) * (kTileY + 2 * R) * 12 + (size_t)window * window * 4;
    hipLaunchKernelGGL(enhance_kernel, dim3(ceil_div(width, kTileX), ceil_div(height, kTileY)), dim3(kThreads), lds, s, d);
    KDE_HIP_TRY(hipGetLastError());
    return KDE_OK;
}

}  // namespace kde
