// kde_internal.h — shared host-side plumbing of libkde_hip.so (not part of the public ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/kde_hip.h"
#include "kde_host_math.h"   // exp_zero_threshold, spatial_table, smallest_*_reaching (pure C++)

namespace kde {

// ---- error reporting -----------------------------------------------------------------------
int fail(int code, const char* fmt, ...);   // records the message for kde_last_error_string() and returns code

#define KDE_HIP_TRY(expr)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::kde::fail(KDE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                           \
    } while (0)

#define KDE_TRY(expr)                \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != KDE_OK) return rc_; \
    } while (0)

// A/B switches (environment variables that select a rejected or alternative kernel) exist only in the measurement build
// tools/hooks/libkde_hip_ab.so (-DKDE_AB_SWITCHES): the product library reads no environment variable and carries none of
// the kernels that were measured slower and dropped.  KDE_AB_ENV(name) is getenv there and a constant nullptr here.
#ifdef KDE_AB_SWITCHES
#include <cstdlib>
#define KDE_AB_ENV(name) (::getenv(name))
#define KDE_AB(...) __VA_ARGS__
#else
#define KDE_AB_ENV(name) (static_cast<const char*>(nullptr))
#define KDE_AB(...)
#endif

#define KDE_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::kde::fail(KDE_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// ---- stage hooks -----------------------------------------------------------------------------
// Compiled ONLY into tools/hooks/libkde_hip_stage.so (-DKDE_STAGE_HOOKS, include/kde_test_hooks.h): the same
// sources as the product library plus per-pixel dumps of K1's / K10's intermediate stages, a switch that forces the
// full-rule bodies, and counters of the body each tile ran.  The product library contains none of it.
#ifdef KDE_STAGE_HOOKS
struct StageCtl {
    float* jbf_avg;          // [n][H][W]: first-pass average of K1 as pass 2 uses it (NaN where the sum of weights is 0)
    float* ers_avg;          // [H][W]: K10 pass-1 average (NaN where the sum of weights is 0)
    float* ers_dev;          // [H][W]: K10 pass-2 mean absolute deviation (undefined where ers_avg is NaN)
    unsigned* counters;      // [8]: K1 tiles per body (colour rule + 2 * depth rule), K10 tiles (small-a, not, no depth rule, depth rule)
    int force_full_rules;    // != 0: every tile runs the body with both Q1 rules / the per-pixel deviation pass
};
extern StageCtl g_stage;
#define KDE_STAGE(...) __VA_ARGS__
#else
#define KDE_STAGE(...)
#endif

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Device buffer with RAII; never throws.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int alloc(size_t count)
    {
        release();
        if (count == 0) return KDE_OK;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(KDE_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        }
        n = count;
        return KDE_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Pinned host buffer (the reference's cudaMallocHost'ed *_Host members), allocated lazily.
template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    int ensure(size_t count)
    {
        if (p && n >= count) return KDE_OK;
        release();
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), count * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(KDE_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        }
        n = count;
        return KDE_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
    ~PinnedBuf() { release(); }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
};

// ---- launchers implemented in the .hip translation units ------------------------------------
struct JbfLaunch {
    int width, height, n;
    int window;
    const float* depth;        // [n][H][W]
    const uint8_t* guide;      // [n][H][W][3]
    float* out;                // [n][H][W]
    const float* s_eff;        // device, window^2 (zeros replaced by 1)
    const float* table_host;   // host, window^2 (calcSpatialFilter as computed)
    const float* log2_pk_dev;  // device, 2 * window^2: the packed kernels' log2(S) pairs, windows > 21 only (jbf_fast.hip)
    float spatial_sigma, color_sigma, depth_sigma;
    float color_den, depth_den;    // 2*sigma^2 as the reference forms it (float)
    int cd_skip;               // colour factor skipped (== underflow to 0) when cd >= cd_skip
    float d2_skip;             // depth factor skipped when (d_q - wavg)^2 >= d2_skip
    int variant;
};
int launch_jbf(const JbfLaunch& a, hipStream_t s);
int jbf_active_variant(const JbfLaunch& a);
// tuned variants (jbf_fast.hip); variant ids there are 0-based, the public id is +1 (0 = generic kernel)
int jbf_fast_variant_count();
const char* jbf_fast_variant_name(int v);
int jbf_fast_variant_window(int v);
int jbf_fast_default_variant(const JbfLaunch& l);
bool jbf_fast_supported(const JbfLaunch& l);
void jbf_fast_fill_table(int window, const float* table_host, bool packed, float* out);
bool jbf_fast_needs_device_table(int window);
int launch_jbf_fast(const JbfLaunch& l, int variant, const float* table_host, hipStream_t s);
int jbf_variant_count();
const char* jbf_variant_name(int v);

struct PresmoothLaunch {
    int width, height, n;
    int radius;                // ksz/2 after OpenCV's adjustment
    const uint8_t* src;        // [n][H][W][3]
    uint8_t* dst;
    const float* lut;          // device, (radius^2+1) x 766 weights
    long long grid_cap;        // persistent-grid size for the tuned kernels (presmooth_resident_blocks at handle creation)
};
int launch_presmooth(const PresmoothLaunch& a, hipStream_t s);
long long presmooth_resident_blocks(int radius);   // on the current device; 0 for radii served by the generic kernel

struct MrfLaunch {
    int width, height, n, window;
    const float* depth;
    const uint8_t* bgr;
    float* out;
    float color_sigma, smooth_sigma;
};
int launch_mrf(const MrfLaunch& a, hipStream_t s);

struct Camera {
    float fx, fy;
    int cx, cy;
    int width, height;
};
int launch_p2r_depth(const Camera& c, int n, const float* depth, kde_float3* out, hipStream_t s);
int launch_p2r_points(const Camera& c, int n, const kde_float3* in, kde_float3* out, hipStream_t s);
int launch_p2r_interp(const Camera& c, int n, const float* depth, kde_float3* out, hipStream_t s);
int launch_r2p(const Camera& c, int n, const kde_float3* in, kde_float3* out, hipStream_t s);

int launch_buf_init(kde_weighted_d* buf, size_t n, hipStream_t s);
int launch_buf_insert_depth(kde_weighted_d* buf, const float* d, size_t n, hipStream_t s);
int launch_buf_insert_float2(kde_weighted_d* buf, const float* xy, int width, int height, hipStream_t s);
int launch_buf_get(const kde_weighted_d* buf, float* out, size_t n, int which, hipStream_t s);
int launch_buf_update(kde_weighted_d* buf, const float* d, size_t n, int n_frames, hipStream_t s);

struct DaspGeom {
    int width, height, rows, cols, wx, wy;
};
// n = frames of a batch (per-frame colour / cloud / cluster tables / outputs, back to back); the single-frame forms take n = 1
int launch_dasp_sample(const DaspGeom& g, int n, const uint8_t* bgr, const kde_float3* pts, kde_superpixel* mean,
                       kde_float3* centers, kde_superpixel* mean2, kde_float3* centers2, hipStream_t s);
int launch_dasp_calc_ld(const DaspGeom& g, const uint8_t* bgr, const kde_float3* pts, kde_label_distance* ld,
                        const kde_superpixel* mean, const kde_float3* centers, int32_t* labels, float color_sigma,
                        float spatial_sigma, float depth_sigma, bool first, hipStream_t s);
int launch_dasp_calc_ld_dual(const DaspGeom& g, int n, const uint8_t* bgr, const kde_float3* pts, kde_label_distance* ld_a,
                             const kde_superpixel* mean_a, const kde_float3* centers_a, int32_t* labels_a,
                             const float sig_a[3], kde_label_distance* ld_b, const kde_superpixel* mean_b,
                             const kde_float3* centers_b, int32_t* labels_b, const float sig_b[3], bool first, bool write_ld,
                             hipStream_t s);
int launch_dasp_analyze(const DaspGeom& g, const uint8_t* bgr, const kde_float3* pts, const int32_t* labels,
                        kde_superpixel* mean, kde_float3* centers, const float* intr_dev, hipStream_t s);
int launch_dasp_analyze_dual(const DaspGeom& g, int n, const uint8_t* bgr, const kde_float3* pts, const int32_t* labels_a,
                             kde_superpixel* mean_a, kde_float3* centers_a, const int32_t* labels_b, kde_superpixel* mean_b,
                             kde_float3* centers_b, const float* intr_dev, hipStream_t s);

int launch_ers_edge_phase(int width, int height, int dir, int window, const int32_t* color_labels, const int32_t* l0,
                          const float* d0, int32_t* l1, float* d1, hipStream_t s);
int launch_ers_edge_refining(int width, int height, int n, int window, const int32_t* color_labels, const int32_t* l0,
                             const float* d0, int32_t* scratch_l, float* scratch_d, int32_t* l2, float* d2, bool two_launches,
                             hipStream_t s);
int launch_ers_enhance(int width, int height, int n, const float* rd, const uint8_t* bgr, const int32_t* labels,
                       const float* s_eff, const float* table_host, int window, float color_sigma, float depth_sigma,
                       float exp_zero, float* out, int variant, hipStream_t s);

int launch_spdsr_init_normalized(const Camera& c, float* nxy, hipStream_t s);
int launch_spdsr_cluster_planes(int width, int height, int n, int nclusters, int table_frames, const int32_t* labels,
                                const kde_float3* pts, double* sums, double* cov, float* nd, int* moments_dirty, hipStream_t s);
// state of the resident sweep launches of one SPDSR handle (spdsr_kernels.hip: mrf_sweeps_resident_kernel)
struct SpdsrResident {
    int* flags = nullptr;       // device: one announcement counter per workgroup, a 128-byte line each
    int* status = nullptr;      // pinned host: 0, or (workgroup + 1) that gave up waiting
    int gen = 0;                // the counters run on across calls
    int cus = 0, cooperative = 0;
};
int spdsr_resident_init(SpdsrResident* r);
void spdsr_resident_release(SpdsrResident* r);
int launch_spdsr_plane_projection(int width, int height, int n, int nclusters, const float* nd, const int32_t* labels,
                                  const kde_float3* pts, const float* nxy, kde_float3* plane_fitted, kde_float3* opt_a,
                                  kde_float3* opt_b, int sweeps, kde_float3** result, SpdsrResident* res, hipStream_t s);

}  // namespace kde
