// kde_api.cpp — the extern "C" surface of libkde_hip.so (include/kde_hip.h): handle objects that
// mirror the reference classes' members and ownership, argument validation, error codes.
#include "kde_internal.h"

#include <cfloat>
#include <climits>

namespace kde {

// ---- errors ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "no error";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#ifdef KDE_STAGE_HOOKS
StageCtl g_stage = {nullptr, nullptr, nullptr, nullptr, 0};
#endif

}  // namespace kde

using namespace kde;

// A handle belongs to the device that was current when it was created (its buffers live there).  Calls made while
// another device is current are rejected instead of launching kernels on the wrong device's memory.
static int current_device()
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return d;
}
#define KDE_ON_DEVICE(h, who)                                                                                              \
    do {                                                                                                                   \
        const int cur_ = current_device();                                                                                 \
        if (cur_ != (h)->device)                                                                                           \
            return fail(KDE_ERR_INVALID, "%s: the handle was created on device %d but device %d is current (kde_set_device)", \
                        who, (h)->device, cur_);                                                                           \
    } while (0)

#ifdef KDE_STAGE_HOOKS
// include/kde_test_hooks.h: exists only in tools/hooks/libkde_hip_stage.so
extern "C" int kde_stage_set(float* jbf_avg_dev, float* ers_avg_dev, float* ers_dev_dev, unsigned* counters_dev, int force_full_rules)
{
    g_stage.jbf_avg = jbf_avg_dev;
    g_stage.ers_avg = ers_avg_dev;
    g_stage.ers_dev = ers_dev_dev;
    g_stage.counters = counters_dev;
    g_stage.force_full_rules = force_full_rules;
    return KDE_OK;
}
#endif

// =====================================================================================================
// library
// =====================================================================================================
extern "C" int kde_abi_version(void) { return KDE_ABI_VERSION; }
extern "C" const char* kde_last_error_string(void) { return g_err; }

extern "C" int kde_device_count(int* count)
{
    KDE_REQUIRE(count, "kde_device_count: null argument");
    KDE_HIP_TRY(hipGetDeviceCount(count));
    return KDE_OK;
}

extern "C" int kde_set_device(int device)
{
    KDE_HIP_TRY(hipSetDevice(device));
    return KDE_OK;
}

extern "C" int kde_device_info(char* arch_buf, size_t arch_cap, int* cu_count)
{
    int dev = 0;
    KDE_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    KDE_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (arch_buf && arch_cap) snprintf(arch_buf, arch_cap, "%s", prop.gcnArchName);
    if (cu_count) *cu_count = prop.multiProcessorCount;
    return KDE_OK;
}

extern "C" int kde_device_pci_bus_id(char* buf, size_t cap)
{
    KDE_REQUIRE(buf && cap >= 13, "kde_device_pci_bus_id: a buffer of at least 13 bytes is required");
    int dev = 0;
    KDE_HIP_TRY(hipGetDevice(&dev));
    KDE_HIP_TRY(hipDeviceGetPCIBusId(buf, (int)std::min<size_t>(cap, 64), dev));
    return KDE_OK;
}

// =====================================================================================================
// JointBilateralFilter
// =====================================================================================================
struct kde_jbf {
    int device = -1;                // hipGetDevice() at creation
    int width = 0, height = 0, max_batch = 1;
    kde_jbf_params p{};
    std::vector<float> table;       // SpatialFilter_Host
    DevBuf<float> s_eff;            // SpatialFilter_Device (zeros replaced by 1: "skip the factor")
    DevBuf<float> log2_pk;          // windows 23..31: the packed kernels' log2(S) pairs (too large for the kernel-argument block)
    DevBuf<float> filtered;         // Filtered_Device
    DevBuf<uint8_t> smooth;         // smooth_Device
    DevBuf<float> pre_lut;          // K0 weight table
    PinnedBuf<float> filtered_host; // Filtered_Host
    int pre_radius = 0;
    long long pre_grid_cap = 0;     // persistent-grid size of K0 on the device this handle was created on
    float color_den = 0, depth_den = 0;
    int cd_skip = INT_MAX;
    float d2_skip = INFINITY;
    int variant = -1;
    int n_last = 0;                 // frames of the last call that wrote Filtered_Device (0: none yet)
};

extern "C" int kde_jbf_default_params(kde_jbf_params* p)
{
    KDE_REQUIRE(p, "kde_jbf_default_params: null argument");
    p->window_size = 5;              // JointBilateralFilter.cpp:3
    p->spatial_sigma = 70.0f;        // :4
    p->color_sigma = 50.0f;          // :5
    p->depth_sigma = 20.0f;          // :6
    p->presmooth = 1;                // JointBilateralFilter.cu:285
    p->presmooth_kernel_size = 5;
    p->presmooth_sigma_color = 30.0f;
    p->presmooth_sigma_spatial = 30.0f;
    return KDE_OK;
}

static int jbf_create_impl(kde_jbf** out, int width, int height, int max_batch, const kde_jbf_params* params);

extern "C" int kde_jbf_create(kde_jbf** out, int width, int height, int max_batch, const kde_jbf_params* params)
{
    // the only entry point that builds std::vectors: nothing may cross the C boundary (kde_hip.h: "never aborts")
    try {
        return jbf_create_impl(out, width, height, max_batch, params);
    } catch (const std::bad_alloc&) {
        if (out) *out = nullptr;
        return fail(KDE_ERR_NOMEM, "kde_jbf_create: out of host memory");
    } catch (...) {
        if (out) *out = nullptr;
        return fail(KDE_ERR_INVALID, "kde_jbf_create: unexpected exception");
    }
}

static int jbf_create_impl(kde_jbf** out, int width, int height, int max_batch, const kde_jbf_params* params)
{
    KDE_REQUIRE(out, "kde_jbf_create: null out");
    *out = nullptr;
    KDE_REQUIRE(width >= 1 && height >= 1 && (long long)width * height <= (1ll << 30), "kde_jbf_create: bad size %dx%d", width, height);
    KDE_REQUIRE(max_batch >= 1 && max_batch <= 65535, "kde_jbf_create: max_batch must be in 1..65535");
    kde_jbf_params p;
    kde_jbf_default_params(&p);
    if (params) p = *params;
    KDE_REQUIRE(p.window_size >= 1 && p.window_size <= 31 && (p.window_size & 1), "kde_jbf_create: window_size must be odd in 1..31");
    KDE_REQUIRE(p.spatial_sigma == p.spatial_sigma && p.color_sigma >= 0.0f && p.depth_sigma >= 0.0f && p.spatial_sigma != 0.0f,
                "kde_jbf_create: sigmas must be >= 0 (spatial != 0)");
    struct Guard {                      // frees the half-built handle on every early return and on an exception
        kde_jbf* h;
        ~Guard() { delete h; }
    } guard{new (std::nothrow) kde_jbf};
    kde_jbf* h = guard.h;
    if (!h) return fail(KDE_ERR_NOMEM, "kde_jbf_create: out of host memory");
    h->device = current_device();
    h->width = width;
    h->height = height;
    h->max_batch = max_batch;
    h->p = p;
    const int w = p.window_size;
    h->table.resize((size_t)w * w);
    spatial_table(w, p.spatial_sigma, h->table.data());
    std::vector<float> eff(h->table);
    for (float& v : eff)
        if (v == 0.0f) v = 1.0f;   // "if(spatial != 0) filter *= spatial" (JointBilateralFilter.cu:30-31)
    const size_t px = (size_t)width * height;
    int rc = h->s_eff.alloc(eff.size());
    if (rc == KDE_OK) rc = h->filtered.alloc(px * max_batch);
    if (rc == KDE_OK) rc = h->smooth.alloc(px * 3 * max_batch);
    if (rc == KDE_OK && hipMemcpy(h->s_eff.p, eff.data(), eff.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(KDE_ERR_HIP, "kde_jbf_create: table upload failed");
    if (rc == KDE_OK && jbf_fast_needs_device_table(w)) {
        std::vector<float> pk((size_t)2 * w * w);
        jbf_fast_fill_table(w, h->table.data(), /*packed=*/true, pk.data());
        rc = h->log2_pk.alloc(pk.size());
        if (rc == KDE_OK && hipMemcpy(h->log2_pk.p, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(KDE_ERR_HIP, "kde_jbf_create: log2 table upload failed");
    }
    // thresholds of the "factor == 0 -> skipped" rule (JointBilateralFilter.cu:32-33, 65-68)
    const float xz = exp_zero_threshold();
    h->color_den = 2 * (p.color_sigma * p.color_sigma);
    h->depth_den = 2.0f * (p.depth_sigma * p.depth_sigma);
    h->cd_skip = INT_MAX;
    if (p.color_sigma != 0.0f) h->cd_skip = smallest_cd_reaching(h->color_den, xz);   // 195076 = never reached
    h->d2_skip = p.depth_sigma != 0.0f ? smallest_q_reaching(h->depth_den, xz) : INFINITY;
    // K0 table: weight(space2, n1) = expf(space2*ss + n1^2*sc), the expression of OpenCV's kernel
    if (rc == KDE_OK && p.presmooth) {
        float sc_ = p.presmooth_sigma_color, ss_ = p.presmooth_sigma_spatial;
        sc_ = (sc_ <= 0) ? 1 : sc_;
        ss_ = (ss_ <= 0) ? 1 : ss_;
        int radius = (p.presmooth_kernel_size <= 0) ? (int)rint((double)ss_ * 1.5) : p.presmooth_kernel_size / 2;
        radius = radius > 1 ? radius : 1;
        // radii 1..4 have tuned kernels (LDS-resident weight table); larger ones (the OpenCV function takes any kernel
        // size) run the generic kernel with the table in global memory.  64 bounds the table at 12.5 MB.
        if (radius > 64) return fail(KDE_ERR_UNSUPPORTED, "kde_jbf_create: pre-smoothing radius %d > 64", radius);
        h->pre_radius = radius;
        h->pre_grid_cap = presmooth_resident_blocks(radius);
        const float ss = -0.5f / (ss_ * ss_), sc = -0.5f / (sc_ * sc_);
        std::vector<float> lut((size_t)(radius * radius + 1) * 766);
        for (int s2 = 0; s2 <= radius * radius; s2++)
            for (int n1 = 0; n1 < 766; n1++) {
                const float fn = (float)n1;
                lut[(size_t)s2 * 766 + n1] = expf((float)s2 * ss + (fn * fn) * sc);
            }
        rc = h->pre_lut.alloc(lut.size());
        if (rc == KDE_OK && hipMemcpy(h->pre_lut.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(KDE_ERR_HIP, "kde_jbf_create: lut upload failed");
    }
    if (rc != KDE_OK) return rc;
    guard.h = nullptr;
    *out = h;
    return KDE_OK;
}

extern "C" int kde_jbf_destroy(kde_jbf* h)
{
    delete h;
    return KDE_OK;
}

static void jbf_fill_launch(const kde_jbf* h, JbfLaunch& a);

static int jbf_filter(kde_jbf* h, int n, const float* depth, const uint8_t* guide, float* out, hipStream_t s)
{
    JbfLaunch a;
    jbf_fill_launch(h, a);
    a.n = n;
    a.depth = depth;
    a.guide = guide;
    a.out = out;
    return launch_jbf(a, s);
}

static void jbf_fill_launch(const kde_jbf* h, JbfLaunch& a)
{
    a.width = h->width;
    a.height = h->height;
    a.n = 0;
    a.window = h->p.window_size;
    a.depth = nullptr;
    a.guide = nullptr;
    a.out = nullptr;
    a.s_eff = h->s_eff.p;
    a.table_host = h->table.data();
    a.log2_pk_dev = h->log2_pk.p;
    a.spatial_sigma = h->p.spatial_sigma;
    a.color_sigma = h->p.color_sigma;
    a.depth_sigma = h->p.depth_sigma;
    a.color_den = h->color_den;
    a.depth_den = h->depth_den;
    a.cd_skip = h->cd_skip;
    a.d2_skip = h->d2_skip;
    a.variant = h->variant;
}

static int jbf_presmooth(kde_jbf* h, int n, const uint8_t* bgr, uint8_t* dst, hipStream_t s)
{
    PresmoothLaunch a;
    a.width = h->width;
    a.height = h->height;
    a.n = n;
    a.radius = h->pre_radius;
    a.src = bgr;
    a.dst = dst;
    a.lut = h->pre_lut.p;
    a.grid_cap = h->pre_grid_cap;
    return launch_presmooth(a, s);
}

extern "C" int kde_jbf_process_batch(kde_jbf* h, int n, const float* depth_dev, const uint8_t* bgr_dev,
                                     float* filtered_dev, void* stream)
{
    KDE_REQUIRE(h && depth_dev && bgr_dev, "kde_jbf_process_batch: null argument");
    KDE_ON_DEVICE(h, "kde_jbf_process_batch");
    KDE_REQUIRE(n >= 1 && n <= h->max_batch, "kde_jbf_process_batch: n=%d outside 1..max_batch=%d", n, h->max_batch);
    hipStream_t s = as_stream(stream);
    float* out = filtered_dev ? filtered_dev : h->filtered.p;
    const uint8_t* guide = bgr_dev;
    if (h->p.presmooth) {
        KDE_TRY(jbf_presmooth(h, n, bgr_dev, h->smooth.p, s));
        guide = h->smooth.p;
    }
    KDE_TRY(jbf_filter(h, n, depth_dev, guide, out, s));
    if (!filtered_dev) h->n_last = n;       // results in a caller's buffer are the caller's: the host getter never reads them
    return KDE_OK;
}

extern "C" int kde_jbf_process(kde_jbf* h, const float* depth_dev, const uint8_t* bgr_dev, size_t bgr_step, void* stream)
{
    KDE_REQUIRE(h, "kde_jbf_process: null handle");
    KDE_REQUIRE(bgr_step == (size_t)h->width * 3, "kde_jbf_process: colour image must be continuous (step %zu != 3*width)", bgr_step);
    return kde_jbf_process_batch(h, 1, depth_dev, bgr_dev, nullptr, stream);
}

extern "C" int kde_jbf_presmooth_batch(kde_jbf* h, int n, const uint8_t* bgr_dev, uint8_t* smooth_dev, void* stream)
{
    KDE_REQUIRE(h && bgr_dev, "kde_jbf_presmooth_batch: null argument");
    KDE_ON_DEVICE(h, "kde_jbf_presmooth_batch");
    KDE_REQUIRE(h->p.presmooth, "kde_jbf_presmooth_batch: handle was created with presmooth = 0");
    KDE_REQUIRE(n >= 1 && (smooth_dev || n <= h->max_batch), "kde_jbf_presmooth_batch: bad n");
    return jbf_presmooth(h, n, bgr_dev, smooth_dev ? smooth_dev : h->smooth.p, as_stream(stream));
}

extern "C" int kde_jbf_filter_batch(kde_jbf* h, int n, const float* depth_dev, const uint8_t* guide_bgr_dev,
                                    float* filtered_dev, void* stream)
{
    KDE_REQUIRE(h && depth_dev && guide_bgr_dev, "kde_jbf_filter_batch: null argument");
    KDE_ON_DEVICE(h, "kde_jbf_filter_batch");
    KDE_REQUIRE(n >= 1 && n <= 65535 && (filtered_dev || n <= h->max_batch), "kde_jbf_filter_batch: bad n");
    float* out = filtered_dev ? filtered_dev : h->filtered.p;
    KDE_TRY(jbf_filter(h, n, depth_dev, guide_bgr_dev, out, as_stream(stream)));
    if (!filtered_dev) h->n_last = n;
    return KDE_OK;
}

extern "C" int kde_jbf_filtered_device(kde_jbf* h, float** out)
{
    KDE_REQUIRE(h && out, "kde_jbf_filtered_device: null argument");
    *out = h->filtered.p;
    return KDE_OK;
}

extern "C" int kde_jbf_filtered_host(kde_jbf* h, void* stream, const float** out)
{
    KDE_REQUIRE(h && out, "kde_jbf_filtered_host: null argument");
    KDE_ON_DEVICE(h, "kde_jbf_filtered_host");
    // Filtered_Host mirrors the object's own Filtered_Device (JointBilateralFilter.cpp:45-49): n_last <= max_batch
    // frames of it, never a caller-owned output buffer (which may be larger than the pinned buffer, or freed)
    const int frames = h->n_last > 0 ? (h->n_last < h->max_batch ? h->n_last : h->max_batch) : 1;
    const size_t count = (size_t)h->width * h->height * frames;
    KDE_TRY(h->filtered_host.ensure((size_t)h->width * h->height * h->max_batch));
    KDE_HIP_TRY(hipMemcpyAsync(h->filtered_host.p, h->filtered.p, count * sizeof(float), hipMemcpyDeviceToHost, as_stream(stream)));
    KDE_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    *out = h->filtered_host.p;
    return KDE_OK;
}

extern "C" int kde_jbf_smooth_device(kde_jbf* h, uint8_t** out)
{
    KDE_REQUIRE(h && out, "kde_jbf_smooth_device: null argument");
    *out = h->smooth.p;
    return KDE_OK;
}

extern "C" int kde_jbf_spatial_table(kde_jbf* h, float* table_host, int capacity)
{
    KDE_REQUIRE(h && table_host, "kde_jbf_spatial_table: null argument");
    KDE_REQUIRE(capacity >= (int)h->table.size(), "kde_jbf_spatial_table: capacity %d < %zu", capacity, h->table.size());
    memcpy(table_host, h->table.data(), h->table.size() * sizeof(float));
    return KDE_OK;
}

extern "C" int kde_jbf_set_variant(kde_jbf* h, int variant)
{
    KDE_REQUIRE(h, "kde_jbf_set_variant: null handle");
    KDE_REQUIRE(variant >= -1 && variant < jbf_variant_count(), "kde_jbf_set_variant: variant %d out of range", variant);
    h->variant = variant;
    return KDE_OK;
}

extern "C" int kde_jbf_active_variant(kde_jbf* h, int* variant)
{
    KDE_REQUIRE(h && variant, "kde_jbf_active_variant: null argument");
    JbfLaunch a;
    jbf_fill_launch(h, a);
    *variant = jbf_active_variant(a);
    return KDE_OK;
}

extern "C" int kde_jbf_variant_count(void) { return jbf_variant_count(); }
extern "C" const char* kde_jbf_variant_name(int variant) { return jbf_variant_name(variant); }

// =====================================================================================================
// MarkovRandomField
// =====================================================================================================
struct kde_mrf {
    int device = -1;
    int width, height, max_batch, window;
    float color_sigma, smooth_sigma;
    DevBuf<float> filtered;          // Filtered_Device
    PinnedBuf<float> filtered_host;  // Filtered_Host (MarkovRandomField.h:16)
    int n_last = 0;                  // frames of the last call that wrote Filtered_Device
};

extern "C" int kde_mrf_create(kde_mrf** out, int width, int height, int max_batch, int window, float color_sigma, float smooth_sigma)
{
    KDE_REQUIRE(out, "kde_mrf_create: null out");
    *out = nullptr;
    KDE_REQUIRE(width >= 1 && height >= 1 && max_batch >= 1 && max_batch <= 65535, "kde_mrf_create: bad size");
    if (window <= 0) window = 5;                  // MarkovRandomField.cpp:3
    if (color_sigma < 0.0f) color_sigma = 50.0f;  // :5
    if (smooth_sigma < 0.0f) smooth_sigma = 150.0f;  // :6
    KDE_REQUIRE(window <= 31 && (window & 1), "kde_mrf_create: window must be odd <= 31");
    kde_mrf* h = new (std::nothrow) kde_mrf;
    if (!h) return fail(KDE_ERR_NOMEM, "kde_mrf_create: out of host memory");
    h->device = current_device();
    h->width = width; h->height = height; h->max_batch = max_batch; h->window = window;
    h->color_sigma = color_sigma; h->smooth_sigma = smooth_sigma;
    int rc = h->filtered.alloc((size_t)width * height * max_batch);
    if (rc != KDE_OK) { delete h; return rc; }
    *out = h;
    return KDE_OK;
}

extern "C" int kde_mrf_destroy(kde_mrf* h) { delete h; return KDE_OK; }

extern "C" int kde_mrf_process_batch(kde_mrf* h, int n, const float* depth_dev, const uint8_t* bgr_dev, float* filtered_dev, void* stream)
{
    KDE_REQUIRE(h && depth_dev && bgr_dev, "kde_mrf_process_batch: null argument");
    KDE_ON_DEVICE(h, "kde_mrf_process_batch");
    KDE_REQUIRE(n >= 1 && n <= 65535 && (filtered_dev || n <= h->max_batch), "kde_mrf_process_batch: bad n");
    MrfLaunch a{h->width, h->height, n, h->window, depth_dev, bgr_dev, filtered_dev ? filtered_dev : h->filtered.p,
                h->color_sigma, h->smooth_sigma};
    KDE_TRY(launch_mrf(a, as_stream(stream)));
    if (!filtered_dev) h->n_last = n;
    return KDE_OK;
}

// float* MarkovRandomField::getFiltered_Host() (MarkovRandomField.h:16; the reference refreshes it after every Process,
// MarkovRandomField.cu:48): here a lazy copy of the object's own Filtered_Device, never of a caller's output buffer
extern "C" int kde_mrf_filtered_host(kde_mrf* h, void* stream, const float** out)
{
    KDE_REQUIRE(h && out, "kde_mrf_filtered_host: null argument");
    KDE_ON_DEVICE(h, "kde_mrf_filtered_host");
    const int frames = h->n_last > 0 ? (h->n_last < h->max_batch ? h->n_last : h->max_batch) : 1;
    const size_t count = (size_t)h->width * h->height * frames;
    KDE_TRY(h->filtered_host.ensure((size_t)h->width * h->height * h->max_batch));
    KDE_HIP_TRY(hipMemcpyAsync(h->filtered_host.p, h->filtered.p, count * sizeof(float), hipMemcpyDeviceToHost, as_stream(stream)));
    KDE_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    *out = h->filtered_host.p;
    return KDE_OK;
}

extern "C" int kde_mrf_filtered_device(kde_mrf* h, float** out)
{
    KDE_REQUIRE(h && out, "kde_mrf_filtered_device: null argument");
    *out = h->filtered.p;
    return KDE_OK;
}

// =====================================================================================================
// DimensionConvertor
// =====================================================================================================
struct kde_dimconv {
    Camera cam{};
    bool set = false;
};

extern "C" int kde_dimconv_create(kde_dimconv** out)
{
    KDE_REQUIRE(out, "kde_dimconv_create: null out");
    *out = new (std::nothrow) kde_dimconv;
    return *out ? KDE_OK : fail(KDE_ERR_NOMEM, "kde_dimconv_create: out of host memory");
}

extern "C" int kde_dimconv_destroy(kde_dimconv* h) { delete h; return KDE_OK; }

extern "C" int kde_dimconv_set_camera(kde_dimconv* h, const double* K, int width, int height)
{
    KDE_REQUIRE(h && K, "kde_dimconv_set_camera: null argument");
    KDE_REQUIRE(width >= 1 && height >= 1 && (long long)width * height <= (1ll << 30), "kde_dimconv_set_camera: bad size");
    // DimensionConvertor.cpp:3-13
    h->cam.fx = (float)K[0];
    h->cam.fy = (float)K[4];
    h->cam.cx = (int)K[2];
    h->cam.cy = (int)K[5];
    h->cam.width = width;
    h->cam.height = height;
    h->set = true;
    return KDE_OK;
}

static int dimconv_check(kde_dimconv* h, int n, const void* in, const void* out, const char* who)
{
    KDE_REQUIRE(h && in && out, "%s: null argument", who);
    KDE_REQUIRE(h->set, "%s: setCameraParameters was not called", who);
    KDE_REQUIRE(n >= 1 && n <= 65535, "%s: bad frame count %d", who, n);
    // any float* / float3* is accepted, as by the reference: pointers that are not 16-byte aligned (or batched frames
    // whose size is not a multiple of 4) take the scalar kernels of stream_kernels.hip
    return KDE_OK;
}

extern "C" int kde_dimconv_projective_to_real_depth(kde_dimconv* h, int n, const float* depth_dev, kde_float3* out_dev, void* stream)
{
    KDE_TRY(dimconv_check(h, n, depth_dev, out_dev, "kde_dimconv_projective_to_real_depth"));
    return launch_p2r_depth(h->cam, n, depth_dev, out_dev, as_stream(stream));
}

extern "C" int kde_dimconv_projective_to_real_points(kde_dimconv* h, int n, const kde_float3* in_dev, kde_float3* out_dev, void* stream)
{
    KDE_TRY(dimconv_check(h, n, in_dev, out_dev, "kde_dimconv_projective_to_real_points"));
    return launch_p2r_points(h->cam, n, in_dev, out_dev, as_stream(stream));
}

extern "C" int kde_dimconv_projective_to_real_interp(kde_dimconv* h, int n, const float* depth_dev, kde_float3* out_dev, void* stream)
{
    KDE_TRY(dimconv_check(h, n, depth_dev, out_dev, "kde_dimconv_projective_to_real_interp"));
    return launch_p2r_interp(h->cam, n, depth_dev, out_dev, as_stream(stream));
}

extern "C" int kde_dimconv_real_to_projective(kde_dimconv* h, int n, const kde_float3* in_dev, kde_float3* out_dev, void* stream)
{
    KDE_TRY(dimconv_check(h, n, in_dev, out_dev, "kde_dimconv_real_to_projective"));
    return launch_r2p(h->cam, n, in_dev, out_dev, as_stream(stream));
}

// =====================================================================================================
// Buffer2D
// =====================================================================================================
struct kde_buffer2d {
    int device = -1;
    int width, height;
    DevBuf<kde_weighted_d> buf;   // devPtr
};

extern "C" int kde_buffer2d_create(kde_buffer2d** out, int width, int height)
{
    KDE_REQUIRE(out, "kde_buffer2d_create: null out");
    *out = nullptr;
    KDE_REQUIRE(width >= 1 && height >= 1 && (long long)width * height <= (1ll << 30), "kde_buffer2d_create: bad size");
    kde_buffer2d* h = new (std::nothrow) kde_buffer2d;
    if (!h) return fail(KDE_ERR_NOMEM, "kde_buffer2d_create: out of host memory");
    h->device = current_device();
    h->width = width;
    h->height = height;
    int rc = h->buf.alloc((size_t)width * height);
    if (rc == KDE_OK) rc = launch_buf_init(h->buf.p, h->buf.n, nullptr);   // initDeviceMemoryElements
    if (rc == KDE_OK && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(KDE_ERR_HIP, "kde_buffer2d_create: init failed");
    if (rc != KDE_OK) { delete h; return rc; }
    *out = h;
    return KDE_OK;
}

extern "C" int kde_buffer2d_destroy(kde_buffer2d* h) { delete h; return KDE_OK; }

extern "C" int kde_buffer2d_insert_depth(kde_buffer2d* h, const float* depth_dev, void* stream)
{
    KDE_REQUIRE(h && depth_dev, "kde_buffer2d_insert_depth: null argument");
    KDE_ON_DEVICE(h, "kde_buffer2d_insert_depth");
    return launch_buf_insert_depth(h->buf.p, depth_dev, h->buf.n, as_stream(stream));
}

extern "C" int kde_buffer2d_insert_float2(kde_buffer2d* h, const float* xy_dev, void* stream)
{
    KDE_REQUIRE(h && xy_dev, "kde_buffer2d_insert_float2: null argument");
    KDE_ON_DEVICE(h, "kde_buffer2d_insert_float2");
    return launch_buf_insert_float2(h->buf.p, xy_dev, h->width, h->height, as_stream(stream));
}

extern "C" int kde_buffer2d_insert_weighted(kde_buffer2d* h, const kde_weighted_d* data_dev, void* stream)
{
    KDE_REQUIRE(h && data_dev, "kde_buffer2d_insert_weighted: null argument");
    KDE_ON_DEVICE(h, "kde_buffer2d_insert_weighted");
    KDE_HIP_TRY(hipMemcpyAsync(h->buf.p, data_dev, h->buf.n * sizeof(kde_weighted_d), hipMemcpyDeviceToDevice, as_stream(stream)));
    return KDE_OK;
}

extern "C" int kde_buffer2d_get_depth_map(kde_buffer2d* h, float* out_dev, void* stream)
{
    KDE_REQUIRE(h && out_dev, "kde_buffer2d_get_depth_map: null argument");
    KDE_ON_DEVICE(h, "kde_buffer2d_get_depth_map");
    return launch_buf_get(h->buf.p, out_dev, h->buf.n, 0, as_stream(stream));
}

extern "C" int kde_buffer2d_get_weight_map(kde_buffer2d* h, float* out_dev, void* stream)
{
    KDE_REQUIRE(h && out_dev, "kde_buffer2d_get_weight_map: null argument");
    KDE_ON_DEVICE(h, "kde_buffer2d_get_weight_map");
    return launch_buf_get(h->buf.p, out_dev, h->buf.n, 1, as_stream(stream));
}

extern "C" int kde_buffer2d_update_sequence(kde_buffer2d* h, int n_frames, const float* depth_dev, void* stream)
{
    KDE_REQUIRE(h && depth_dev, "kde_buffer2d_update: null argument");
    KDE_ON_DEVICE(h, "kde_buffer2d_update");
    KDE_REQUIRE(n_frames >= 1, "kde_buffer2d_update: n_frames must be >= 1");
    return launch_buf_update(h->buf.p, depth_dev, h->buf.n, n_frames, as_stream(stream));
}

extern "C" int kde_buffer2d_update(kde_buffer2d* h, const float* depth_dev, void* stream)
{
    return kde_buffer2d_update_sequence(h, 1, depth_dev, stream);
}

extern "C" int kde_buffer2d_raw_pointer(kde_buffer2d* h, kde_weighted_d** out)
{
    KDE_REQUIRE(h && out, "kde_buffer2d_raw_pointer: null argument");
    *out = h->buf.p;
    return KDE_OK;
}

// =====================================================================================================
// DepthAdaptiveSuperpixel
// =====================================================================================================
struct kde_dasp {
    int device = -1;
    int width, height;
    int max_batch = 1;                   // > 1 only for the private segmenters of a batched pipeline object
    bool set = false;
    DaspGeom g{};
    DevBuf<int32_t> labels;              // Labels_Device                [max_batch][H][W]
    DevBuf<kde_label_distance> ld;       // LD_Device                    [max_batch][H][W]
    DevBuf<kde_superpixel> mean;         // meanData_Device              [max_batch][rows*cols]
    DevBuf<kde_float3> centers;          // superpixelCenters_Device     [max_batch][rows*cols]
    DevBuf<float> intr;                  // intrinsicDevice
    PinnedBuf<int32_t> labels_host;      // Labels_Host
    PinnedBuf<kde_superpixel> mean_host; // meanData_Host
    // Set by the pipeline objects (RGBF / SPDSR) for their PRIVATE segmenters: the analyzeClusters that
    // follows the last calculateLD only refreshes mean/centres, which nothing reads before the next
    // Segmentation re-samples them (DepthAdaptiveSuperpixel.cu:576-586) and which the pipelines do not expose.
    bool skip_trailing_analyze = false;
};

static int dasp_create_impl(kde_dasp** out, int width, int height, int max_batch)
{
    KDE_REQUIRE(out, "kde_dasp_create: null out");
    *out = nullptr;
    KDE_REQUIRE(width >= 1 && height >= 1 && (long long)width * height <= (1ll << 30), "kde_dasp_create: bad size");
    KDE_REQUIRE(max_batch >= 1 && max_batch <= 65535, "create: max_batch must be in 1..65535");
    kde_dasp* h = new (std::nothrow) kde_dasp;
    if (!h) return fail(KDE_ERR_NOMEM, "kde_dasp_create: out of host memory");
    h->device = current_device();
    h->width = width;
    h->height = height;
    h->max_batch = max_batch;
    const size_t px = (size_t)width * height * max_batch;
    int rc = h->labels.alloc(px);                 // SuperpixelSegmentation.cpp (ctor)
    if (rc == KDE_OK) rc = h->ld.alloc(px);
    if (rc == KDE_OK) rc = h->intr.alloc(9);      // DepthAdaptiveSuperpixel.cpp:6
    if (rc != KDE_OK) { delete h; return rc; }
    *out = h;
    return KDE_OK;
}

extern "C" int kde_dasp_create(kde_dasp** out, int width, int height) { return dasp_create_impl(out, width, height, 1); }

extern "C" int kde_dasp_destroy(kde_dasp* h) { delete h; return KDE_OK; }

static int dasp_geometry(int width, int height, int rows, int cols, DaspGeom* g)
{
    KDE_REQUIRE(rows >= 1 && cols >= 1, "SetParametor: rows and cols must be >= 1");
    const int wx = width / cols, wy = height / rows;   // DepthAdaptiveSuperpixel.cpp:19-21
    KDE_REQUIRE(wx >= 4 && wy >= 4, "SetParametor: window %dx%d < 4x4 (the 4x4 candidate grid would leave the image)", wx, wy);
    KDE_REQUIRE(width / wx == cols, "SetParametor: width/(width/cols) != cols (cluster table would be indexed out of bounds)");
    KDE_REQUIRE(height >= 6, "SetParametor: height must be >= 6");
    g->width = width; g->height = height; g->rows = rows; g->cols = cols; g->wx = wx; g->wy = wy;
    return KDE_OK;
}

extern "C" int kde_dasp_set_parameters(kde_dasp* h, int rows, int cols, const double* K)
{
    KDE_REQUIRE(h && K, "kde_dasp_set_parameters: null argument");
    KDE_ON_DEVICE(h, "kde_dasp_set_parameters");
    DaspGeom g;
    KDE_TRY(dasp_geometry(h->width, h->height, rows, cols, &g));
    const size_t k = (size_t)rows * cols * h->max_batch;
    KDE_TRY(h->mean.alloc(k));       // initMemory, DepthAdaptiveSuperpixel.cpp:40-50
    KDE_TRY(h->centers.alloc(k));
    KDE_HIP_TRY(hipMemset(h->mean.p, 0, k * sizeof(kde_superpixel)));
    KDE_HIP_TRY(hipMemset(h->centers.p, 0, k * sizeof(kde_float3)));
    float intr[9];
    for (int i = 0; i < 9; i++) intr[i] = (float)K[i];   // DepthAdaptiveSuperpixel.cpp:33-37
    KDE_HIP_TRY(hipMemcpy(h->intr.p, intr, sizeof(intr), hipMemcpyHostToDevice));
    h->g = g;
    h->set = true;
    return KDE_OK;
}

extern "C" int kde_dasp_segmentation(kde_dasp* h, const uint8_t* bgr_dev, const kde_float3* points_dev,
                                     float color_sigma, float spatial_sigma, float depth_sigma, int iteration, void* stream)
{
    KDE_REQUIRE(h && bgr_dev && points_dev, "kde_dasp_segmentation: null argument");
    KDE_ON_DEVICE(h, "kde_dasp_segmentation");
    KDE_REQUIRE(h->set, "kde_dasp_segmentation: SetParametor was not called");
    KDE_REQUIRE(iteration >= 0, "kde_dasp_segmentation: negative iteration count");
    // the weights are (sigma / sum of sigmas)^2 (.cu:209-217): a zero sum is 0/0 in the reference
    KDE_REQUIRE(spatial_sigma + color_sigma + depth_sigma != 0.0f, "kde_dasp_segmentation: the sigmas must not sum to zero");
    hipStream_t s = as_stream(stream);
    // DepthAdaptiveSuperpixel.cu:570-586
    // init_LD (K5) is folded into the first calculateLD: its output is only ever read there
    KDE_TRY(launch_dasp_sample(h->g, 1, bgr_dev, points_dev, h->mean.p, h->centers.p, nullptr, nullptr, s));
    for (int i = 0; i < iteration; i++) {
        KDE_TRY(launch_dasp_calc_ld(h->g, bgr_dev, points_dev, h->ld.p, h->mean.p, h->centers.p, h->labels.p,
                                    color_sigma, spatial_sigma, depth_sigma, i == 0, s));
        if (h->skip_trailing_analyze && i == iteration - 1) break;
        KDE_TRY(launch_dasp_analyze(h->g, bgr_dev, points_dev, h->labels.p, h->mean.p, h->centers.p, h->intr.p, s));
    }
    return KDE_OK;
}

extern "C" int kde_dasp_labels_device(kde_dasp* h, int32_t** out)
{
    KDE_REQUIRE(h && out, "kde_dasp_labels_device: null argument");
    *out = h->labels.p;
    return KDE_OK;
}

extern "C" int kde_dasp_mean_device(kde_dasp* h, kde_superpixel** out)
{
    KDE_REQUIRE(h && out, "kde_dasp_mean_device: null argument");
    *out = h->mean.p;
    return KDE_OK;
}

extern "C" int kde_dasp_centers_device(kde_dasp* h, kde_float3** out)
{
    KDE_REQUIRE(h && out, "kde_dasp_centers_device: null argument");
    *out = h->centers.p;
    return KDE_OK;
}

extern "C" int kde_dasp_ld_device(kde_dasp* h, kde_label_distance** out)
{
    KDE_REQUIRE(h && out, "kde_dasp_ld_device: null argument");
    *out = h->ld.p;
    return KDE_OK;
}

extern "C" int kde_dasp_labels_host(kde_dasp* h, void* stream, const int32_t** out)
{
    KDE_REQUIRE(h && out, "kde_dasp_labels_host: null argument");
    KDE_ON_DEVICE(h, "kde_dasp_labels_host");
    const size_t px = (size_t)h->width * h->height;
    KDE_TRY(h->labels_host.ensure(px));
    KDE_HIP_TRY(hipMemcpyAsync(h->labels_host.p, h->labels.p, px * sizeof(int32_t), hipMemcpyDeviceToHost, as_stream(stream)));
    KDE_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    *out = h->labels_host.p;
    return KDE_OK;
}

extern "C" int kde_dasp_mean_host(kde_dasp* h, void* stream, const kde_superpixel** out, int* count)
{
    KDE_REQUIRE(h && out && count, "kde_dasp_mean_host: null argument");
    KDE_ON_DEVICE(h, "kde_dasp_mean_host");
    KDE_REQUIRE(h->set, "kde_dasp_mean_host: SetParametor has not been called");
    const size_t nc = (size_t)h->g.rows * h->g.cols;
    KDE_TRY(h->mean_host.ensure(nc));
    KDE_HIP_TRY(hipMemcpyAsync(h->mean_host.p, h->mean.p, nc * sizeof(kde_superpixel), hipMemcpyDeviceToHost, as_stream(stream)));
    KDE_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    *out = h->mean_host.p;
    *count = (int)nc;
    return KDE_OK;
}

// =====================================================================================================
// EdgeRefinedSuperpixel
// =====================================================================================================
struct kde_ers {
    int device = -1;
    int width, height;
    int max_batch = 1;                    // > 1 only inside a batched pipeline object
    int n_last = 1;                       // frames of the last EdgeRefining
    static constexpr int WindowSize = 7;            // EdgeRefinedSuperpixel.cpp:4
    static constexpr float SpatialSigma = 30.0f;    // :5
    static constexpr float ColorSigma = 50.0f;      // :6
    static constexpr float DepthSigma = 70.0f;      // :7
    DevBuf<float> s_eff;                  // SpatialFilter_Device
    DevBuf<int32_t> labels_a, labels_b;   // refinedLabels_Device [max_batch] + one frame of phase scratch
    DevBuf<float> depth_a, depth_b;       // K9 result [max_batch] + one frame of phase scratch
    DevBuf<float> refined_depth;          // refinedDepth_Device [max_batch]
    PinnedBuf<int32_t> labels_host;
    PinnedBuf<float> depth_host;
    float exp_zero = 0;
    float table_host[49];                 // SpatialFilter_Host as calcSpatialFilter computed it
    int enhance_variant = 0;              // kde_ers_set_variant
};

static int ers_create_impl(kde_ers** out, int width, int height, int max_batch)
{
    KDE_REQUIRE(out, "kde_ers_create: null out");
    *out = nullptr;
    KDE_REQUIRE(width >= 1 && height >= 1 && (long long)width * height <= (1ll << 30), "kde_ers_create: bad size");
    KDE_REQUIRE(max_batch >= 1 && max_batch <= 65535, "create: max_batch must be in 1..65535");
    kde_ers* h = new (std::nothrow) kde_ers;
    if (!h) return fail(KDE_ERR_NOMEM, "kde_ers_create: out of host memory");
    h->device = current_device();
    h->width = width;
    h->height = height;
    h->max_batch = max_batch;
    h->exp_zero = exp_zero_threshold();
    const size_t px = (size_t)width * height;
    float table[49];
    spatial_table(kde_ers::WindowSize, kde_ers::SpatialSigma, table);
    memcpy(h->table_host, table, sizeof(table));
    for (float& v : table)
        if (v == 0.0f) v = 1.0f;
    int rc = h->s_eff.alloc(49);
    if (rc == KDE_OK) rc = h->labels_a.alloc(px * max_batch);
    if (rc == KDE_OK) rc = h->labels_b.alloc(px);
    if (rc == KDE_OK) rc = h->depth_a.alloc(px * max_batch);
    if (rc == KDE_OK) rc = h->depth_b.alloc(px);
    if (rc == KDE_OK) rc = h->refined_depth.alloc(px * max_batch);
    if (rc == KDE_OK && hipMemcpy(h->s_eff.p, table, sizeof(table), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(KDE_ERR_HIP, "kde_ers_create: table upload failed");
    if (rc != KDE_OK) { delete h; return rc; }
    *out = h;
    return KDE_OK;
}

extern "C" int kde_ers_create(kde_ers** out, int width, int height) { return ers_create_impl(out, width, height, 1); }

extern "C" int kde_ers_destroy(kde_ers* h) { delete h; return KDE_OK; }

// n frames back to back in every argument (n = 1: the reference's call)
static int ers_edge_refining_n(kde_ers* h, int n, const int32_t* color_labels_dev, const int32_t* depth_labels_dev,
                               const float* depth_dev, const uint8_t* bgr_dev, void* stream)
{
    KDE_REQUIRE(h && color_labels_dev && depth_labels_dev && depth_dev && bgr_dev, "kde_ers_edge_refining: null argument");
    KDE_ON_DEVICE(h, "kde_ers_edge_refining");
    KDE_REQUIRE(n >= 1 && n <= h->max_batch, "EdgeRefining: n=%d outside 1..max_batch=%d", n, h->max_batch);
    hipStream_t s = as_stream(stream);
    const int W = h->width, H = h->height;
    // EdgeRefinedSuperpixel.cu:210-211 copies labels/depth, then edge_refining works in place; here the
    // horizontal phase reads the caller's buffers and the vertical phase reads the horizontal result (kept in
    // LDS by the fused kernel), so the two D2D copies disappear.
    KDE_TRY(launch_ers_edge_refining(W, H, n, kde_ers::WindowSize, color_labels_dev, depth_labels_dev, depth_dev,
                                     h->labels_b.p, h->depth_b.p, h->labels_a.p, h->depth_a.p,
                                     /*two_launches=*/h->enhance_variant == 3, s));
    // depthmap_enhancement (.cu:220-221)
    KDE_TRY(launch_ers_enhance(W, H, n, h->depth_a.p, bgr_dev, h->labels_a.p, h->s_eff.p, h->table_host,
                               kde_ers::WindowSize, kde_ers::ColorSigma, kde_ers::DepthSigma, h->exp_zero,
                               h->refined_depth.p, h->enhance_variant, s));
    h->n_last = n;
    return KDE_OK;
}

extern "C" int kde_ers_edge_refining(kde_ers* h, const int32_t* color_labels_dev, const int32_t* depth_labels_dev,
                                     const float* depth_dev, const uint8_t* bgr_dev, void* stream)
{
    return ers_edge_refining_n(h, 1, color_labels_dev, depth_labels_dev, depth_dev, bgr_dev, stream);
}

extern "C" int kde_ers_set_variant(kde_ers* h, int variant)
{
    KDE_REQUIRE(h, "kde_ers_set_variant: null handle");
    KDE_REQUIRE(variant >= 0 && variant <= 3, "kde_ers_set_variant: variant %d out of range (0..3)", variant);
    h->enhance_variant = variant;
    return KDE_OK;
}

extern "C" int kde_ers_stage_edge_depth_device(kde_ers* h, float** out)
{
    KDE_REQUIRE(h && out, "kde_ers_stage_edge_depth_device: null argument");
    *out = h->depth_a.p;
    return KDE_OK;
}

extern "C" int kde_ers_refined_labels_device(kde_ers* h, int32_t** out)
{
    KDE_REQUIRE(h && out, "kde_ers_refined_labels_device: null argument");
    *out = h->labels_a.p;
    return KDE_OK;
}

extern "C" int kde_ers_refined_depth_device(kde_ers* h, float** out)
{
    KDE_REQUIRE(h && out, "kde_ers_refined_depth_device: null argument");
    *out = h->refined_depth.p;
    return KDE_OK;
}

// the *_Host getters mirror the frames the last call produced (one for the reference's single-frame calls)
extern "C" int kde_ers_refined_labels_host(kde_ers* h, void* stream, const int32_t** out)
{
    KDE_REQUIRE(h && out, "kde_ers_refined_labels_host: null argument");
    KDE_ON_DEVICE(h, "kde_ers_refined_labels_host");
    const size_t px = (size_t)h->width * h->height;
    KDE_TRY(h->labels_host.ensure(px * h->max_batch));
    KDE_HIP_TRY(hipMemcpyAsync(h->labels_host.p, h->labels_a.p, px * h->n_last * sizeof(int32_t), hipMemcpyDeviceToHost, as_stream(stream)));
    KDE_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    *out = h->labels_host.p;
    return KDE_OK;
}

extern "C" int kde_ers_refined_depth_host(kde_ers* h, void* stream, const float** out)
{
    KDE_REQUIRE(h && out, "kde_ers_refined_depth_host: null argument");
    KDE_ON_DEVICE(h, "kde_ers_refined_depth_host");
    const size_t px = (size_t)h->width * h->height;
    KDE_TRY(h->depth_host.ensure(px * h->max_batch));
    KDE_HIP_TRY(hipMemcpyAsync(h->depth_host.p, h->refined_depth.p, px * h->n_last * sizeof(float), hipMemcpyDeviceToHost, as_stream(stream)));
    KDE_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    *out = h->depth_host.p;
    return KDE_OK;
}

// =====================================================================================================
// RegionGrowingBilateralFilter / SPDepthSuperResolution (pipeline objects)
// =====================================================================================================
struct Pipeline {
    int width = 0, height = 0, max_batch = 1;
    kde_dasp* SP = nullptr;     // colour segmentation
    kde_dasp* DASP = nullptr;   // depth-adaptive segmentation
    kde_ers* ERS = nullptr;
    ~Pipeline()
    {
        kde_dasp_destroy(SP);
        kde_dasp_destroy(DASP);
        kde_ers_destroy(ERS);
    }
    int init(int w, int h, int batch)
    {
        width = w;
        height = h;
        max_batch = batch;
        KDE_TRY(dasp_create_impl(&DASP, w, h, batch));
        KDE_TRY(dasp_create_impl(&SP, w, h, batch));
        DASP->skip_trailing_analyze = SP->skip_trailing_analyze = true;
        KDE_TRY(ers_create_impl(&ERS, w, h, batch));
        return KDE_OK;
    }
    // n frames back to back in depth / pts / bgr.  Every kernel of the chain takes the whole batch in one launch
    // (blockIdx -> (frame, tile); per-frame cluster tables, label maps and outputs), so a batch costs the same four
    // launches as one frame and each frame's result is bit-identical to its single-frame call.
    int run(int n, const float* depth, const kde_float3* pts, const uint8_t* bgr, float c1, float s1, float d1, float c2,
            float s2, float d2, int iters, void* stream)
    {
        KDE_REQUIRE(SP->set && DASP->set, "Process: SetParametor was not called");
        KDE_REQUIRE(bgr && pts && depth, "Process: null argument");
        KDE_ON_DEVICE(SP, "Process");
        KDE_REQUIRE(n >= 1 && n <= max_batch, "Process: n=%d outside 1..max_batch=%d", n, max_batch);
        // SP->Segmentation(...) and DASP->Segmentation(...) (RegionGrowingBilateralFilter.cpp:28-29) run on the same
        // colour + cloud with the same grid, so they share the work that does not depend on the sigmas:
        // sampleInitialClusters is computed once (its result is identical for both) and every assignment step
        // labels both maps in one pass.  Per-object results are exactly those of two separate Segmentation calls.
        hipStream_t s = as_stream(stream);
        const DaspGeom& g = SP->g;
        // The first assignment step reads the sampled clusters once for both segmenters and forms init_LD's
        // assignment in registers (calc_ld_kernel<.., FIRST>); DASP's own copy of the sampled clusters is only
        // needed as the starting point of its first analyzeClusters, i.e. when there is more than one iteration.
        // (the sampling kernel writes that copy itself: no device-to-device copies between the launches)
        KDE_TRY(launch_dasp_sample(g, n, bgr, pts, SP->mean.p, SP->centers.p, iters > 1 ? DASP->mean.p : nullptr,
                                   iters > 1 ? DASP->centers.p : nullptr, s));
        const float sa[3] = {c1, s1, d1}, sb[3] = {c2, s2, d2};
        for (int i = 0; i < iters; i++) {
            // the (distance, label) records are only read by a LATER assignment step: the last step does not store them
            // (the private segmenters of a pipeline expose labels only)
            KDE_TRY(launch_dasp_calc_ld_dual(g, n, bgr, pts, SP->ld.p, SP->mean.p, SP->centers.p, SP->labels.p, sa, DASP->ld.p,
                                             DASP->mean.p, DASP->centers.p, DASP->labels.p, sb, i == 0, /*write_ld=*/i < iters - 1, s));
            if (i == iters - 1) break;   // the trailing analyzeClusters is dead for the private segmenters
            // both objects got the same intrinsics in SetParametor, so one launch updates both cluster sets
            KDE_TRY(launch_dasp_analyze_dual(g, n, bgr, pts, SP->labels.p, SP->mean.p, SP->centers.p, DASP->labels.p,
                                             DASP->mean.p, DASP->centers.p, SP->intr.p, s));
        }
        return ers_edge_refining_n(ERS, n, SP->labels.p, DASP->labels.p, depth, bgr, stream);
    }
};

struct kde_rgbf {
    Pipeline p;
};

// max_batch frames per call (kde_rgbf_process_batch); the reference's constructor is max_batch = 1
extern "C" int kde_rgbf_create_batch(kde_rgbf** out, int width, int height, int max_batch)
{
    KDE_REQUIRE(out, "kde_rgbf_create: null out");
    *out = nullptr;
    kde_rgbf* h = new (std::nothrow) kde_rgbf;
    if (!h) return fail(KDE_ERR_NOMEM, "kde_rgbf_create: out of host memory");
    int rc = h->p.init(width, height, max_batch);
    if (rc != KDE_OK) { delete h; return rc; }
    *out = h;
    return KDE_OK;
}

extern "C" int kde_rgbf_create(kde_rgbf** out, int width, int height) { return kde_rgbf_create_batch(out, width, height, 1); }

extern "C" int kde_rgbf_destroy(kde_rgbf* h) { delete h; return KDE_OK; }

extern "C" int kde_rgbf_set_parameters(kde_rgbf* h, int rows, int cols, const double* K)
{
    KDE_REQUIRE(h, "kde_rgbf_set_parameters: null handle");
    KDE_TRY(kde_dasp_set_parameters(h->p.SP, rows, cols, K));      // RegionGrowingBilateralFilter.cpp:24
    return kde_dasp_set_parameters(h->p.DASP, rows, cols, K);      // :25
}

extern "C" int kde_rgbf_process_batch(kde_rgbf* h, int n, const float* depth_dev, const kde_float3* points_dev,
                                      const uint8_t* bgr_dev, void* stream)
{
    KDE_REQUIRE(h && depth_dev && points_dev && bgr_dev, "kde_rgbf_process: null argument");
    // RegionGrowingBilateralFilter.cpp:28-31, per frame
    return h->p.run(n, depth_dev, points_dev, bgr_dev, 200.0f, 40.0f, 0.0f, 100.0f, 20.0f, 200.0f, 1, stream);
}

extern "C" int kde_rgbf_process(kde_rgbf* h, const float* depth_dev, const kde_float3* points_dev, const uint8_t* bgr_dev, void* stream)
{
    return kde_rgbf_process_batch(h, 1, depth_dev, points_dev, bgr_dev, stream);
}

extern "C" int kde_rgbf_refined_depth_device(kde_rgbf* h, float** out)
{
    KDE_REQUIRE(h, "kde_rgbf_refined_depth_device: null handle");
    return kde_ers_refined_depth_device(h->p.ERS, out);
}

extern "C" int kde_rgbf_refined_depth_host(kde_rgbf* h, void* stream, const float** out)
{
    KDE_REQUIRE(h, "kde_rgbf_refined_depth_host: null handle");
    return kde_ers_refined_depth_host(h->p.ERS, stream, out);
}

extern "C" int kde_rgbf_refined_labels_device(kde_rgbf* h, int32_t** out)
{
    KDE_REQUIRE(h, "kde_rgbf_refined_labels_device: null handle");
    return kde_ers_refined_labels_device(h->p.ERS, out);
}

extern "C" int kde_rgbf_sp_labels_device(kde_rgbf* h, int32_t** out)
{
    KDE_REQUIRE(h, "kde_rgbf_sp_labels_device: null handle");
    return kde_dasp_labels_device(h->p.SP, out);
}

extern "C" int kde_rgbf_dasp_labels_device(kde_rgbf* h, int32_t** out)
{
    KDE_REQUIRE(h, "kde_rgbf_dasp_labels_device: null handle");
    return kde_dasp_labels_device(h->p.DASP, out);
}

struct kde_spdsr {
    Pipeline p;
    kde_dimconv conv;
    int nclusters = 0;
    DevBuf<kde_float3> edge_points;   // EdgeEnhanced3DPoints_Device            [max_batch][H][W]
    DevBuf<float> cluster_nd;         // ClusterND_Device (float4 per cluster)   [max_batch][rows*cols]
    DevBuf<double> sums, cov;         // per-cluster moments (replace the host cv::Mat / cv::PCA round trip)
    int moments_dirty = 0;            // raised while sums / cov hold accumulated moments nobody has consumed (spdsr_kernels.hip)
    DevBuf<float> nxy;                // Projection_GPU::Normalized3D_Device (x, y of the unit-depth ray; the camera's)
    DevBuf<kde_float3> plane_fitted;  // Projection_GPU::PlaneFitted3D_Device    [max_batch][H][W]
    DevBuf<kde_float3> opt_a, opt_b;  // Projection_GPU::Optimized3D_Device, double-buffered (D5)
    kde_float3* optimized = nullptr;
    int n_last = 1;
    PinnedBuf<kde_float3> optimized_host;
    SpdsrResident resident;           // one frame: the 20 sweeps as one LDS-resident cooperative launch
    ~kde_spdsr() { spdsr_resident_release(&resident); }
};

extern "C" int kde_spdsr_create_batch(kde_spdsr** out, int width, int height, int max_batch)
{
    KDE_REQUIRE(out, "kde_spdsr_create: null out");
    *out = nullptr;
    kde_spdsr* h = new (std::nothrow) kde_spdsr;
    if (!h) return fail(KDE_ERR_NOMEM, "kde_spdsr_create: out of host memory");
    int rc = h->p.init(width, height, max_batch);
    const size_t px = (size_t)width * height;
    if (rc == KDE_OK) rc = h->edge_points.alloc(px * max_batch);   // SPDepthSuperResolution.cpp:19
    if (rc == KDE_OK) rc = h->nxy.alloc(px * 2);                    // Projection_GPU::initMemory (Projection_GPU.cpp:45-51)
    if (rc == KDE_OK) rc = h->plane_fitted.alloc(px * max_batch);
    if (rc == KDE_OK) rc = h->opt_a.alloc(px * max_batch);
    if (rc == KDE_OK) rc = h->opt_b.alloc(px * max_batch);
    if (rc == KDE_OK) rc = spdsr_resident_init(&h->resident);
    if (rc != KDE_OK) { delete h; return rc; }
    *out = h;
    return KDE_OK;
}

extern "C" int kde_spdsr_create(kde_spdsr** out, int width, int height) { return kde_spdsr_create_batch(out, width, height, 1); }

extern "C" int kde_spdsr_destroy(kde_spdsr* h) { delete h; return KDE_OK; }

extern "C" int kde_spdsr_set_parameters(kde_spdsr* h, int rows, int cols, const double* K)
{
    KDE_REQUIRE(h, "kde_spdsr_set_parameters: null handle");
    KDE_TRY(kde_dasp_set_parameters(h->p.SP, rows, cols, K));       // SPDepthSuperResolution.cpp:46
    KDE_TRY(kde_dasp_set_parameters(h->p.DASP, rows, cols, K));     // :47
    KDE_TRY(kde_dimconv_set_camera(&h->conv, K, h->p.width, h->p.height));   // :48
    // Projector = new Projection_GPU(Width, Height, intrinsic) (:49): same truncated intrinsics, initNormalized3D
    h->nclusters = rows * cols;
    const size_t kb = (size_t)h->nclusters * h->p.max_batch;
    KDE_TRY(h->cluster_nd.alloc(kb * 4));          // :52-53
    KDE_TRY(h->sums.alloc(kb * 4));
    KDE_TRY(h->cov.alloc(kb * 6));
    // the moment tables are zero between calls: cluster_planes_kernel clears what it has consumed (no memsets per frame)
    KDE_HIP_TRY(hipMemset(h->sums.p, 0, kb * 4 * sizeof(double)));
    KDE_HIP_TRY(hipMemset(h->cov.p, 0, kb * 6 * sizeof(double)));
    KDE_HIP_TRY(hipMemset(h->cluster_nd.p, 0, kb * 4 * sizeof(float)));
    h->moments_dirty = 0;
    KDE_TRY(launch_spdsr_init_normalized(h->conv.cam, h->nxy.p, nullptr));
    KDE_HIP_TRY(hipStreamSynchronize(nullptr));
    return KDE_OK;
}

extern "C" int kde_spdsr_process_batch(kde_spdsr* h, int n, const float* depth_dev, const kde_float3* points_dev,
                                       const uint8_t* bgr_dev, void* stream)
{
    KDE_REQUIRE(h && depth_dev && points_dev && bgr_dev, "kde_spdsr_process: null argument");
    // SPDepthSuperResolution.cpp:59-64, per frame
    KDE_REQUIRE(h->nclusters > 0, "kde_spdsr_process: SetParametor was not called");
    KDE_TRY(h->p.run(n, depth_dev, points_dev, bgr_dev, 200.0f, 10.0f, 0.0f, 0.0f, 10.0f, 200.0f, 5, stream));
    KDE_TRY(kde_dimconv_projective_to_real_depth(&h->conv, n, h->p.ERS->refined_depth.p, h->edge_points.p, stream));
    // :65-170 on the device: per-cluster plane of the labelled cloud (no D2H / host PCA / H2D)
    hipStream_t s = as_stream(stream);
    KDE_TRY(launch_spdsr_cluster_planes(h->p.width, h->p.height, n, h->nclusters, h->p.max_batch, h->p.ERS->labels_a.p, h->edge_points.p,
                                        h->sums.p, h->cov.p, h->cluster_nd.p, &h->moments_dirty, s));
    // Projector->PlaneProjection(ClusterND_Device, refined labels, EdgeEnhanced3DPoints_Device) (:190)
    h->n_last = n;
    return launch_spdsr_plane_projection(h->p.width, h->p.height, n, h->nclusters, h->cluster_nd.p, h->p.ERS->labels_a.p,
                                         h->edge_points.p, h->nxy.p, h->plane_fitted.p, h->opt_a.p, h->opt_b.p, 20,
                                         &h->optimized, &h->resident, s);
}

extern "C" int kde_spdsr_process(kde_spdsr* h, const float* depth_dev, const kde_float3* points_dev, const uint8_t* bgr_dev, void* stream)
{
    return kde_spdsr_process_batch(h, 1, depth_dev, points_dev, bgr_dev, stream);
}

extern "C" int kde_spdsr_refined_depth_device(kde_spdsr* h, float** out)
{
    KDE_REQUIRE(h, "kde_spdsr_refined_depth_device: null handle");
    return kde_ers_refined_depth_device(h->p.ERS, out);
}

extern "C" int kde_spdsr_refined_depth_host(kde_spdsr* h, void* stream, const float** out)
{
    KDE_REQUIRE(h, "kde_spdsr_refined_depth_host: null handle");
    return kde_ers_refined_depth_host(h->p.ERS, stream, out);
}

extern "C" int kde_spdsr_refined_labels_device(kde_spdsr* h, int32_t** out)
{
    KDE_REQUIRE(h, "kde_spdsr_refined_labels_device: null handle");
    return kde_ers_refined_labels_device(h->p.ERS, out);
}

extern "C" int kde_spdsr_edge_enhanced_points_device(kde_spdsr* h, kde_float3** out)
{
    KDE_REQUIRE(h && out, "kde_spdsr_edge_enhanced_points_device: null argument");
    *out = h->edge_points.p;
    return KDE_OK;
}

extern "C" int kde_spdsr_optimized_points_device(kde_spdsr* h, kde_float3** out)
{
    KDE_REQUIRE(h && out, "kde_spdsr_optimized_points_device: null argument");
    KDE_REQUIRE(h->optimized, "getOptimizedPoints: Process has not run yet");
    *out = h->optimized;
    return KDE_OK;
}

extern "C" int kde_spdsr_optimized_points_host(kde_spdsr* h, void* stream, const kde_float3** out)
{
    KDE_REQUIRE(h && out, "kde_spdsr_optimized_points_host: null argument");
    KDE_REQUIRE(h->optimized, "getOptimizedPoints: Process has not run yet");
    KDE_ON_DEVICE(h->p.SP, "kde_spdsr_optimized_points_host");
    const size_t px = (size_t)h->p.width * h->p.height;
    KDE_TRY(h->optimized_host.ensure(px * h->p.max_batch));
    KDE_HIP_TRY(hipMemcpyAsync(h->optimized_host.p, h->optimized, px * h->n_last * sizeof(kde_float3), hipMemcpyDeviceToHost, as_stream(stream)));
    KDE_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    *out = h->optimized_host.p;
    return KDE_OK;
}

extern "C" int kde_spdsr_plane_fitted_points_device(kde_spdsr* h, kde_float3** out)
{
    KDE_REQUIRE(h && out, "kde_spdsr_plane_fitted_points_device: null argument");
    *out = h->plane_fitted.p;
    return KDE_OK;
}

extern "C" int kde_spdsr_cluster_nd_device(kde_spdsr* h, float** out)
{
    KDE_REQUIRE(h && out, "kde_spdsr_cluster_nd_device: null argument");
    *out = h->cluster_nd.p;
    return KDE_OK;
}
