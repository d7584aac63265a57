// shim_selftest.cpp — the boundary guards and extensions of include/kde/kde.hpp exercised from C++ (run by
// tests/test_gpu_cpp_shims.py): a padded / wrongly sized colour image is rejected by every class that takes one;
// MarkovRandomField::getFiltered_Host() (MarkovRandomField.h:16) mirrors Filtered_Device; ProcessBatch of the pipeline
// classes returns, per frame, the bits of the single-frame Process (RegionGrowingBilateralFilter.cpp:27-38).
// Prints "ok <what>" lines; exit code 0 = all passed.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/kde/kde.hpp"

#define HIP_OK(x)                                                        \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            std::exit(2);                                                \
        }                                                                \
    } while (0)

static int failures = 0;
static void expect(bool ok, const char* what)
{
    std::printf("%s %s\n", ok ? "ok" : "FAIL", what);
    if (!ok) failures++;
}

template <class F>
static bool throws_invalid(F f)
{
    try {
        f();
    } catch (const kde::Error& e) {
        return e.code() == KDE_ERR_INVALID;
    }
    return false;
}

int main()
{
    const int W = 160, H = 120, N = 3;
    const size_t px = (size_t)W * H;
    // synthetic frames: LCG colour, sloped depth with a step and a few holes
    std::vector<uint8_t> bgr(px * 3 * N);
    std::vector<float> depth(px * N);
    uint32_t st = 12345u;
    for (int f = 0; f < N; f++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const size_t i = f * px + (size_t)y * W + x;
                st = st * 1664525u + 1013904223u;
                const int block = ((x / 20) + (y / 15) * 3 + f) % 7;
                for (int c = 0; c < 3; c++) bgr[i * 3 + c] = (uint8_t)(30 * block + 20 * c + ((st >> (8 * c)) & 7));
                depth[i] = 1000.0f + 300.0f * block + 0.5f * x + 0.25f * y + (float)((st >> 24) & 3);
                if ((st >> 10) % 97 == 0) depth[i] = 0.0f;
            }
    uint8_t* d_bgr;
    float* d_depth;
    float3* d_pts;
    HIP_OK(hipMalloc(&d_bgr, bgr.size()));
    HIP_OK(hipMalloc(&d_depth, depth.size() * 4));
    HIP_OK(hipMalloc(&d_pts, px * N * sizeof(float3)));
    HIP_OK(hipMemcpy(d_bgr, bgr.data(), bgr.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_depth, depth.data(), depth.size() * 4, hipMemcpyHostToDevice));
    kde::Mat33d K{{575.8 * W / 640.0, 0, W / 2.0, 0, 575.8 * W / 640.0, H / 2.0, 0, 0, 1}};
    DimensionConvertor conv;
    conv.setCameraParameters(K, W, H);
    for (int f = 0; f < N; f++) conv.projectiveToReal(d_depth + f * px, d_pts + f * px);

    kde::GpuImage8UC3 good{d_bgr, H, W, (size_t)W * 3};
    kde::GpuImage8UC3 padded{d_bgr, H, W, (size_t)W * 3 + 64};
    kde::GpuImage8UC3 wrong{d_bgr, H, W - 1, (size_t)(W - 1) * 3};

    // ---- guards: every class that takes a colour image ----
    JointBilateralFilter jbf(W, H);
    MarkovRandomField mrf(W, H);
    DepthAdaptiveSuperpixel dasp(W, H);
    dasp.SetParametor(6, 8, K);
    EdgeRefinedSuperpixel ers(W, H);
    RegionGrowingBilateralFilter rg(W, H);
    rg.SetParametor(6, 8, K);
    SPDepthSuperResolution sr(W, H);
    sr.SetParametor(6, 8, K);
    expect(throws_invalid([&] { jbf.Process(d_depth, padded); }) && throws_invalid([&] { jbf.Process(d_depth, wrong); }), "JBF rejects padded / wrong-size images");
    expect(throws_invalid([&] { mrf.Process(d_depth, padded); }), "MRF rejects a padded image");
    {   // which kernel an object runs (kde_jbf_active_variant): the reference constants select a tuned window-5 kernel, a
        // window of 31 a tuned one too, a zero colour sigma (term off) the generic kernel
        kde_jbf_params wide, nocolour;
        kde_jbf_default_params(&wide);
        wide.window_size = 31;
        kde_jbf_default_params(&nocolour);
        nocolour.color_sigma = 0.0f;
        JointBilateralFilter jbf31(W, H, wide), jbf0(W, H, nocolour);
        expect(std::strncmp(jbf.activeKernel(), "w5-pk", 5) == 0 && std::strncmp(jbf31.activeKernel(), "w31-pk", 6) == 0 &&
                   std::strcmp(jbf0.activeKernel(), "generic-32x8-1px") == 0,
               "activeKernel() names the tuned / generic kernel");
    }
    expect(throws_invalid([&] { dasp.Segmentation(padded, d_pts, 200.f, 40.f, 0.f, 1); }) &&
               throws_invalid([&] { dasp.Segmentation(wrong, d_pts, 200.f, 40.f, 0.f, 1); }), "DASP rejects padded / wrong-size images");
    dasp.Segmentation(good, d_pts, 200.f, 40.f, 0.f, 1);
    expect(throws_invalid([&] { ers.EdgeRefining(dasp.getLabelDevice(), dasp.getLabelDevice(), d_depth, padded); }), "ERS rejects a padded image");
    expect(throws_invalid([&] { rg.Process(d_depth, d_pts, padded); }) && throws_invalid([&] { rg.Process(d_depth, d_pts, wrong); }), "RGBF rejects padded / wrong-size images");
    expect(throws_invalid([&] { sr.Process(d_depth, d_pts, padded); }), "SPDSR rejects a padded image");

    // ---- MarkovRandomField::getFiltered_Host ----
    mrf.Process(d_depth, good);
    std::vector<float> dev_copy(px);
    const float* host = mrf.getFiltered_Host();      // synchronises
    HIP_OK(hipMemcpy(dev_copy.data(), mrf.getFiltered_Device(), px * 4, hipMemcpyDeviceToHost));
    expect(std::memcmp(host, dev_copy.data(), px * 4) == 0, "MRF getFiltered_Host mirrors Filtered_Device");

    // ---- viewer members (host side; the reference's windows are out of scope, the pictures are not) ----
    {
        uint8_t c[3];
        struct { float r; int b, g, rr; } known[] = {{0.165f, 127, 0, 0}, {0.5f, 123, 131, 0}, {0.8f, 0, 146, 108}, {2.0f, 0, 0, 254},
                                                     {0.0f, 0, 0, 0}, {0.33f, 255, 0, 0}, {0.66f, 0, 254, 0}};
        bool ramp_ok = true;
        for (auto& k : known) {
            kde::viewers::depth_ramp(k.r, c);
            ramp_ok = ramp_ok && c[0] == k.b && c[1] == k.g && c[2] == k.rr;
        }
        expect(ramp_ok, "depth ramp reproduces the reference's getRGB at known ratios");
        std::vector<float> in(px);
        HIP_OK(hipMemcpy(in.data(), d_depth, px * 4, hipMemcpyDeviceToHost));
        mrf.visualize(in.data());
        jbf.Process(d_depth, good);
        jbf.visualize(in.data());                    // refreshes Filtered_Host, renders both pictures
        std::vector<float> filt(px);
        HIP_OK(hipMemcpy(filt.data(), jbf.getFiltered_Device(), px * 4, hipMemcpyDeviceToHost));
        bool pic_ok = jbf.getOutputDepthImage().rows == H && jbf.getInputDepthImage().cols == W;
        for (size_t q = 0; q < px && pic_ok; q++) {
            uint8_t e[3] = {0, 0, 0};
            if (filt[q] > 50.0f) kde::viewers::depth_ramp(filt[q] / 5000.0f, e);
            const uint8_t* got = jbf.getOutputDepthImage().data() + q * 3;
            pic_ok = got[0] == e[0] && got[1] == e[1] && got[2] == e[2];
            const uint8_t* gin = jbf.getInputDepthImage().data() + q * 3;
            if (!(in[q] > 50.0f)) pic_ok = pic_ok && gin[0] == 0 && gin[1] == 0 && gin[2] == 0;
        }
        expect(pic_ok, "JointBilateralFilter::visualize renders input and filtered depth (invalid pixels black)");
        // EdgeRefinedSuperpixel viewers on a real refinement
        ers.EdgeRefining(dasp.getLabelDevice(), dasp.getLabelDevice(), d_depth, good);
        const int* lab = ers.getRefinedLabels_Host();
        const float* rd = ers.getRefinedDepth_Host();
        kde::HostImage8UC3& seg = ers.getSegmentedImage(3000);
        std::vector<uint8_t> host_bgr(px * 3);
        HIP_OK(hipMemcpy(host_bgr.data(), d_bgr, px * 3, hipMemcpyDeviceToHost));
        struct HostMat { uint8_t* data; int rows, cols; size_t step; } hm{host_bgr.data(), H, W, (size_t)W * 3};
        kde::HostImage8UC3& lines = ers.getSegmentedImage(hm);
        kde::HostImage8UC3& rnd = ers.getRandomColorImage();
        bool seg_ok = true, border_seen = false;
        for (int y = 0; y + 1 < H && seg_ok; y++)
            for (int x = 0; x + 1 < W && seg_ok; x++) {
                const size_t q = (size_t)y * W + x;
                const bool border = lab[q] != lab[q + W] || lab[q] != lab[q + 1];
                border_seen = border_seen || border;
                const uint8_t* a = seg.at(y, x);
                const uint8_t* b = lines.at(y, x);
                if (border) seg_ok = a[0] == 255 && a[1] == 255 && a[2] == 255 && b[0] == 255 && b[1] == 255 && b[2] == 255;
                else {
                    seg_ok = std::memcmp(b, host_bgr.data() + q * 3, 3) == 0;
                    if (rd[q] == 0.0f && lab[q] != -100) seg_ok = seg_ok && a[0] == 0 && a[1] == 0 && a[2] == 0;
                }
                // one colour per label
                if (lab[q] == lab[q + 1]) seg_ok = seg_ok && std::memcmp(rnd.at(y, x), rnd.at(y, x + 1), 3) == 0;
            }
        expect(seg_ok && border_seen, "EdgeRefinedSuperpixel viewers: borders white, holes black, one colour per label");
        kde::HostImage8UC3& avg = dasp.getSegmentedImage(hm, DepthAdaptiveSuperpixel::Average);
        kde::HostImage8UC3& dl = dasp.getSegmentedImage(hm, DepthAdaptiveSuperpixel::Line);
        dasp.getRandomColorImage();
        dasp.releaseVideo();
        expect(avg.rows == H && dl.cols == W, "DepthAdaptiveSuperpixel viewers render");
        // as<M>(): a view for any cv::Mat_<cv::Vec3b>-like type (rows, cols, pixel pointer, step)
        struct Vec3bLike { uint8_t val[3]; };
        struct MatLike {
            typedef Vec3bLike value_type;
            int rows, cols;
            value_type* data;
            size_t step;
            MatLike(int r, int c, value_type* d, size_t s) : rows(r), cols(c), data(d), step(s) {}
        };
        MatLike view = rnd.as<MatLike>();
        expect(view.rows == H && view.cols == W && view.step == (size_t)W * 3 && (uint8_t*)view.data == rnd.data(), "HostImage8UC3::as<MatLike>() is a view");
    }

    // ---- ProcessBatch == per-frame Process, to the bit ----
    RegionGrowingBilateralFilter rgb(W, H, N);
    rgb.SetParametor(6, 8, K);
    rgb.ProcessBatch(N, d_depth, d_pts, d_bgr);
    std::vector<float> batch(px * N), single(px);
    std::vector<int> lbatch(px * N), lsingle(px);
    HIP_OK(hipMemcpy(batch.data(), rgb.getRefinedDepth_Device(), px * N * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(lbatch.data(), rgb.getRefinedLabels_Device(), px * N * 4, hipMemcpyDeviceToHost));
    bool same = true;
    for (int f = 0; f < N; f++) {
        kde::GpuImage8UC3 img{d_bgr + f * px * 3, H, W, (size_t)W * 3};
        rg.Process(d_depth + f * px, d_pts + f * px, img);
        HIP_OK(hipMemcpy(single.data(), rg.getRefinedDepth_Device(), px * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(lsingle.data(), rg.getRefinedLabels_Device(), px * 4, hipMemcpyDeviceToHost));
        same = same && std::memcmp(single.data(), batch.data() + f * px, px * 4) == 0 &&
               std::memcmp(lsingle.data(), lbatch.data() + f * px, px * 4) == 0;
    }
    expect(same, "RGBF ProcessBatch == Process per frame (labels and depth, bitwise)");
    expect(throws_invalid([&] { rgb.ProcessBatch(N + 1, d_depth, d_pts, d_bgr); }), "RGBF ProcessBatch rejects n > max_batch");

    SPDepthSuperResolution srb(W, H, N);
    srb.SetParametor(6, 8, K);
    srb.ProcessBatch(N, d_depth, d_pts, d_bgr);
    HIP_OK(hipMemcpy(batch.data(), srb.getRefinedDepth_Device(), px * N * 4, hipMemcpyDeviceToHost));
    same = true;
    for (int f = 0; f < N; f++) {
        kde::GpuImage8UC3 img{d_bgr + f * px * 3, H, W, (size_t)W * 3};
        sr.Process(d_depth + f * px, d_pts + f * px, img);
        HIP_OK(hipMemcpy(single.data(), sr.getRefinedDepth_Device(), px * 4, hipMemcpyDeviceToHost));
        same = same && std::memcmp(single.data(), batch.data() + f * px, px * 4) == 0;
    }
    expect(same, "SPDSR ProcessBatch == Process per frame (refined depth, bitwise)");

    HIP_OK(hipFree(d_bgr));
    HIP_OK(hipFree(d_depth));
    HIP_OK(hipFree(d_pts));
    std::printf("%s: %d failure(s)\n", failures ? "FAILED" : "all passed", failures);
    return failures ? 1 : 0;
}
