// kde/kde.hpp — header-only C++ classes with the reference's names and member signatures for the hot
// path, forwarding to the C ABI of libkde_hip.so (include/kde_hip.h).
//
//   reference class (file)                                        -> class below
//   JointBilateralFilter        (JointBilateralFilter/JointBilateralFilter.h:9-36)
//   MarkovRandomField           (MarkovRandomField/MarkovRandomField.h)
//   DimensionConvertor          (DimensionConvertor/DimensionConvertor.h:152-171)
//   Buffer2D / ArrayBuffer      (ArrayBuffer/Buffer2D.h:9-37, ArrayBuffer.h:9-45)
//   DepthAdaptiveSuperpixel     (SuperpixelSegmentation/DepthAdaptiveSuperpixel.h:15-28)
//   EdgeRefinedSuperpixel       (EdgeRefinedSuperpixel/EdgeRefinedSuperpixel.h:14-45)
//   RegionGrowingBilateralFilter(RegionGrowingBilateralFilter.h:11-27)
//   SPDepthSuperResolution      (SPDepthSuperResolution.h:17-46)
//
// What differs from the reference headers, and why:
//   * cv::gpu::GpuMat parameters are templates over "anything with .data/.rows/.cols/.step" — a real
//     cv::gpu::GpuMat binds unchanged, and builds without OpenCV can pass kde::GpuImage8UC3;
//   * cv::Mat_<double> intrinsic parameters are templates over "anything callable as K(row, col)"
//     (cv::Mat_<double> is) with an overload for a row-major double[9];
//   * failures throw kde::Error (std::runtime_error) instead of being ignored (the reference checks no
//     CUDA return code); nothing aborts;
//   * every object owns one stream-ordered context; an optional stream (void* = hipStream_t) can be set
//     with setStream(), default is the null stream like the reference;
//   * the viewer members (visualize, getSegmentedImage, getRandomColorImage) render into object-owned host images
//     (kde::HostImage8UC3, viewers.hpp) instead of cv::Mat_ / cv::imshow windows; releaseVideo() is a no-op;
//   * classes live in the global namespace like the reference's unless KDE_NO_GLOBAL_NAMES is defined
//     (then they are only reachable as kde::ref::Name).
#ifndef KDE_KDE_HPP
#define KDE_KDE_HPP

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "../kde_hip.h"
#include "viewers.hpp"   // host-side renderers behind visualize() / getSegmentedImage() / getRandomColorImage()

#if defined(__has_include)
#if __has_include(<hip/hip_vector_types.h>)
#include <hip/hip_vector_types.h>   // float2 / float3 as the reference's CUDA headers provide them
#define KDE_HAVE_VECTOR_TYPES 1
#endif
#endif
#ifndef KDE_HAVE_VECTOR_TYPES
struct float2 { float x, y; };
struct float3 { float x, y, z; };
#endif

namespace kde {

class Error : public std::runtime_error {
public:
    Error(int code, const std::string& what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }
private:
    int code_;
};

inline void check(int rc)
{
    if (rc != KDE_OK) throw Error(rc, std::string("libkde_hip: ") + kde_last_error_string());
}

// stand-in for a continuous CV_8UC3 cv::gpu::GpuMat (packed BGR on the device)
struct GpuImage8UC3 {
    uint8_t* data;
    int rows, cols;
    size_t step;
};

struct Mat33d {   // stand-in for cv::Mat_<double>(3,3)
    double v[9];
    double operator()(int r, int c) const { return v[r * 3 + c]; }
};

template <class MatLike>
inline void intrinsic_to_array(const MatLike& K, double out[9])
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) out[r * 3 + c] = static_cast<double>(K(r, c));
}
inline void intrinsic_to_array(const double* K, double out[9])
{
    for (int i = 0; i < 9; i++) out[i] = K[i];
}

static_assert(sizeof(float3) == sizeof(kde_float3), "float3 must be 12 bytes packed");

// every class takes a colour image as "continuous CV_8UC3 of the object's size" (cv::gpu::createContinuous,
// JointBilateralFilter.cpp:13, main.cpp:62): a padded or wrongly sized GpuMat is rejected, not silently misread
template <class GpuMatLike>
inline void require_continuous_8uc3(const GpuMatLike& img, int width, int height, const char* who)
{
    if (img.rows != height || img.cols != width || static_cast<size_t>(img.step) != static_cast<size_t>(width) * 3)
        throw Error(KDE_ERR_INVALID, std::string(who) + ": colour image must be a continuous " + std::to_string(width) + "x" +
                                         std::to_string(height) + " 8UC3 (step == 3 * width)");
}

namespace ref {

// ------------------------------------------------------------------------------------------------
class JointBilateralFilter {
public:
    JointBilateralFilter(int width, int height) : Width(width), Height(height)
    {
        check(kde_jbf_create(&h_, width, height, 1, nullptr));   // reference constants (JointBilateralFilter.cpp:3-6)
    }
    // extension: explicit parameters / batch capacity (the reference hard-codes them)
    JointBilateralFilter(int width, int height, const kde_jbf_params& params, int max_batch = 1)
        : Width(width), Height(height)
    {
        check(kde_jbf_create(&h_, width, height, max_batch, &params));
    }
    ~JointBilateralFilter() { kde_jbf_destroy(h_); }
    JointBilateralFilter(const JointBilateralFilter&) = delete;
    JointBilateralFilter& operator=(const JointBilateralFilter&) = delete;

    template <class GpuMatLike>
    void Process(float* depth_device, const GpuMatLike& color_image)
    {
        if (color_image.rows != Height || color_image.cols != Width)
            throw Error(KDE_ERR_INVALID, "JointBilateralFilter::Process: colour image size mismatch");
        check(kde_jbf_process(h_, depth_device, color_image.data, color_image.step, stream_));
    }
    void ProcessBatch(int n, const float* depth_device, const uint8_t* bgr_device, float* filtered_device = nullptr)
    {
        check(kde_jbf_process_batch(h_, n, depth_device, bgr_device, filtered_device, stream_));
    }
    float* getFiltered_Device() const
    {
        float* p = nullptr;
        check(kde_jbf_filtered_device(h_, &p));
        return p;
    }
    float* getFiltered_Host() const
    {
        const float* p = nullptr;
        check(kde_jbf_filtered_host(h_, stream_, &p));
        return const_cast<float*>(p);
    }
    GpuImage8UC3 getSmoothImage_Device()
    {
        uint8_t* p = nullptr;
        check(kde_jbf_smooth_device(h_, &p));
        return GpuImage8UC3{p, Height, Width, static_cast<size_t>(Width) * 3};
    }
    // void visualize(float* depth_host) (JointBilateralFilter.h:18, .cpp:50-79): refreshes Filtered_Host (in the reference this
    // member is the ONLY thing that does) and renders the caller's input depth and the filtered depth with the reference's
    // ramp over 5000 mm, pixels <= 50 mm black.  The reference then shows both with cv::imshow + cv::waitKey(1); windows are
    // outside the scope, the pictures are kept in the object instead (getInputDepthImage / getOutputDepthImage).
    void visualize(float* depth_host)
    {
        if (InputDepth.rows != Height) InputDepth = HostImage8UC3(Height, Width), OutputDepth = HostImage8UC3(Height, Width);
        viewers::render_depth(depth_host, 5000.0f, true, 50.0f, InputDepth);
        viewers::render_depth(getFiltered_Host(), 5000.0f, true, 50.0f, OutputDepth);
    }
    HostImage8UC3& getInputDepthImage() { return InputDepth; }       // extension: what "depth_input" would have shown
    HostImage8UC3& getOutputDepthImage() { return OutputDepth; }     // extension: what "depth_filtered" would have shown
    void setStream(void* hip_stream) { stream_ = hip_stream; }
    kde_jbf* handle() const { return h_; }
    // extension: the name of the kernel this object's parameters select ("generic-32x8-1px" when no tuned one applies:
    // window 1, a zero sigma, or a colour sigma outside the tuned range); kde_jbf_active_variant
    const char* activeKernel() const
    {
        int v = 0;
        check(kde_jbf_active_variant(h_, &v));
        return kde_jbf_variant_name(v);
    }

private:
    int Width, Height;
    kde_jbf* h_ = nullptr;
    void* stream_ = nullptr;
    HostImage8UC3 InputDepth, OutputDepth;
};

// ------------------------------------------------------------------------------------------------
class MarkovRandomField {
public:
    MarkovRandomField(int width, int height) : Width(width), Height(height)
    {
        check(kde_mrf_create(&h_, width, height, 1, 0, -1.0f, -1.0f));   // MarkovRandomField.cpp:3-6
    }
    ~MarkovRandomField() { kde_mrf_destroy(h_); }
    MarkovRandomField(const MarkovRandomField&) = delete;
    MarkovRandomField& operator=(const MarkovRandomField&) = delete;
    template <class GpuMatLike>
    void Process(float* depth_device, const GpuMatLike& color_image)
    {
        if (color_image.rows != Height || color_image.cols != Width || color_image.step != static_cast<size_t>(Width) * 3)
            throw Error(KDE_ERR_INVALID, "MarkovRandomField::Process: colour image must be continuous WxH 8UC3");
        check(kde_mrf_process_batch(h_, 1, depth_device, color_image.data, nullptr, stream_));
    }
    float* getFiltered_Device() const
    {
        float* p = nullptr;
        check(kde_mrf_filtered_device(h_, &p));
        return p;
    }
    float* getFiltered_Host() const      // MarkovRandomField.h:16
    {
        const float* p = nullptr;
        check(kde_mrf_filtered_host(h_, stream_, &p));
        return const_cast<float*>(p);
    }
    // cv::gpu::GpuMat getSmoothImage_Device() (MarkovRandomField.h:17): the reference allocates smooth_Device and never
    // writes it (MarkovRandomField.cpp:13, no kernel takes it) -- there is nothing to return but an empty image
    GpuImage8UC3 getSmoothImage_Device() { return GpuImage8UC3{nullptr, 0, 0, 0}; }
    // void visualize(float* depth_host) (MarkovRandomField.h:18, .cpp:50-79): as JointBilateralFilter::visualize
    void visualize(float* depth_host)
    {
        if (InputDepth.rows != Height) InputDepth = HostImage8UC3(Height, Width), OutputDepth = HostImage8UC3(Height, Width);
        viewers::render_depth(depth_host, 5000.0f, true, 50.0f, InputDepth);
        viewers::render_depth(getFiltered_Host(), 5000.0f, true, 50.0f, OutputDepth);
    }
    HostImage8UC3& getInputDepthImage() { return InputDepth; }
    HostImage8UC3& getOutputDepthImage() { return OutputDepth; }
    void setStream(void* hip_stream) { stream_ = hip_stream; }

private:
    int Width, Height;
    kde_mrf* h_ = nullptr;
    void* stream_ = nullptr;
    HostImage8UC3 InputDepth, OutputDepth;
};

// ------------------------------------------------------------------------------------------------
class DimensionConvertor {
public:
    DimensionConvertor() { check(kde_dimconv_create(&h_)); }
    ~DimensionConvertor() { kde_dimconv_destroy(h_); }
    DimensionConvertor(const DimensionConvertor&) = delete;
    DimensionConvertor& operator=(const DimensionConvertor&) = delete;

    template <class MatLike>
    void setCameraParameters(const MatLike& intrinsic, int width, int height)
    {
        double K[9];
        intrinsic_to_array(intrinsic, K);
        check(kde_dimconv_set_camera(h_, K, width, height));
    }
    void projectiveToReal(float* data, float3* out)
    {
        check(kde_dimconv_projective_to_real_depth(h_, 1, data, reinterpret_cast<kde_float3*>(out), stream_));
    }
    void projectiveToReal(float3* data, float3* out)
    {
        check(kde_dimconv_projective_to_real_points(h_, 1, reinterpret_cast<const kde_float3*>(data),
                                                    reinterpret_cast<kde_float3*>(out), stream_));
    }
    void projectiveToRealInterp(float* data, float3* out)
    {
        check(kde_dimconv_projective_to_real_interp(h_, 1, data, reinterpret_cast<kde_float3*>(out), stream_));
    }
    void realToProjective(float3* data, float3* out)
    {
        check(kde_dimconv_real_to_projective(h_, 1, reinterpret_cast<const kde_float3*>(data),
                                             reinterpret_cast<kde_float3*>(out), stream_));
    }
    void setStream(void* hip_stream) { stream_ = hip_stream; }

private:
    kde_dimconv* h_ = nullptr;
    void* stream_ = nullptr;
};

// ------------------------------------------------------------------------------------------------
class ArrayBuffer {
public:
    typedef kde_weighted_d weighted_d;   // ArrayBuffer.h:12-15
    virtual ~ArrayBuffer() {}
    virtual void insertData(float* data) = 0;
    virtual void insertData(weighted_d* data) = 0;
    virtual void getDepthMap(float* out) = 0;
    virtual void getWeightMap(float* out) = 0;
    virtual void updateData(float* data) = 0;
};

class Buffer2D : public ArrayBuffer {
public:
    explicit Buffer2D(int width, int height) { check(kde_buffer2d_create(&h_, width, height)); }
    ~Buffer2D() override { kde_buffer2d_destroy(h_); }
    Buffer2D(const Buffer2D&) = delete;
    Buffer2D& operator=(const Buffer2D&) = delete;

    void insertData(float* data) override { check(kde_buffer2d_insert_depth(h_, data, stream_)); }
    void insertData(weighted_d* data) override { check(kde_buffer2d_insert_weighted(h_, data, stream_)); }
    void insertData(float2* data) { check(kde_buffer2d_insert_float2(h_, reinterpret_cast<const float*>(data), stream_)); }
    void getDepthMap(float* out) override { check(kde_buffer2d_get_depth_map(h_, out, stream_)); }
    void getWeightMap(float* out) override { check(kde_buffer2d_get_weight_map(h_, out, stream_)); }
    void updateData(float* data) override { check(kde_buffer2d_update(h_, data, stream_)); }
    // extension: n_frames consecutive frames fused into one pass over the buffer
    void updateData(float* data, int n_frames) { check(kde_buffer2d_update_sequence(h_, n_frames, data, stream_)); }
    weighted_d* getRawPointer()
    {
        weighted_d* p = nullptr;
        check(kde_buffer2d_raw_pointer(h_, &p));
        return p;
    }
    void setStream(void* hip_stream) { stream_ = hip_stream; }

private:
    kde_buffer2d* h_ = nullptr;
    void* stream_ = nullptr;
};

// ------------------------------------------------------------------------------------------------
class DepthAdaptiveSuperpixel {
public:
    typedef kde_superpixel superpixel;             // SuperpixelSegmentation.h:17-24
    typedef kde_label_distance label_distance;     // :26-29
    DepthAdaptiveSuperpixel(int width, int height) : Width(width), Height(height) { check(kde_dasp_create(&h_, width, height)); }
    virtual ~DepthAdaptiveSuperpixel() { kde_dasp_destroy(h_); }
    DepthAdaptiveSuperpixel(const DepthAdaptiveSuperpixel&) = delete;
    DepthAdaptiveSuperpixel& operator=(const DepthAdaptiveSuperpixel&) = delete;

    template <class MatLike>
    void SetParametor(int rows, int cols, const MatLike& intrinsic)   // [sic]
    {
        double K[9];
        intrinsic_to_array(intrinsic, K);
        check(kde_dasp_set_parameters(h_, rows, cols, K));
    }
    template <class GpuMatLike>
    void Segmentation(const GpuMatLike& color_image, float3* points3d_device, float color_sigma,
                      float spatial_sigma, float depth_sigma, int iteration)
    {
        require_continuous_8uc3(color_image, Width, Height, "DepthAdaptiveSuperpixel::Segmentation");
        check(kde_dasp_segmentation(h_, color_image.data, reinterpret_cast<const kde_float3*>(points3d_device),
                                    color_sigma, spatial_sigma, depth_sigma, iteration, stream_));
    }
    int* getLabelDevice()
    {
        int32_t* p = nullptr;
        check(kde_dasp_labels_device(h_, &p));
        return p;
    }
    superpixel* getMeanDataDevice()
    {
        superpixel* p = nullptr;
        check(kde_dasp_mean_device(h_, &p));
        return p;
    }
    // ---- viewer members of the base class (SuperpixelSegmentation.h:36-41, .cpp:50-200); host side, no window, no video ----
    static const int Line = 0, Average = 1;
    // options == Line: the input with segment borders in white; otherwise every pixel in its cluster's mean colour
    // (.r,.g,.b into channels 0,1,2 as the reference writes them), label -1 black
    template <class ImageLike>
    HostImage8UC3& getSegmentedImage(const ImageLike& input_host, int options)
    {
        const int32_t* labels = nullptr;
        check(kde_dasp_labels_host(h_, stream_, &labels));
        if (SegmentedColor.rows != Height) SegmentedColor = HostImage8UC3(Height, Width);
        if (options == Line) {
            viewers::copy_from(input_host, SegmentedColor);
            viewers::mark_label_borders(labels, SegmentedColor);
        } else {
            const superpixel* mean = nullptr;
            int count = 0;
            check(kde_dasp_mean_host(h_, stream_, &mean, &count));
            for (int y = 0; y < Height; y++)
                for (int x = 0; x < Width; x++) {
                    const int32_t id = labels[static_cast<size_t>(y) * Width + x];
                    if (id >= 0 && id < count) SegmentedColor.set(y, x, mean[id].r, mean[id].g, mean[id].b);
                    else SegmentedColor.set(y, x, 0, 0, 0);
                }
        }
        return SegmentedColor;
    }
    HostImage8UC3& getRandomColorImage()
    {
        const int32_t* labels = nullptr;
        check(kde_dasp_labels_host(h_, stream_, &labels));
        if (SegmentedRandomColor.rows != Height) SegmentedRandomColor = HostImage8UC3(Height, Width);
        viewers::render_random_colours(labels, SegmentedRandomColor);
        return SegmentedRandomColor;
    }
    void releaseVideo() {}      // the reference's constructor opens an .avi writer (SuperpixelSegmentation.cpp:9); this one does not
    void setStream(void* hip_stream) { stream_ = hip_stream; }

private:
    int Width, Height;
    kde_dasp* h_ = nullptr;
    void* stream_ = nullptr;
    HostImage8UC3 SegmentedColor, SegmentedRandomColor;
};

// ------------------------------------------------------------------------------------------------
class EdgeRefinedSuperpixel {
public:
    EdgeRefinedSuperpixel(int width, int height) : Width(width), Height(height) { check(kde_ers_create(&h_, width, height)); }
    ~EdgeRefinedSuperpixel() { kde_ers_destroy(h_); }
    EdgeRefinedSuperpixel(const EdgeRefinedSuperpixel&) = delete;
    EdgeRefinedSuperpixel& operator=(const EdgeRefinedSuperpixel&) = delete;

    template <class GpuMatLike>
    void EdgeRefining(int* color_label_device, int* depth_label_device, float* depth_device, const GpuMatLike& color_image)
    {
        require_continuous_8uc3(color_image, Width, Height, "EdgeRefinedSuperpixel::EdgeRefining");
        check(kde_ers_edge_refining(h_, color_label_device, depth_label_device, depth_device, color_image.data, stream_));
    }
    int* getRefinedLabels_Device()
    {
        int32_t* p = nullptr;
        check(kde_ers_refined_labels_device(h_, &p));
        return p;
    }
    int* getRefinedLabels_Host()
    {
        const int32_t* p = nullptr;
        check(kde_ers_refined_labels_host(h_, stream_, &p));
        return const_cast<int*>(p);
    }
    float* getRefinedDepth_Device()
    {
        float* p = nullptr;
        check(kde_ers_refined_depth_device(h_, &p));
        return p;
    }
    float* getRefinedDepth_Host()
    {
        const float* p = nullptr;
        check(kde_ers_refined_depth_host(h_, stream_, &p));
        return const_cast<float*>(p);
    }
    // ---- viewer members (EdgeRefinedSuperpixel.h:27-29, .cpp:70-147); host side ----
    // refined depth on the reference's ramp over 3000 mm (`max_depth` is ignored there too, .cpp:79-80); then, except in the
    // last row / column: depth == 0 black, segment borders and label -100 white
    HostImage8UC3& getSegmentedImage(const int max_depth)
    {
        (void)max_depth;
        const float* depth = getRefinedDepth_Host();
        const int32_t* labels = getRefinedLabels_Host();
        if (segmentedImage.rows != Height) segmentedImage = HostImage8UC3(Height, Width);
        viewers::render_depth(depth, 3000.0f, false, 0.0f, segmentedImage);
        for (int y = 0; y + 1 < Height; y++)
            for (int x = 0; x + 1 < Width; x++) {
                const size_t q = static_cast<size_t>(y) * Width + x;
                if (depth[q] == 0.0f) segmentedImage.set(y, x, 0, 0, 0);
                if (labels[q] != labels[q + Width] || labels[q] != labels[q + 1] || labels[q] == -100) segmentedImage.set(y, x, 255, 255, 255);
            }
        return segmentedImage;
    }
    template <class ImageLike>
    HostImage8UC3& getSegmentedImage(const ImageLike& input_host)
    {
        if (SegmentedColor.rows != Height) SegmentedColor = HostImage8UC3(Height, Width);
        viewers::copy_from(input_host, SegmentedColor);
        viewers::mark_label_borders(getRefinedLabels_Host(), SegmentedColor);
        return SegmentedColor;
    }
    HostImage8UC3& getRandomColorImage()
    {
        if (SegmentedRandomColor.rows != Height) SegmentedRandomColor = HostImage8UC3(Height, Width);
        viewers::render_random_colours(getRefinedLabels_Host(), SegmentedRandomColor);
        return SegmentedRandomColor;
    }
    void setStream(void* hip_stream) { stream_ = hip_stream; }

private:
    int Width, Height;
    kde_ers* h_ = nullptr;
    void* stream_ = nullptr;
    HostImage8UC3 segmentedImage, SegmentedColor, SegmentedRandomColor;
};

// ------------------------------------------------------------------------------------------------
class RegionGrowingBilateralFilter {
public:
    RegionGrowingBilateralFilter(int width, int height) : Width(width), Height(height) { check(kde_rgbf_create(&h_, width, height)); }
    // extension: buffers for max_batch independent frames per ProcessBatch (the reference processes one frame per call)
    RegionGrowingBilateralFilter(int width, int height, int max_batch) : Width(width), Height(height)
    {
        check(kde_rgbf_create_batch(&h_, width, height, max_batch));
    }
    ~RegionGrowingBilateralFilter() { kde_rgbf_destroy(h_); }
    RegionGrowingBilateralFilter(const RegionGrowingBilateralFilter&) = delete;
    RegionGrowingBilateralFilter& operator=(const RegionGrowingBilateralFilter&) = delete;

    template <class MatLike>
    void SetParametor(int rows, int cols, const MatLike& intrinsic)   // [sic]
    {
        double K[9];
        intrinsic_to_array(intrinsic, K);
        check(kde_rgbf_set_parameters(h_, rows, cols, K));
    }
    template <class GpuMatLike>
    void Process(float* depth_device, float3* points_device, const GpuMatLike& color_device)
    {
        require_continuous_8uc3(color_device, Width, Height, "RegionGrowingBilateralFilter::Process");
        check(kde_rgbf_process(h_, depth_device, reinterpret_cast<const kde_float3*>(points_device), color_device.data, stream_));
    }
    // n frames back to back in every argument; the getters then return n frames back to back, each bit-identical to Process
    void ProcessBatch(int n, const float* depth_device, const float3* points_device, const uint8_t* bgr_device)
    {
        check(kde_rgbf_process_batch(h_, n, depth_device, reinterpret_cast<const kde_float3*>(points_device), bgr_device, stream_));
    }
    float* getRefinedDepth_Device()
    {
        float* p = nullptr;
        check(kde_rgbf_refined_depth_device(h_, &p));
        return p;
    }
    float* getRefinedDepth_Host()
    {
        const float* p = nullptr;
        check(kde_rgbf_refined_depth_host(h_, stream_, &p));
        return const_cast<float*>(p);
    }
    int* getRefinedLabels_Device()
    {
        int32_t* p = nullptr;
        check(kde_rgbf_refined_labels_device(h_, &p));
        return p;
    }
    void setStream(void* hip_stream) { stream_ = hip_stream; }

private:
    int Width, Height;
    kde_rgbf* h_ = nullptr;
    void* stream_ = nullptr;
};

// ------------------------------------------------------------------------------------------------
class SPDepthSuperResolution {
public:
    SPDepthSuperResolution(int width, int height) : Width(width), Height(height) { check(kde_spdsr_create(&h_, width, height)); }
    SPDepthSuperResolution(int width, int height, int max_batch) : Width(width), Height(height)      // extension, as RGBF's
    {
        check(kde_spdsr_create_batch(&h_, width, height, max_batch));
    }
    ~SPDepthSuperResolution() { kde_spdsr_destroy(h_); }
    SPDepthSuperResolution(const SPDepthSuperResolution&) = delete;
    SPDepthSuperResolution& operator=(const SPDepthSuperResolution&) = delete;

    template <class MatLike>
    void SetParametor(int rows, int cols, const MatLike& intrinsic)   // [sic]
    {
        double K[9];
        intrinsic_to_array(intrinsic, K);
        check(kde_spdsr_set_parameters(h_, rows, cols, K));
    }
    template <class GpuMatLike>
    void Process(float* depth_device, float3* points_device, const GpuMatLike& color_device)
    {
        require_continuous_8uc3(color_device, Width, Height, "SPDepthSuperResolution::Process");
        check(kde_spdsr_process(h_, depth_device, reinterpret_cast<const kde_float3*>(points_device), color_device.data, stream_));
    }
    void ProcessBatch(int n, const float* depth_device, const float3* points_device, const uint8_t* bgr_device)
    {
        check(kde_spdsr_process_batch(h_, n, depth_device, reinterpret_cast<const kde_float3*>(points_device), bgr_device, stream_));
    }
    float* getRefinedDepth_Device()
    {
        float* p = nullptr;
        check(kde_spdsr_refined_depth_device(h_, &p));
        return p;
    }
    float* getRefinedDepth_Host()
    {
        const float* p = nullptr;
        check(kde_spdsr_refined_depth_host(h_, stream_, &p));
        return const_cast<float*>(p);
    }
    float3* getEdgeEnhanced3DPoints_Device()
    {
        kde_float3* p = nullptr;
        check(kde_spdsr_edge_enhanced_points_device(h_, &p));
        return reinterpret_cast<float3*>(p);
    }
    float3* getOptimizedPoints_Device()
    {
        kde_float3* p = nullptr;
        check(kde_spdsr_optimized_points_device(h_, &p));
        return reinterpret_cast<float3*>(p);
    }
    float3* getOptimizedPoints_Host()
    {
        const kde_float3* p = nullptr;
        check(kde_spdsr_optimized_points_host(h_, stream_, &p));
        return reinterpret_cast<float3*>(const_cast<kde_float3*>(p));
    }
    void setStream(void* hip_stream) { stream_ = hip_stream; }

private:
    int Width, Height;
    kde_spdsr* h_ = nullptr;
    void* stream_ = nullptr;
};

}  // namespace ref
}  // namespace kde

#ifndef KDE_NO_GLOBAL_NAMES
using kde::ref::ArrayBuffer;
using kde::ref::Buffer2D;
using kde::ref::DepthAdaptiveSuperpixel;
using kde::ref::DimensionConvertor;
using kde::ref::EdgeRefinedSuperpixel;
using kde::ref::JointBilateralFilter;
using kde::ref::MarkovRandomField;
using kde::ref::RegionGrowingBilateralFilter;
using kde::ref::SPDepthSuperResolution;
#endif

#endif  // KDE_KDE_HPP
