/*
 * kde_hip.h — C ABI of libkde_hip.so: the MI355X (gfx950) depth-enhancement filter stage.
 *
 * Drop-in boundary for the hot path of stevesuyao/KinectDepthMapEnhancement
 *   DimensionConvertor / Buffer2D -> JointBilateralFilter::Process -> RegionGrowingBilateralFilter::Process
 * The reference has no FFI; its boundary is the public C++ class surface.  Each entry
 * point below names the reference member (file:line, relative to the reference root) it replaces.
 * include/kde/ holds header-only C++ classes with the reference's names and signatures that
 * forward to this ABI (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C types only: device pointers are raw pointers into HIP device memory owned by the
 *     caller unless a getter says "object-owned"; streams are passed as void* (a hipStream_t,
 *     NULL = the null stream).  Every call is asynchronous on that stream unless stated.
 *   - every function returns KDE_OK (0) or a KDE_ERR_* code and never aborts;
 *     kde_last_error_string() describes the last failure on the calling thread.
 *   - colour images are packed 8UC3 BGR ("CV_8UC3 continuous", cv::gpu::createContinuous,
 *     JointBilateralFilter.cpp:13); depth is float32 millimetres, row-major W x H;
 *     "valid" means depth > 50.0f everywhere (JointBilateralFilter.cu:21).
 *   - a handle is one stream-ordered context (scratch buffers are members, like the reference
 *     objects); handles are independent, a single handle is not thread-safe.
 *   - a handle belongs to the device that was current when it was created (kde_set_device): calls
 *     made while another device is current return KDE_ERR_INVALID instead of launching on it.
 *   - batched entry points take n independent frames laid out back to back
 *     (frame f at base + f * W*H elements) and are the unit that is sharded across GPUs.
 */
#ifndef KDE_HIP_H
#define KDE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KDE_ABI_VERSION 1

enum {
    KDE_OK = 0,
    KDE_ERR_INVALID = 1,      /* bad argument / unsupported geometry            */
    KDE_ERR_HIP = 2,          /* a HIP runtime call failed                      */
    KDE_ERR_NOMEM = 3,        /* host or device allocation failed               */
    KDE_ERR_UNSUPPORTED = 4   /* feature outside the built scope                */
};

/* ---- device-visible record layouts (identical to the reference's) ------------------------- */
typedef struct kde_float3 { float x, y, z; } kde_float3;            /* CUDA float3, 12 B packed          */
typedef struct kde_weighted_d { float d, w; } kde_weighted_d;       /* ArrayBuffer/ArrayBuffer.h:12-15    */
typedef struct kde_superpixel {                                     /* SuperpixelSegmentation.h:17-24     */
    uint8_t r, g, b, pad_;
    int32_t x, y, size;
} kde_superpixel;                                                   /* 16 B                               */
typedef struct kde_label_distance { float d; int32_t l; } kde_label_distance; /* SuperpixelSegmentation.h:26-29 */

/* ---- library ------------------------------------------------------------------------------- */
int kde_abi_version(void);
const char* kde_last_error_string(void);
int kde_device_count(int* count);
int kde_set_device(int device);
/* name/arch of the current device, e.g. "gfx950:sramecc+:xnack-"; buf may be NULL to query cu_count only */
int kde_device_info(char* arch_buf, size_t arch_cap, int* cu_count);
/* PCI address of the current device ("0000:05:00.0", hipDeviceGetPCIBusId): what tells two GPUs of a node apart.
 * The reference is single-device (main.cpp:160-163); the sharding hosts put it in their report so that a run
 * over N devices shows N distinct addresses */
int kde_device_pci_bus_id(char* buf, size_t cap);

/* ============================================================================================
 * JointBilateralFilter — JointBilateralFilter/JointBilateralFilter.{h,cpp,cu}
 * ========================================================================================== */
typedef struct kde_jbf_params {
    int   window_size;              /* WindowSize = 5     JointBilateralFilter.cpp:3  (odd, 1..31) */
    float spatial_sigma;            /* SpatialSigma = 70  :4  (pixels)                              */
    float color_sigma;              /* ColorSigma = 50    :5  (0..255 colour levels; 0 = term off)  */
    float depth_sigma;              /* DepthSigma = 20    :6  (millimetres; 0 = term off)           */
    int   presmooth;                /* 1: guide = cv::gpu::bilateralFilter(colour) (.cu:285); 0: guide = colour */
    int   presmooth_kernel_size;    /* 5    (.cu:285)                                               */
    float presmooth_sigma_color;    /* 30.0 (.cu:285)                                               */
    float presmooth_sigma_spatial;  /* 30.0 (.cu:285)                                               */
} kde_jbf_params;

typedef struct kde_jbf kde_jbf;

/* fills the reference's compile-time constants */
int kde_jbf_default_params(kde_jbf_params* p);
/* JointBilateralFilter::JointBilateralFilter(width, height) + calcSpatialFilter()
 * (JointBilateralFilter.cpp:8-20, 31-40).  params == NULL -> defaults.  max_batch >= 1 sizes the
 * object-owned output / guide buffers for the batched entry points. */
int kde_jbf_create(kde_jbf** out, int width, int height, int max_batch, const kde_jbf_params* params);
/* JointBilateralFilter::~JointBilateralFilter (JointBilateralFilter.cpp:21-30) */
int kde_jbf_destroy(kde_jbf* h);
/* void JointBilateralFilter::Process(float* depth_device, cv::gpu::GpuMat color_image)
 * (JointBilateralFilter.cu:283-290).  bgr_step is GpuMat::step and must equal 3*width.
 * Depth is in millimetres; samples that are NaN, -inf or <= 50 are absent taps, as in the reference.  +inf and samples
 * above 2^64 make every output whose window holds them non-finite garbage in the reference; the tuned kernels keep all
 * other pixels exact to the usual bar but do not reproduce that garbage class for class (kde_jbf_set_variant(h, 0)
 * does).  DESIGN.md section 3, "input domain". */
int kde_jbf_process(kde_jbf* h, const float* depth_dev, const uint8_t* bgr_dev, size_t bgr_step, void* stream);
/* the same over n <= max_batch independent frames; filtered_dev == NULL writes the object-owned buffer */
int kde_jbf_process_batch(kde_jbf* h, int n, const float* depth_dev, const uint8_t* bgr_dev,
                          float* filtered_dev, void* stream);
/* the two kernels of Process individually (K0 = pre-smoothing, K1 = joint_bilateral_filtering):
 * used by the roofline benchmark and the per-kernel parity tests */
int kde_jbf_presmooth_batch(kde_jbf* h, int n, const uint8_t* bgr_dev, uint8_t* smooth_dev, void* stream);
int kde_jbf_filter_batch(kde_jbf* h, int n, const float* depth_dev, const uint8_t* guide_bgr_dev,
                         float* filtered_dev, void* stream);
/* float* getFiltered_Device() const (JointBilateralFilter.cpp:41-43): object-owned, valid until the
 * next Process or destroy */
int kde_jbf_filtered_device(kde_jbf* h, float** out);
/* float* getFiltered_Host() const (:44-46).  Unlike the reference (stale unless visualize() ran) the
 * pinned host copy is refreshed here: synchronises `stream` and copies the frames the last call wrote into the
 * object's own Filtered_Device (at most max_batch).  Results a caller directed into its own buffer
 * (filtered_dev != NULL) are not mirrored -- the object keeps no pointer to caller-owned memory. */
int kde_jbf_filtered_host(kde_jbf* h, void* stream, const float** out);
/* cv::gpu::GpuMat getSmoothImage_Device() (:47-49): object-owned packed BGR, step = 3*width */
int kde_jbf_smooth_device(kde_jbf* h, uint8_t** out);
/* the window_size^2 spatial table as uploaded to the device (calcSpatialFilter) */
int kde_jbf_spatial_table(kde_jbf* h, float* table_host, int capacity);
/* tuning knob for the LDS tile sweep (BASELINE config 3); variant -1 = built-in choice */
int kde_jbf_set_variant(kde_jbf* h, int variant);
/* which kernel kde_jbf_process / kde_jbf_filter_batch will launch for this handle's parameters and variant setting:
 * an index into kde_jbf_variant_name(); 0 = "generic-32x8-1px" (one pixel per thread, any odd window <= 31, zero sigmas).
 * Tuned kernels exist for every odd window from 3 to 31 with non-zero sigmas (23..31 read their log2(S) table from a device
 * copy the handle uploads at creation: it no longer fits the 4 KB kernel-argument block); the reference takes
 * window_size as a run-time argument (JointBilateralFilter.cu:10,18-19) */
int kde_jbf_active_variant(kde_jbf* h, int* variant);
int kde_jbf_variant_count(void);
const char* kde_jbf_variant_name(int variant);

/* ============================================================================================
 * MarkovRandomField — MarkovRandomField/MarkovRandomField.{cpp,cu} (sibling filter)
 * ========================================================================================== */
typedef struct kde_mrf kde_mrf;
/* MarkovRandomField(width,height); constants MarkovRandomField.cpp:3-6 (window 5, ColorSigma 50, SmoothSigma 150);
 * pass window<=0 / negative sigmas for the defaults */
int kde_mrf_create(kde_mrf** out, int width, int height, int max_batch, int window, float color_sigma, float smooth_sigma);
int kde_mrf_destroy(kde_mrf* h);
/* void MarkovRandomField::Process(float* depth_device, cv::gpu::GpuMat color_image) (MarkovRandomField.cu:42-49) */
int kde_mrf_process_batch(kde_mrf* h, int n, const float* depth_dev, const uint8_t* bgr_dev, float* filtered_dev, void* stream);
int kde_mrf_filtered_device(kde_mrf* h, float** out);
/* float* getFiltered_Host() (MarkovRandomField.h:16; refreshed after every Process in the reference,
 * MarkovRandomField.cu:48): object-owned pinned memory, lazily copied from the object's own Filtered_Device here;
 * synchronises the stream.  Holds the frames of the last call that wrote Filtered_Device. */
int kde_mrf_filtered_host(kde_mrf* h, void* stream, const float** out);

/* ============================================================================================
 * DimensionConvertor — DimensionConvertor/DimensionConvertor.{h,cpp,cu}
 * ========================================================================================== */
typedef struct kde_dimconv kde_dimconv;
int kde_dimconv_create(kde_dimconv** out);
int kde_dimconv_destroy(kde_dimconv* h);
/* void setCameraParameters(const cv::Mat_<double> intrinsic, int width, int height) (DimensionConvertor.cpp:3-13):
 * K is the row-major 3x3 intrinsic matrix; Fx,Fy = (float)K00,K11; Cx,Cy = (int)K02,K12 (truncated) */
int kde_dimconv_set_camera(kde_dimconv* h, const double* K9, int width, int height);
/* void projectiveToReal(float* data, float3* out) (DimensionConvertor.cu:3-23); n frames */
int kde_dimconv_projective_to_real_depth(kde_dimconv* h, int n, const float* depth_dev, kde_float3* out_dev, void* stream);
/* void projectiveToReal(float3* data, float3* out) (DimensionConvertor.cu:25-33) */
int kde_dimconv_projective_to_real_points(kde_dimconv* h, int n, const kde_float3* in_dev, kde_float3* out_dev, void* stream);
/* void projectiveToRealInterp(float* data, float3* out) (DimensionConvertor.cu:44-77) */
int kde_dimconv_projective_to_real_interp(kde_dimconv* h, int n, const float* depth_dev, kde_float3* out_dev, void* stream);
/* void realToProjective(float3* data, float3* out) (DimensionConvertor.cu:35-43) */
int kde_dimconv_real_to_projective(kde_dimconv* h, int n, const kde_float3* in_dev, kde_float3* out_dev, void* stream);

/* ============================================================================================
 * Buffer2D / ArrayBuffer — ArrayBuffer/{ArrayBuffer,Buffer2D}.{h,cpp,cu}
 * ========================================================================================== */
typedef struct kde_buffer2d kde_buffer2d;
/* Buffer2D(width,height): allocates weighted_d[W*H] and zero-initialises it (Buffer2D.cpp:4-11, ArrayBuffer.cu:9-30) */
int kde_buffer2d_create(kde_buffer2d** out, int width, int height);
int kde_buffer2d_destroy(kde_buffer2d* h);
/* void insertData(float* data) (Buffer2D.cu:33-56): d = data, w = 1 */
int kde_buffer2d_insert_depth(kde_buffer2d* h, const float* depth_dev, void* stream);
/* void insertData(float2* data) (Buffer2D.cu:123-147): d = data.x, w = (float)row  [sic, :137] */
int kde_buffer2d_insert_float2(kde_buffer2d* h, const float* xy_dev, void* stream);
/* void insertData(weighted_d* data) (Buffer2D.cpp:13-15): device-to-device copy */
int kde_buffer2d_insert_weighted(kde_buffer2d* h, const kde_weighted_d* data_dev, void* stream);
/* void getDepthMap(float* out) (Buffer2D.cu:59-77) / getWeightMap (:79-94) */
int kde_buffer2d_get_depth_map(kde_buffer2d* h, float* out_dev, void* stream);
int kde_buffer2d_get_weight_map(kde_buffer2d* h, float* out_dev, void* stream);
/* void updateData(float* data) (Buffer2D.cu:97-120 -> updateWaitedDepth :13-30) */
int kde_buffer2d_update(kde_buffer2d* h, const float* depth_dev, void* stream);
/* the same for a sequence of n_frames frames (frame f at depth_dev + f*W*H), fused into one pass
 * over the buffer: the streaming temporal-fusion front end (main.cpp:86-116) */
int kde_buffer2d_update_sequence(kde_buffer2d* h, int n_frames, const float* depth_dev, void* stream);
/* weighted_d* getRawPointer() (ArrayBuffer.cpp:19-21): object-owned */
int kde_buffer2d_raw_pointer(kde_buffer2d* h, kde_weighted_d** out);

/* ============================================================================================
 * DepthAdaptiveSuperpixel — SuperpixelSegmentation/DepthAdaptiveSuperpixel.{h,cpp,cu}
 * (+ base SuperpixelSegmentation.{h,cpp}); DASP path only
 * ========================================================================================== */
typedef struct kde_dasp kde_dasp;
/* DepthAdaptiveSuperpixel(width,height) (DepthAdaptiveSuperpixel.cpp:4-8) */
int kde_dasp_create(kde_dasp** out, int width, int height);
int kde_dasp_destroy(kde_dasp* h);
/* void SetParametor(int rows, int cols, cv::Mat_<double> intrinsic) (DepthAdaptiveSuperpixel.cpp:15-39).
 * Rejects geometries the reference would index out of bounds with (KDE_ERR_INVALID):
 * width/cols >= 4, height/rows >= 4, width/(width/cols) == cols, height >= 6. */
int kde_dasp_set_parameters(kde_dasp* h, int rows, int cols, const double* K9);
/* void Segmentation(GpuMat color, float3* points3d, float color_sigma, float spatial_sigma,
 *                   float depth_sigma, int iteration) (DepthAdaptiveSuperpixel.cu:570-588) */
int kde_dasp_segmentation(kde_dasp* h, const uint8_t* bgr_dev, const kde_float3* points_dev,
                          float color_sigma, float spatial_sigma, float depth_sigma, int iteration, void* stream);
/* int* getLabelDevice() / superpixel* getMeanDataDevice() (SuperpixelSegmentation.cpp) + DASP members */
int kde_dasp_labels_device(kde_dasp* h, int32_t** out);
int kde_dasp_mean_device(kde_dasp* h, kde_superpixel** out);
int kde_dasp_centers_device(kde_dasp* h, kde_float3** out);
int kde_dasp_ld_device(kde_dasp* h, kde_label_distance** out);
/* Labels_Host: refreshed by a blocking copy like DepthAdaptiveSuperpixel.cu:587, but lazily */
int kde_dasp_labels_host(kde_dasp* h, void* stream, const int32_t** out);
/* meanData_Host: rows*cols records, refreshed by a blocking copy as the viewers do it (SuperpixelSegmentation.cpp:97) */
int kde_dasp_mean_host(kde_dasp* h, void* stream, const kde_superpixel** out, int* count);

/* ============================================================================================
 * EdgeRefinedSuperpixel — EdgeRefinedSuperpixel/EdgeRefinedSuperpixel.{h,cpp,cu}
 * ========================================================================================== */
typedef struct kde_ers kde_ers;
/* EdgeRefinedSuperpixel(width,height) + calcSpatialFilter (EdgeRefinedSuperpixel.cpp:9-55);
 * constants WindowSize 7, SpatialSigma 30, ColorSigma 50, DepthSigma 70 (:4-7) */
int kde_ers_create(kde_ers** out, int width, int height);
int kde_ers_destroy(kde_ers* h);
/* void EdgeRefining(int* color_label_device, int* depth_label_device, float* depth_device,
 *                   cv::gpu::GpuMat color_image) (EdgeRefinedSuperpixel.cu:208-223).
 * Labels are what the segmenters write: -1 (unassigned) or a superpixel index in [0, width * height).
 * Depth domain as for kde_jbf_process (+inf / > 2^64 samples: kde_ers_set_variant(h, 3) reproduces the reference). */
int kde_ers_edge_refining(kde_ers* h, const int32_t* color_labels_dev, const int32_t* depth_labels_dev,
                          const float* depth_dev, const uint8_t* bgr_dev, void* stream);
/* (no reference counterpart) which depthmap_enhancement kernel serves the handle: 0 = built-in choice,
 * 1 = packed-pair kernel (labels are superpixel indices, exact as floats for frames of <= 2^24 pixels),
 * 2 = scalar tuned kernel, 3 = generic kernels (any-window depthmap_enhancement, and edge_refining as two
 * launches on global memory instead of the fused LDS kernel).  For A/B measurements and the per-kernel parity tests. */
int kde_ers_set_variant(kde_ers* h, int variant);
/* the two kernels individually, for per-kernel parity tests:
 * edge_refining (.cu:4-102; snapshot semantics) on the object-owned label/depth copies,
 * leaving its result readable through kde_ers_stage_* ; depthmap_enhancement (.cu:104-205; D3) */
int kde_ers_stage_edge_depth_device(kde_ers* h, float** out);
int kde_ers_refined_labels_device(kde_ers* h, int32_t** out);    /* getRefinedLabels_Device (EdgeRefinedSuperpixel.cpp:56-58) */
int kde_ers_refined_depth_device(kde_ers* h, float** out);       /* getRefinedDepth_Device  (:63-65)                         */
int kde_ers_refined_labels_host(kde_ers* h, void* stream, const int32_t** out); /* getRefinedLabels_Host (:59-62) */
int kde_ers_refined_depth_host(kde_ers* h, void* stream, const float** out);    /* getRefinedDepth_Host  (:66-69) */

/* ============================================================================================
 * RegionGrowingBilateralFilter — RegionGrowingBilateralFilter.{h,cpp}
 * ========================================================================================== */
typedef struct kde_rgbf kde_rgbf;
int kde_rgbf_create(kde_rgbf** out, int width, int height);                       /* .cpp:5-11  */
int kde_rgbf_destroy(kde_rgbf* h);                                                /* .cpp:12-20 */
int kde_rgbf_set_parameters(kde_rgbf* h, int rows, int cols, const double* K9);   /* SetParametor, .cpp:21-26 */
/* void Process(float* depth_device, float3* points_device, cv::gpu::GpuMat color_device) (.cpp:27-38):
 * SP->Segmentation(200,40,0,1); DASP->Segmentation(100,20,200,1); ERS->EdgeRefining(SP labels, DASP labels, ...) */
int kde_rgbf_process(kde_rgbf* h, const float* depth_dev, const kde_float3* points_dev,
                     const uint8_t* bgr_dev, void* stream);
/* A batch of independent frames (the unit north_star shards across GPUs; the reference has only the single-frame call):
 * kde_rgbf_create_batch sizes the object's buffers for max_batch frames, kde_rgbf_process_batch runs Process on n <=
 * max_batch frames laid out back to back -- every kernel of the chain takes the whole batch in one launch, and each
 * frame's labels and refined depth are bit-identical to its single-frame kde_rgbf_process.  The getters then return
 * n frames back to back (frame f at + f * W*H). */
int kde_rgbf_create_batch(kde_rgbf** out, int width, int height, int max_batch);
int kde_rgbf_process_batch(kde_rgbf* h, int n, const float* depth_dev, const kde_float3* points_dev,
                           const uint8_t* bgr_dev, void* stream);
int kde_rgbf_refined_depth_device(kde_rgbf* h, float** out);                       /* .cpp:39-41 */
int kde_rgbf_refined_depth_host(kde_rgbf* h, void* stream, const float** out);     /* .cpp:42-44 */
int kde_rgbf_refined_labels_device(kde_rgbf* h, int32_t** out);
int kde_rgbf_sp_labels_device(kde_rgbf* h, int32_t** out);
int kde_rgbf_dasp_labels_device(kde_rgbf* h, int32_t** out);

/* ============================================================================================
 * SPDepthSuperResolution — SPDepthSuperResolution.{h,cpp}: class surface named by the north star.
 * Process = .cpp:57-191: SP(200,10,0,5), DASP(0,10,200,5), ERS, projectiveToReal, per-superpixel plane fit
 * (the reference's host cv::PCA loop, here on the device) and Projection_GPU::PlaneProjection(nd, labels,
 * points) (Projection_GPU/Projection_GPU.cu:55-81, 148-187, 274-294; 20 sweeps, snapshot semantics D5).
 * ========================================================================================== */
typedef struct kde_spdsr kde_spdsr;
int kde_spdsr_create(kde_spdsr** out, int width, int height);
int kde_spdsr_destroy(kde_spdsr* h);
int kde_spdsr_set_parameters(kde_spdsr* h, int rows, int cols, const double* K9);
int kde_spdsr_process(kde_spdsr* h, const float* depth_dev, const kde_float3* points_dev,
                      const uint8_t* bgr_dev, void* stream);
/* batched form, as kde_rgbf_process_batch: head (labels, refined depth, edge-enhanced points) bit-identical per frame;
 * the plane fit accumulates double-precision moments with atomics, so the tail agrees with the single-frame call to
 * the same 1e-4 it agrees with itself from run to run.  ClusterND: n tables of rows*cols float4 back to back. */
int kde_spdsr_create_batch(kde_spdsr** out, int width, int height, int max_batch);
int kde_spdsr_process_batch(kde_spdsr* h, int n, const float* depth_dev, const kde_float3* points_dev,
                            const uint8_t* bgr_dev, void* stream);
int kde_spdsr_refined_depth_device(kde_spdsr* h, float** out);
int kde_spdsr_refined_depth_host(kde_spdsr* h, void* stream, const float** out);
int kde_spdsr_refined_labels_device(kde_spdsr* h, int32_t** out);
int kde_spdsr_edge_enhanced_points_device(kde_spdsr* h, kde_float3** out);         /* EdgeEnhanced3DPoints_Device */
int kde_spdsr_optimized_points_device(kde_spdsr* h, kde_float3** out);             /* getOptimizedPoints_Device (Projection_GPU.cpp:62-64) */
int kde_spdsr_optimized_points_host(kde_spdsr* h, void* stream, const kde_float3** out); /* getOptimizedPoints_Host (:59-61) */
int kde_spdsr_plane_fitted_points_device(kde_spdsr* h, kde_float3** out);         /* Projection_GPU::GetPlaneFitted3D_Device (:56-58) */
int kde_spdsr_cluster_nd_device(kde_spdsr* h, float** out);                       /* ClusterND_Device: float4 {normal, distance} per cluster */

#ifdef __cplusplus
}
#endif
#endif /* KDE_HIP_H */
