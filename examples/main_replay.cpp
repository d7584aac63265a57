// main_replay.cpp — the reference's offline evaluation sequence (main.cpp:159-197) on the drop-in
// classes of include/kde/kde.hpp, without OpenNI / OpenCV / PCL:
//   upload -> projectiveToReal(input) -> JBF.Process -> projectiveToReal(JBF) -> MRF.Process ->
//   projectiveToReal(MRF) -> RGBF.Process(input depth, input points) -> projectiveToReal(RGBF)
// and the mean 3-D error of every method against the cloud of the averaged depth (main.cpp:220-308).
//
// usage: main_replay <width> <height> <color.bgr> <depth.f32> <averaged_depth.f32> <out_prefix>
//        main_replay <width> <height> <color.bgr> <depth.xml> - <out_prefix>      (the reference's own depth.xml)
//   color.bgr: W*H*3 bytes packed BGR;  *.f32: W*H float32 millimetres;
//   depth.xml: OpenCV FileStorage with "averaged_depth" and "depth" (main.cpp:112-114,146-149)
// writes <out_prefix>{jbf,mrf,rgbf}.f32 and prints one "name error count" line per method.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "../include/kde/depth_xml.hpp"
#include "../include/kde/kde.hpp"

#define HIP_OK(x)                                                                          \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            std::exit(2);                                                                  \
        }                                                                                  \
    } while (0)

template <class T>
static std::vector<T> read_file(const std::string& path, size_t count)
{
    std::vector<T> v(count);
    std::ifstream f(path, std::ios::binary);
    if (!f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(count * sizeof(T)))) {
        std::fprintf(stderr, "cannot read %zu elements from %s\n", count, path.c_str());
        std::exit(2);
    }
    return v;
}

static void write_file(const std::string& path, const float* p, size_t count)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(p), (std::streamsize)(count * sizeof(float)));
}

// main.cpp:220-308: mean Euclidean distance over pixels with 50 < z < 15000 in both clouds (float accumulation)
static float mean_error(const std::vector<float3>& a, const std::vector<float3>& truth, int* count)
{
    float acc = 0.0f;
    int n = 0;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i].z > 50.0f && a[i].z < 15000.0f && truth[i].z > 50.0f && truth[i].z < 15000.0f) {
            const float dz = a[i].z - truth[i].z, dy = a[i].y - truth[i].y, dx = a[i].x - truth[i].x;
            acc += std::sqrt(dz * dz + dy * dy + dx * dx);
            n++;
        }
    }
    *count = n;
    return acc / (float)n;
}

int main(int argc, char** argv)
{
    if (argc != 7) {
        std::fprintf(stderr, "usage: %s width height color.bgr depth.f32 averaged_depth.f32 out_prefix\n", argv[0]);
        return 2;
    }
    const int W = std::atoi(argv[1]), H = std::atoi(argv[2]);
    const size_t N = (size_t)W * H;
    const std::string prefix = argv[6];
    try {
        const std::vector<uint8_t> color = read_file<uint8_t>(argv[3], N * 3);
        std::vector<float> depth, averaged;
        if (std::string(argv[5]) == "-") {
            kde::DepthMatrix dm, am;
            kde::read_depth_xml(argv[4], dm, am);
            if (dm.rows != H || dm.cols != W || am.rows != H || am.cols != W) throw std::runtime_error("depth.xml size mismatch");
            depth = dm.data;
            averaged = am.data;
        } else {
            depth = read_file<float>(argv[4], N);
            averaged = read_file<float>(argv[5], N);
        }

        float *inputDepth_Device, *bufferDepth_Device;
        float3 *inputPoints_Device, *bufferPoints_Device, *tmpPoints_Device;
        uint8_t* color_dev;
        HIP_OK(hipMalloc(&inputDepth_Device, N * sizeof(float)));
        HIP_OK(hipMalloc(&bufferDepth_Device, N * sizeof(float)));
        HIP_OK(hipMalloc(&inputPoints_Device, N * sizeof(float3)));
        HIP_OK(hipMalloc(&bufferPoints_Device, N * sizeof(float3)));
        HIP_OK(hipMalloc(&tmpPoints_Device, N * sizeof(float3)));
        HIP_OK(hipMalloc(&color_dev, N * 3));
        HIP_OK(hipMemcpy(inputDepth_Device, depth.data(), N * sizeof(float), hipMemcpyHostToDevice));     // main.cpp:160
        HIP_OK(hipMemcpy(bufferDepth_Device, averaged.data(), N * sizeof(float), hipMemcpyHostToDevice)); // :162
        HIP_OK(hipMemcpy(color_dev, color.data(), N * 3, hipMemcpyHostToDevice));                         // :163
        const kde::GpuImage8UC3 Color_Device{color_dev, H, W, (size_t)W * 3};

        // Kinect v1 nominal intrinsics scaled to the frame (Kinect/Kinect.cpp:89-95)
        const double f = 120.0 / (2.0 * 0.1042) * (W / 640.0);
        const kde::Mat33d K{{f, 0.0, W / 2.0, 0.0, f, H / 2.0, 0.0, 0.0, 1.0}};

        JointBilateralFilter JBF(W, H);                 // main.cpp:67
        MarkovRandomField MRF(W, H);                    // :69
        DimensionConvertor convertor;                   // :71-72
        convertor.setCameraParameters(K, W, H);
        RegionGrowingBilateralFilter RGBF(W, H);        // :74-75
        RGBF.SetParametor(15, 20, K);

        std::vector<float3> truth(N), cloud(N);
        std::vector<float> out(N);
        auto report = [&](const char* name, float* depth_dev, bool save) {
            convertor.projectiveToReal(depth_dev, tmpPoints_Device);
            HIP_OK(hipMemcpy(cloud.data(), tmpPoints_Device, N * sizeof(float3), hipMemcpyDeviceToHost));
            int n = 0;
            const float e = mean_error(cloud, truth, &n);
            std::printf("%s %.6f %d\n", name, e, n);
            if (save) {
                HIP_OK(hipMemcpy(out.data(), depth_dev, N * sizeof(float), hipMemcpyDeviceToHost));
                write_file(prefix + name + ".f32", out.data(), N);
            }
        };

        convertor.projectiveToReal(inputDepth_Device, inputPoints_Device);       // main.cpp:168
        convertor.projectiveToReal(bufferDepth_Device, bufferPoints_Device);     // :175
        HIP_OK(hipMemcpy(truth.data(), bufferPoints_Device, N * sizeof(float3), hipMemcpyDeviceToHost));
        report("input", inputDepth_Device, false);
        JBF.Process(inputDepth_Device, Color_Device);                            // :179
        report("jbf", JBF.getFiltered_Device(), true);                           // :182-183
        MRF.Process(inputDepth_Device, Color_Device);                            // :186
        report("mrf", MRF.getFiltered_Device(), true);
        RGBF.Process(inputDepth_Device, inputPoints_Device, Color_Device);       // :193
        report("rgbf", RGBF.getRefinedDepth_Device(), true);                     // :196-197
        HIP_OK(hipDeviceSynchronize());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
