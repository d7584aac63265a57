// kde/viewers.hpp — host-side renderers behind the reference classes' viewer members, so that a source that calls them
// still compiles and gets the same pictures:
//   JointBilateralFilter::visualize / MarkovRandomField::visualize   (JointBilateralFilter.cpp:50-79, MarkovRandomField.cpp:50-79)
//   EdgeRefinedSuperpixel::getSegmentedImage(int) / (image) / getRandomColorImage  (EdgeRefinedSuperpixel.cpp:70-147)
//   SuperpixelSegmentation::getSegmentedImage(image, options) / getRandomColorImage / releaseVideo
//                                                                    (SuperpixelSegmentation.cpp:50-200)
// Visualisation is outside the hot path: nothing here touches the GPU, opens a window (the reference's
// cv::imshow / cv::waitKey) or writes a video.  The images are object-owned host buffers of packed BGR bytes; as<M>()
// wraps one in a cv::Mat_<cv::Vec3b>-like type (anything constructible from rows, cols, pixel pointer, step).
#ifndef KDE_VIEWERS_HPP
#define KDE_VIEWERS_HPP

#include <cstddef>
#include <cstdint>
#include <vector>

namespace kde {

struct HostImage8UC3 {
    std::vector<uint8_t> px;
    int rows = 0, cols = 0;
    HostImage8UC3() = default;
    HostImage8UC3(int r, int c) : px(static_cast<size_t>(r) * c * 3, 0), rows(r), cols(c) {}
    uint8_t* data() { return px.data(); }
    const uint8_t* data() const { return px.data(); }
    size_t step() const { return static_cast<size_t>(cols) * 3; }
    uint8_t* at(int y, int x) { return px.data() + (static_cast<size_t>(y) * cols + x) * 3; }
    const uint8_t* at(int y, int x) const { return px.data() + (static_cast<size_t>(y) * cols + x) * 3; }
    void set(int y, int x, uint8_t b, uint8_t g, uint8_t r)
    {
        uint8_t* p = at(y, x);
        p[0] = b;
        p[1] = g;
        p[2] = r;
    }
    template <class MatLike>
    MatLike as()    // a view, valid while this image lives: cv::Mat_<cv::Vec3b>(rows, cols, (cv::Vec3b*)data, step)
    {
        return MatLike(rows, cols, reinterpret_cast<typename MatLike::value_type*>(px.data()), step());
    }
};

namespace viewers {

// The reference's depth ramp (getRGB, JointBilateralFilter.cpp:80-92): the ratio is capped at 0.99 and split into three
// bands of 0.33 -- channel 0 rises, then channel 0 falls while channel 1 rises, then channel 1 falls while channel 2
// rises; the conversions truncate.  Channels a band does not name stay 0 (cv::Vec3b is zero-initialised).
inline void depth_ramp(float ratio, uint8_t out[3])
{
    static const float edge[4] = {0.0f, 0.33f, 0.66f, 0.99f};       // the reference's literals (2 * 0.33f is not 0.66f in general)
    const float t = ratio >= edge[3] ? edge[3] : ratio;
    out[0] = out[1] = out[2] = 0;
    const int k = t < edge[1] ? 0 : (t < edge[2] ? 1 : 2);
    if (k > 0) out[k - 1] = static_cast<unsigned char>((edge[k + 1] - t) / edge[1] * 255.0f);
    out[k] = static_cast<unsigned char>((t - edge[k]) / edge[1] * 255.0f);
}

// depth in mm -> ramp of depth / full_scale; pixels not above `valid_above` are black when `mask_invalid`
inline void render_depth(const float* depth, float full_scale, bool mask_invalid, float valid_above, HostImage8UC3& img)
{
    for (int y = 0; y < img.rows; y++)
        for (int x = 0; x < img.cols; x++) {
            const float d = depth[static_cast<size_t>(y) * img.cols + x];
            uint8_t c[3] = {0, 0, 0};
            if (!mask_invalid || d > valid_above) depth_ramp(d / full_scale, c);
            img.set(y, x, c[0], c[1], c[2]);
        }
}

// white where a pixel's label differs from the pixel below or to the right; the last row and column are left alone
// (the reference's loops stop at Height-1 / Width-1)
inline void mark_label_borders(const int32_t* labels, HostImage8UC3& img)
{
    const int W = img.cols, H = img.rows;
    for (int y = 0; y + 1 < H; y++)
        for (int x = 0; x + 1 < W; x++) {
            const int32_t l = labels[static_cast<size_t>(y) * W + x];
            if (l != labels[static_cast<size_t>(y + 1) * W + x] || l != labels[static_cast<size_t>(y) * W + x + 1])
                img.set(y, x, 255, 255, 255);
        }
}

// one fixed pseudo-random colour per label id (the reference draws rand() % 255 per id in its constructors; the
// sequence is the C library's, so only "one stable colour per id, channels in 0..254" is kept); label -1 is black
inline void render_random_colours(const int32_t* labels, HostImage8UC3& img)
{
    for (int y = 0; y < img.rows; y++)
        for (int x = 0; x < img.cols; x++) {
            const int32_t id = labels[static_cast<size_t>(y) * img.cols + x];
            if (id == -1) {
                img.set(y, x, 0, 0, 0);
                continue;
            }
            uint32_t h = static_cast<uint32_t>(id) * 2654435761u + 0x9E3779B9u;
            h ^= h >> 15;
            h *= 2246822519u;
            h ^= h >> 13;
            img.set(y, x, static_cast<uint8_t>(h % 255u), static_cast<uint8_t>((h >> 8) % 255u), static_cast<uint8_t>((h >> 16) % 255u));
        }
}

template <class ImageLike>   // anything with .rows / .cols / .data / .step holding packed 8UC3 on the host
inline void copy_from(const ImageLike& src, HostImage8UC3& dst)
{
    for (int y = 0; y < dst.rows; y++) {
        const uint8_t* row = reinterpret_cast<const uint8_t*>(src.data) + static_cast<size_t>(y) * static_cast<size_t>(src.step);
        for (int i = 0; i < dst.cols * 3; i++) dst.px[static_cast<size_t>(y) * dst.cols * 3 + i] = row[i];
    }
}

}  // namespace viewers
}  // namespace kde

#endif  // KDE_VIEWERS_HPP
