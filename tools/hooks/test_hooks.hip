// tools/hooks/test_hooks.hip -> tools/hooks/libkde_hooks.so (include/kde_test_hooks.h).  Test / measurement
// infrastructure only; the product library does not contain or export any of this.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/kde_test_hooks.h"
#include "../../kinectdepthmapenhancement_amd/csrc/kde_device_math.h"

namespace {

typedef float v4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_kernel(const v4* __restrict__ src, v4* __restrict__ dst, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void sqrt_int24_probe_kernel(uint32_t first, uint32_t n, float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = kde::sqrt_int24((float)(first + i));
}

__global__ __launch_bounds__(256) void fastdiv24_probe_kernel(kde::FastDiv24 f, uint32_t n, const uint32_t* __restrict__ xs,
                                                             uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = kde::fastdiv24(xs[i], f);
}

// packed-BGR copy with K0's own access pattern (calibration of FETCH_SIZE / WRITE_SIZE for it: the guide calibrates those
// counters for 16-byte-per-lane streams only): every thread reads its pixel pair as two unaligned dwords (3 payload
// bytes each, as K0 stages its tile) and writes it as three 16-bit stores (as K0 writes its output)
__global__ __launch_bounds__(256) void bgr3_copy_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t npix)
{
    const size_t pair = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t p0 = 2 * pair;
    if (p0 + 1 >= npix) return;                       // (the last pixel's dword read would leave the buffer: skipped)
    uint32_t a, b;
    __builtin_memcpy(&a, src + p0 * 3, 4);
    if (p0 + 2 < npix) __builtin_memcpy(&b, src + p0 * 3 + 3, 4);
    else b = (uint32_t)src[p0 * 3 + 3] | ((uint32_t)src[p0 * 3 + 4] << 8) | ((uint32_t)src[p0 * 3 + 5] << 16);
    a &= 0x00ffffffu;
    b &= 0x00ffffffu;
    uint16_t* oh = reinterpret_cast<uint16_t*>(dst + p0 * 3);
    oh[0] = (uint16_t)(a & 0xffffu);
    oh[1] = (uint16_t)((a >> 16) | ((b & 0xffu) << 8));
    oh[2] = (uint16_t)(b >> 8);
}

}  // namespace

extern "C" int kde_bench_bgr3_copy(const void* src_dev, void* dst_dev, size_t npix, void* stream)
{
    if (!src_dev || !dst_dev || (reinterpret_cast<uintptr_t>(dst_dev) & 1u)) return 1;
    const size_t pairs = npix / 2;
    if (pairs == 0) return 0;
    hipLaunchKernelGGL(bgr3_copy_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint8_t*>(src_dev), reinterpret_cast<uint8_t*>(dst_dev), npix);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

extern "C" int kde_test_fastdiv24(uint32_t d, uint64_t max_dividend, uint32_t n, const uint32_t* xs_dev, uint32_t* out_dev,
                                  uint32_t* m_sh_ok, void* stream)
{
    if (!xs_dev || !out_dev || d == 0) return 1;
    const kde::FastDiv24 f = kde::make_fastdiv24(d, max_dividend);      // the host function the launchers call
    if (m_sh_ok) {
        m_sh_ok[0] = f.m;
        m_sh_ok[1] = f.sh;
        m_sh_ok[2] = f.ok;
    }
    if (n == 0) return 0;
    hipLaunchKernelGGL(fastdiv24_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), f, n, xs_dev, out_dev);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

extern "C" int kde_bench_copy(const void* src_dev, void* dst_dev, size_t bytes, void* stream)
{
    if (!src_dev || !dst_dev || bytes % 16 != 0 || (reinterpret_cast<uintptr_t>(src_dev) & 15u) || (reinterpret_cast<uintptr_t>(dst_dev) & 15u))
        return 1;
    const size_t n4 = bytes / 16;
    if (n4 == 0) return 0;
    if ((n4 + 255) / 256 > 0x7fffffffull) return 1;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const v4*>(src_dev), reinterpret_cast<v4*>(dst_dev), n4);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}

extern "C" int kde_test_sqrt_int24(uint32_t first, uint32_t n, float* out_dev, void* stream)
{
    if (!out_dev || (uint64_t)first + n > (1ull << 24)) return 1;
    if (n == 0) return 0;
    hipLaunchKernelGGL(sqrt_int24_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), first, n, out_dev);
    return hipGetLastError() == hipSuccess ? 0 : 4;
}
