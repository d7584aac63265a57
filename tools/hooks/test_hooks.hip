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

}  // namespace

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
