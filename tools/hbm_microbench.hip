// tools/hbm_microbench.hip — what does a streaming copy reach on this MI355X, and with which loop shape?
// The library's HBM-bound kernels (projectiveToReal, realToProjective, Buffer2D::updateData) are priced against the
// best line of this table (profiles/r02_hbm_microbench.txt), not against the 8 TB/s datasheet figure alone.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hbm_microbench tools/hbm_microbench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

// grid-stride, U float4 per thread and trip, all loads issued before the stores
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_u(const v4* __restrict__ src, v4* __restrict__ dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4 r[U];
#pragma unroll
        for (int k = 0; k < U; k++) r[k] = NT ? __builtin_nontemporal_load(&src[i + k * stride]) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (NT) __builtin_nontemporal_store(r[k], &dst[i + k * stride]);
            else dst[i + k * stride] = r[k];
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

// a workgroup owns a contiguous chunk (U KiB x 4 per wave): consecutive trips of one wave touch consecutive memory
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_chunk(const v4* __restrict__ src, v4* __restrict__ dst, size_t n4)
{
    const size_t per_block = (size_t)256 * U;
    for (size_t base = (size_t)blockIdx.x * per_block; base < n4; base += (size_t)gridDim.x * per_block) {
        v4 r[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            const size_t i = base + (size_t)k * 256 + threadIdx.x;
            if (i < n4) r[k] = NT ? __builtin_nontemporal_load(&src[i]) : src[i];
        }
#pragma unroll
        for (int k = 0; k < U; k++) {
            const size_t i = base + (size_t)k * 256 + threadIdx.x;
            if (i < n4) {
                if (NT) __builtin_nontemporal_store(r[k], &dst[i]);
                else dst[i] = r[k];
            }
        }
    }
}

__global__ __launch_bounds__(256) void read_only(const v4* __restrict__ src, float* __restrict__ sink, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    v4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const v4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc += a + b + c + d;
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) sink[0] = acc.x;
}

__global__ __launch_bounds__(256) void write_only(v4* __restrict__ dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const v4 z = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = z;
}

template <typename F>
int timeit(const char* name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, 0));
        for (int k = 0; k < 5; k++) launch();
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 5 < best) best = ms / 5;
    }
    printf("%-44s %8.3f ms   %7.1f GB/s   %5.1f %% of 8 TB/s\n", name, best, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12 * 100);
    return 0;
}

int main(int argc, char** argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)1 << 30);     // 1 GiB each way: far beyond the 256 MB cache
    const size_t n4 = bytes / 16;
    v4 *a, *b;
    float* sink;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 0, bytes));
    printf("# streaming copy of %zu MiB (read + write = %zu MiB of traffic), 256-thread workgroups\n", bytes >> 20, bytes >> 19);
    for (int blocks : {2048, 4096, 8192, 16384}) {
        char nm[96];
#define RUN(label, K)                                                                        \
    snprintf(nm, sizeof(nm), "%s, %d blocks", label, blocks);                                \
    if (timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(256), 0, 0, a, b, n4); })) return 1;
        RUN("grid-stride 1 float4/trip", (copy_u<1, false>));
        RUN("grid-stride 4 float4/trip", (copy_u<4, false>));
        RUN("grid-stride 8 float4/trip", (copy_u<8, false>));
        RUN("grid-stride 4 float4/trip nontemporal", (copy_u<4, true>));
        RUN("chunked 4 float4/trip", (copy_chunk<4, false>));
        RUN("chunked 8 float4/trip", (copy_chunk<8, false>));
        RUN("chunked 4 float4/trip nontemporal", (copy_chunk<4, true>));
    }
    {
        const unsigned blocks = (unsigned)((n4 + 255) / 256);
        if (timeit("one float4 per thread (no loop)", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_u<1, false>), dim3(blocks), dim3(256), 0, 0, a, b, n4); })) return 1;
        if (timeit("4 float4 per thread (no loop), chunked", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_chunk<4, false>), dim3(blocks / 4), dim3(256), 0, 0, a, b, n4); })) return 1;
        if (timeit("4 float4 per thread (no loop), chunked nt", 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_chunk<4, true>), dim3(blocks / 4), dim3(256), 0, 0, a, b, n4); })) return 1;
    }
    if (timeit("read only, 4 float4/trip, 4096 blocks", 1.0 * bytes, [&] { hipLaunchKernelGGL(read_only, dim3(4096), dim3(256), 0, 0, a, sink, n4); })) return 1;
    if (timeit("write only, 4096 blocks", 1.0 * bytes, [&] { hipLaunchKernelGGL(write_only, dim3(4096), dim3(256), 0, 0, b, n4); })) return 1;
    if (timeit("hipMemcpyAsync device to device", 2.0 * bytes, [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); })) return 1;
    return 0;
}
