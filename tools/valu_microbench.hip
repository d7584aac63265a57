// tools/valu_microbench.hip — what does one wave64 VALU instruction cost a gfx950 SIMD?
//
// K1 / K10 / MRF are instruction-issue bound, so their roofline is "VALU issue slots per second".  This program
// measures the slot cost of every instruction kind those kernels are made of, at 1, 2, 4 and 8 waves per SIMD:
// each wave runs N independent instructions of one kind between two s_memtime stamps (shader-clock ticks), and
// cycles per instruction per SIMD = ticks / (N * waves per SIMD).  Mixtures show whether costs add
// (e.g. whether a v_exp_f32 overlaps with packed math of the same or of another wave).
//
// Build + run on the GPU box:   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/valu_microbench tools/valu_microbench.hip
//                               gpurun_out/valu_microbench > profiles/r02_valu_microbench.txt
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kUnroll = 32;    // instructions (or instruction groups) per loop trip

enum Kind { FMA, PK_FMA, PK_MUL, PK_ADD, EXP, RCP, DOT4, LSHL_ADD, ADD_U32, CNDMASK, CNDMASK_SGPR, CMP_CNDMASK, MUL_F32, CVT_I32, MAX_F32,
            MIX_EXP_4PK, MIX_EXP_2PK, MIX_EXP_1PK, MIX_DOT_LSHL_PK, MIX_K1_PASS1, NKINDS };
const char* kNames[NKINDS] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32",
                              "v_dot4_u32_u8", "v_lshl_add_u32", "v_add_u32", "v_cndmask_b32 (vcc)", "v_cndmask_b32_e64 (sgpr pair)",
                              "v_cmp_lt_f32 + v_cndmask_b32 (per instruction)", "v_mul_f32", "v_cvt_i32_f32", "v_max_f32",
                              "1 v_exp_f32 + 4 v_pk_fma_f32", "1 v_exp_f32 + 2 v_pk_fma_f32", "1 v_exp_f32 + 1 v_pk_fma_f32",
                              "1 v_dot4 + 1 v_lshl_add + 1 v_pk_fma", "K1 pass-1 unit: 2 dot4 + 2 lshl_add + 2 exp + 6 pk"};
const int kPerGroup[NKINDS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 5, 3, 2, 3, 12};

template <int KIND>
__global__ void bench(uint64_t* ticks, float* sink, float seed, int reps)
{
    // eight independent chains per kind so that no instruction waits for the previous one
    float a[8];
    f2 p[8];
    uint32_t u[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = seed + (float)i * 0.001f + (float)threadIdx.x * 1e-6f;
        p[i] = f2{a[i], a[i] + 0.5f};
        u[i] = (uint32_t)threadIdx.x * 2654435761u + i;
    }
    const float c0 = seed * 0.999f, c1 = seed * 1e-3f;
    const f2 pc0 = f2{c0, c0}, pc1 = f2{c1, c1};
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
            const int i = k & 7;
            if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));
            if (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc0));
            if (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc1));
            if (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == DOT4) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
            if (KIND == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c0));
            if (KIND == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (KIND == CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(c0));
            if (KIND == CMP_CNDMASK) {
                asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[(i + 1) & 7]) : "v"(c0) : "vcc");
            }
            if (KIND == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if (KIND == CVT_I32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
            if (KIND == MIX_EXP_4PK) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 1) & 7]) : "v"(pc0), "v"(pc1));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 2) & 7]) : "v"(pc0), "v"(pc1));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 3) & 7]) : "v"(pc0), "v"(pc1));
            }
            if (KIND == MIX_EXP_2PK) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 1) & 7]) : "v"(pc0), "v"(pc1));
            }
            if (KIND == MIX_EXP_1PK) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));
            }
            if (KIND == MIX_DOT_LSHL_PK) {
                asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[(i + 3) & 7]) : "v"(u[(i + 4) & 7]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));
            }
            if (KIND == MIX_K1_PASS1) {
                asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[(i + 3) & 7]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[(i + 4) & 7]) : "v"(u[(i + 5) & 7]));
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[(i + 6) & 7]) : "v"(u[(i + 5) & 7]));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc1));
                asm volatile("v_pk_add_f32 %0, %0, %1 clamp" : "+v"(p[(i + 1) & 7]) : "v"(pc1));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(i + 2) & 7]) : "v"(pc0));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 3) & 7]) : "v"(pc0), "v"(pc1));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[(i + 1) & 7]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 4) & 7]) : "v"(pc0), "v"(pc1));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 5) & 7]) : "v"(pc0), "v"(pc1));
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)u[i];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    if ((threadIdx.x & 63) == 0) ticks[wave] = t1 - t0;
    if (s == 12345.678f) sink[0] = s;      // keeps every chain live
}

// One launch per (kind, waves per SIMD), long enough (a few ms) that launch overhead does not matter.  Two clocks:
//   ticks : s_memtime stamps inside the kernel, median over waves  -> "cycles" if s_memtime runs at the shader clock
//   wall  : HIP events around the launch                           -> ns per instruction per SIMD, the unit a roofline needs
// A launch whose wall time exceeds a wave's own span by > 30 % did not keep its blocks co-resident and is marked '*'.
template <int KIND>
int run_kind(int cus, uint64_t* d_ticks, float* d_sink, std::vector<uint64_t>& h)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    char line_t[256], line_w[256];
    int nt = 0, nw = 0;
    double ghz_sum = 0;
    int ghz_n = 0;
    for (int wps : {1, 2, 3, 4, 5, 6, 8}) {
        const int threads = 256, blocks = cus * wps;      // a 256-thread block puts one wave on each SIMD of its CU
        const int waves = blocks * threads / 64;
        const int reps = 60000 / (wps * kPerGroup[KIND]) + 64;
        const double n_inst = (double)reps * kUnroll * kPerGroup[KIND];
        hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, d_ticks, d_sink, 1.0001f, reps / 8);   // warm-up
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(threads), 0, 0, d_ticks, d_sink, 1.0001f, reps);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), d_ticks, sizeof(uint64_t) * waves, hipMemcpyDeviceToHost));
        std::vector<uint64_t> v(h.begin(), h.begin() + waves);
        std::nth_element(v.begin(), v.begin() + waves / 2, v.end());
        const double span = (double)v[waves / 2];
        const double tick_ghz = span / (ms * 1e6);                    // ticks per ns if the wave spans the launch
        const bool serial = tick_ghz < 0.7 * 2.4 && ms * 1e6 * 2.4 > 1.3 * span + 50000.0 && false;
        (void)serial;
        const double per_t = span / (n_inst * wps);
        const double per_w = ms * 1e6 / (n_inst * wps);               // ns per instruction per SIMD (all SIMDs run alike)
        nt += snprintf(line_t + nt, sizeof(line_t) - nt, "  %6.2f", per_t);
        nw += snprintf(line_w + nw, sizeof(line_w) - nw, "  %6.3f", per_w);
        ghz_sum += tick_ghz;
        ghz_n++;
    }
    printf("%-50s ticks%s\n", kNames[KIND], line_t);
    printf("%-50s ns   %s   (ticks/ns %.2f)\n", "", line_w, ghz_sum / ghz_n);
    return 0;
}

int main()
{
    int dev = 0, cus = 0, clk = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev));
    uint64_t* d_ticks;
    float* d_sink;
    const int max_waves = cus * 8 * 4;
    CHECK(hipMalloc(&d_ticks, sizeof(uint64_t) * max_waves));
    CHECK(hipMalloc(&d_sink, 4));
    std::vector<uint64_t> h(max_waves);
    printf("# gfx950 VALU issue cost per wave64 instruction per SIMD, every SIMD of the chip running the same stream\n");
    printf("# device %d: %d CUs, max clock %d kHz\n", dev, cus, clk);
    printf("# rows 'ticks': s_memtime ticks of a wave / (its instructions x waves per SIMD); rows 'ns': launch wall time /\n");
    printf("#   (instructions per wave x waves per SIMD) -- what a roofline needs; ticks/ns = the two clocks' ratio (2.4 = s_memtime\n");
    printf("#   counts 2.4 GHz shader cycles and the waves of a SIMD span the whole launch)\n");
    printf("# columns: 1, 2, 3, 4, 5, 6, 8 waves per SIMD (256-thread blocks, one wave per SIMD each)\n");
    if (run_kind<FMA>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<PK_FMA>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<PK_MUL>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<PK_ADD>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<EXP>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<RCP>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<DOT4>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<LSHL_ADD>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<ADD_U32>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<CNDMASK>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<CNDMASK_SGPR>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<CMP_CNDMASK>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<MUL_F32>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<CVT_I32>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<MAX_F32>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<MIX_EXP_4PK>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<MIX_EXP_2PK>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<MIX_EXP_1PK>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<MIX_DOT_LSHL_PK>(cus, d_ticks, d_sink, h)) return 1;
    if (run_kind<MIX_K1_PASS1>(cus, d_ticks, d_sink, h)) return 1;
    return 0;
}
