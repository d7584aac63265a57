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
#include <cstring>
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

// ---- the instruction kinds: X(id, label, instructions per group, code).  Eight independent chains per register class
// (a: float, p: float pair, u: uint32, d: double) so that no instruction waits for the previous one.  A label that is a
// single mnemonic is a pure kind: tools/valu_costs.py turns those rows into the cost table (profiles/valu_costs.json)
// that tools/isa_slots.py and tools/pmc_report.py price a kernel's instruction mix with.
#define J ((i + 1) & 7)
#define K2 ((i + 2) & 7)
#define KINDS(X)                                                                                                                  \
    X(FMA, "v_fma_f32", 1, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));)                            \
    X(FMAC, "v_fmac_f32", 1, asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));)                         \
    X(MUL_F32, "v_mul_f32", 1, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));)                                    \
    X(ADD_F32, "v_add_f32", 1, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));)                                    \
    X(SUB_F32, "v_sub_f32", 1, asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));)                                    \
    X(MAX_F32, "v_max_f32", 1, asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));)                                    \
    X(MIN_F32, "v_min_f32", 1, asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));)                                    \
    X(MIN3_F32, "v_min3_f32", 1, asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));)                     \
    X(MED3_F32, "v_med3_f32", 1, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));)                     \
    X(FRACT, "v_fract_f32", 1, asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));)                                                \
    X(RNDNE, "v_rndne_f32", 1, asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));)                                                \
    X(MOV, "v_mov_b32", 1, asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[J]));)                                          \
    X(PK_MOV, "v_pk_mov_b32", 1, asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(p[i]) : "v"(p[J]), "v"(p[K2]));)                  \
    X(PK_FMA, "v_pk_fma_f32", 1, asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));)                 \
    X(PK_MUL, "v_pk_mul_f32", 1, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc0));)                              \
    X(PK_ADD, "v_pk_add_f32", 1, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc1));)                              \
    X(EXP, "v_exp_f32", 1, asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));)                                                      \
    X(RCP, "v_rcp_f32", 1, asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));)                                                      \
    X(RSQ, "v_rsq_f32", 1, asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));)                                                      \
    X(SQRT, "v_sqrt_f32", 1, asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));)                                                   \
    X(DOT4, "v_dot4_u32_u8", 1, asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[J]), "v"(u[K2]));)              \
    X(SAD_U8, "v_sad_u8", 1, asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[J]), "v"(u[K2]));)                      \
    X(PERM, "v_perm_b32", 1, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[J]), "v"(u[K2]));)                    \
    X(BFE, "v_bfe_u32", 1, asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u[i]));)                                                \
    X(LSHL_ADD, "v_lshl_add_u32", 1, asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[J]));)                    \
    X(ADD_LSHL, "v_add_lshl_u32", 1, asm volatile("v_add_lshl_u32 %0, %0, %1, 1" : "+v"(u[i]) : "v"(u[J]));)                    \
    X(LSHL_OR, "v_lshl_or_b32", 1, asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(u[J]));)                       \
    X(ADD3, "v_add3_u32", 1, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[J]), "v"(u[K2]));)                    \
    X(LSHL_ADD_U64, "v_lshl_add_u64", 1, asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(q[i & 3]) : "v"(q[(i + 1) & 3]));)    \
    X(MAD_U64_U32, "v_mad_u64_u32", 1, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i & 3]) : "v"(u[i]), "v"(u[J]) : "vcc");) \
    X(ADD_U32, "v_add_u32", 1, asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                                  \
    X(SUB_U32, "v_sub_u32", 1, asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                                  \
    X(AND_B32, "v_and_b32", 1, asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                                  \
    X(OR_B32, "v_or_b32", 1, asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                                     \
    X(XOR_B32, "v_xor_b32", 1, asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                                  \
    X(LSHLREV, "v_lshlrev_b32", 1, asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));)                                       \
    X(LSHRREV, "v_lshrrev_b32", 1, asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(u[i]));)                                       \
    X(ASHRREV, "v_ashrrev_i32", 1, asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(u[i]));)                                       \
    X(MIN_U32, "v_min_u32", 1, asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                                  \
    X(MAX_U32, "v_max_u32", 1, asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                                  \
    X(MAX_I32, "v_max_i32", 1, asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                                  \
    X(MUL_U24, "v_mul_u32_u24", 1, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                          \
    X(MAD_U24, "v_mad_u32_u24", 1, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[J]), "v"(u[K2]));)           \
    X(MUL_LO, "v_mul_lo_u32", 1, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                             \
    X(MUL_HI, "v_mul_hi_u32", 1, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                             \
    X(SDWA_ADD, "v_add_u32_sdwa", 1, asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(u[i]) : "v"(u[J]));) \
    X(SDWA_MIN, "v_min_u32_sdwa", 1, asm volatile("v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(u[i]) : "v"(u[J]));) \
    X(DPP_MOV, "v_mov_b32_dpp", 1, asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[J]));) \
    X(CVT_I32, "v_cvt_i32_f32", 1, asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));)                              \
    X(CVT_U32, "v_cvt_u32_f32", 1, asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));)                              \
    X(CVT_F32_U32, "v_cvt_f32_u32", 1, asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));)                          \
    X(CVT_F32_I32, "v_cvt_f32_i32", 1, asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(u[i]));)                          \
    X(CVT_UBYTE, "v_cvt_f32_ubyte1", 1, asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(u[i]));)                      \
    X(CVT_PK_U8, "v_cvt_pk_u8_f32", 1, asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u[i]) : "v"(a[i]));)                 \
    X(CNDMASK_VCC, "v_cndmask_b32", 1, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c0));)                   \
    X(CNDMASK_SGPR, "v_cndmask_b32_e64", 1, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(c0));)      \
    X(CMP_E32, "v_cmp_lt_f32_e32", 1, asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");)             \
    X(CMP_E64, "v_cmp_ge_f32_e64", 1, asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m64) : "v"(a[i]), "v"(c0));)             \
    X(CMP_I32, "v_cmp_gt_i32_e32", 1, asm volatile("v_cmp_gt_i32_e32 vcc, %0, %1" : : "v"(u[i]), "v"(u[J]) : "vcc");)           \
    X(CMP_CLASS, "v_cmp_class_f32_e32", 1, asm volatile("v_cmp_class_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(u[J]) : "vcc");)    \
    X(CMP_CNDMASK, "v_cmp_lt_f32 + v_cndmask_b32 (per instruction)", 2,                                                           \
      asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");                                               \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[J]) : "v"(c0) : "vcc");)                                             \
    X(CMP_2CND, "1 v_cmp_lt_f32 (vcc) + 2 v_cndmask_b32 (vcc)", 3,                                                                 \
      asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");                                               \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[J]) : "v"(c0) : "vcc");                                              \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[K2]) : "v"(c1) : "vcc");)                                            \
    X(CMP_4CND, "1 v_cmp_lt_f32 (vcc) + 4 v_cndmask_b32 (vcc)", 5,                                                                 \
      asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");                                               \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[J]) : "v"(c0) : "vcc");                                              \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[K2]) : "v"(c1) : "vcc");                                             \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[(i + 3) & 7]) : "v"(c0) : "vcc");                                    \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[(i + 4) & 7]) : "v"(c1) : "vcc");)                                   \
    X(CMP64_2CND, "1 v_cmp_lt_f32_e64 (sgpr pair) + 2 v_cndmask_b32_e64 (same pair)", 3,                                            \
      asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" : : "v"(a[i]), "v"(c0) : "s10", "s11");                                   \
      asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[J]) : "v"(c0) : "s10", "s11");                              \
      asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[K2]) : "v"(c1) : "s10", "s11");)                            \
    X(CMP_MUL_CND, "1 v_cmp_lt_f32 (vcc) + 2 v_mul_f32 + 1 v_cndmask_b32 (vcc)", 4,                                                 \
      asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");                                               \
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[J]) : "v"(c0));                                                               \
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[K2]) : "v"(c0));                                                              \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[(i + 3) & 7]) : "v"(c0) : "vcc");)                                   \
    X(MIX_2C_RUN, "4 v_mul_f32 + 1 v_pk_fma_f32", 5,                                                                               \
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));                                                               \
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[J]) : "v"(c0));                                                               \
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[K2]) : "v"(c0));                                                              \
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[(i + 3) & 7]) : "v"(c0));                                                     \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));)                                            \
    X(CMP_CND_MUL_CND, "1 v_cmp_lt_f32 (vcc) + v_cndmask (vcc) + v_mul_f32 + v_cndmask (vcc)", 4,                                  \
      asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");                                               \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[J]) : "v"(c0) : "vcc");                                              \
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[K2]) : "v"(c0));                                                              \
      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[(i + 3) & 7]) : "v"(c1) : "vcc");)                                   \
    X(CMP_2CND_VOP3, "1 v_cmp_lt_f32 (vcc) + 2 v_cndmask_b32_e64 (vcc as explicit operand)", 3,                                     \
      asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");                                               \
      asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[J]) : "v"(c0) : "vcc");                                          \
      asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[K2]) : "v"(c1) : "vcc");)                                        \
    X(CMP_2CND_DIFFSRC, "1 v_cmp_lt_f32 (vcc) + 2 v_cndmask_b32 (vcc), distinct dst and sources", 3,                                \
      asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");                                               \
      asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[J]) : "v"(c0), "v"(c1) : "vcc");                                     \
      asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[K2]) : "v"(c1), "v"(c0) : "vcc");)                                   \
    X(ADDC_CHAIN, "1 v_add_co_u32 (vcc) + 1 v_addc_co_u32 (vcc)", 2,                                                                \
      asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(u[i]) : "v"(u[J]) : "vcc");                                             \
      asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[K2]) : "v"(u[(i + 3) & 7]) : "vcc");)                           \
    X(READLANE, "v_readlane_b32", 1, asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sx) : "v"(u[i]));)                           \
    X(READFIRST, "v_readfirstlane_b32", 1, asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sx) : "v"(u[i]));)                  \
    X(MBCNT, "v_mbcnt_lo_u32_b32", 1, asm volatile("v_mbcnt_lo_u32_b32 %0, -1, %0" : "+v"(u[i]));)                              \
    X(ACC_WRITE, "v_accvgpr_write_b32", 1, asm volatile("v_accvgpr_write_b32 a4, %0" : : "v"(u[i]) : "a4");)                \
    X(ACC_READ, "v_accvgpr_read_b32", 1, asm volatile("v_accvgpr_read_b32 %0, a4" : "=v"(u[i]) : : "a4");)                   \
    X(DIV_SCALE, "v_div_scale_f32", 1, asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(c0) : "vcc");)      \
    X(DIV_FMAS, "v_div_fmas_f32", 1, asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1) : "vcc");)     \
    X(DIV_FIXUP, "v_div_fixup_f32", 1, asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));)          \
    X(MOV_B64, "v_mov_b64", 1, asm volatile("v_mov_b64 %0, %1" : "=v"(q[i & 3]) : "v"(q[(i + 1) & 3]));)                        \
    X(SUBREV_U32, "v_subrev_u32", 1, asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                         \
    X(NOT_B32, "v_not_b32", 1, asm volatile("v_not_b32 %0, %0" : "+v"(u[i]));)                                                  \
    X(ADDC_CO, "v_addc_co_u32", 1, asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(u[J]) : "vcc");)         \
    X(MAD_I64_I32, "v_mad_i64_i32", 1, asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(q[i & 3]) : "v"(u[i]), "v"(u[J]) : "vcc");) \
    X(MUL_I24, "v_mul_i32_i24", 1, asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));)                          \
    X(WRITELANE, "v_writelane_b32", 1, asm volatile("v_writelane_b32 %0, s10, 3" : "+v"(u[i]));)                                \
    X(RCP_IFLAG, "v_rcp_iflag_f32", 1, asm volatile("v_rcp_iflag_f32 %0, %0" : "+v"(a[i]));)                                    \
    X(MAX3_F32, "v_max3_f32", 1, asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));)                     \
    X(LDEXP_F32, "v_ldexp_f32", 1, asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(u[J]));)                            \
    X(LSHRREV_B16, "v_lshrrev_b16", 1, asm volatile("v_lshrrev_b16 %0, 1, %0" : "+v"(u[i]));)                                   \
    X(CVT_F64_F32, "v_cvt_f64_f32", 1, asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i & 3]) : "v"(a[i]));)                      \
    X(FMAC_F64, "v_fmac_f64", 1, asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(d[i & 3]) : "v"(dc0), "v"(dc1));)              \
    X(RCP_F64, "v_rcp_f64", 1, asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i & 3]));)                                              \
    X(FMA_F64, "v_fma_f64", 1, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i & 3]) : "v"(dc0), "v"(dc1));)                  \
    X(MUL_F64, "v_mul_f64", 1, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i & 3]) : "v"(dc0));)                               \
    X(ADD_F64, "v_add_f64", 1, asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 3]) : "v"(dc1));)                               \
    X(MIX_EXP_4PK, "1 v_exp_f32 + 4 v_pk_fma_f32", 5,                                                                             \
      asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                                                             \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));                                             \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[J]) : "v"(pc0), "v"(pc1));                                             \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[K2]) : "v"(pc0), "v"(pc1));                                            \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 3) & 7]) : "v"(pc0), "v"(pc1));)                                  \
    X(MIX_EXP_2PK, "1 v_exp_f32 + 2 v_pk_fma_f32", 3,                                                                             \
      asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                                                             \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));                                             \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[J]) : "v"(pc0), "v"(pc1));)                                            \
    X(MIX_EXP_1PK, "1 v_exp_f32 + 1 v_pk_fma_f32", 2,                                                                             \
      asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                                                             \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));)                                            \
    X(MIX_FMA_PK, "1 v_fma_f32 + 1 v_pk_fma_f32", 2,                                                                              \
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));                                                  \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));)                                            \
    X(MIX_ADDU_FMA, "1 v_add_u32 + 1 v_fma_f32", 2,                                                                               \
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[J]));                                                             \
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));)                                                 \
    X(MIX_DOT_LSHL_PK, "1 v_dot4 + 1 v_lshl_add + 1 v_pk_fma", 3,                                                                 \
      asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[J]), "v"(u[K2]));                                         \
      asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[(i + 3) & 7]) : "v"(u[(i + 4) & 7]));                                 \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc0), "v"(pc1));)                                            \
    X(MIX_K1_PASS1, "K1 pass-1 unit: 2 dot4 + 2 lshl_add + 2 exp + 6 pk", 12,                                                     \
      asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[J]), "v"(u[K2]));                                         \
      asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[(i + 3) & 7]) : "v"(u[J]), "v"(u[K2]));                               \
      asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[(i + 4) & 7]) : "v"(u[(i + 5) & 7]));                                 \
      asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[(i + 6) & 7]) : "v"(u[(i + 5) & 7]));                                 \
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc1));                                                           \
      asm volatile("v_pk_add_f32 %0, %0, %1 clamp" : "+v"(p[J]) : "v"(pc1));                                                     \
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[K2]) : "v"(pc0));                                                          \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 3) & 7]) : "v"(pc0), "v"(pc1));                                   \
      asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));                                                                             \
      asm volatile("v_exp_f32 %0, %0" : "+v"(a[J]));                                                                             \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 4) & 7]) : "v"(pc0), "v"(pc1));                                   \
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i + 5) & 7]) : "v"(pc0), "v"(pc1));)

#define X(id, label, grp, ...) id,
enum Kind { KINDS(X) NKINDS };
#undef X
#define X(id, label, grp, ...) label,
const char* kNames[NKINDS] = {KINDS(X)};
#undef X
#define X(id, label, grp, ...) grp,
const int kPerGroup[NKINDS] = {KINDS(X)};
#undef X

template <int KIND>
__global__ void bench(uint64_t* ticks, float* sink, float seed, int reps)
{
    float a[8];
    f2 p[8];
    uint32_t u[8];
    uint64_t q[4];
    double d[4];
    uint64_t m64 = 0;
    uint32_t sx = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = seed + (float)i * 0.001f + (float)threadIdx.x * 1e-6f;
        p[i] = f2{a[i], a[i] + 0.5f};
        u[i] = (uint32_t)threadIdx.x * 2654435761u + i;
        q[i & 3] = (uint64_t)u[i] * 0x9E3779B97F4A7C15ull;
        d[i & 3] = (double)a[i];
    }
    const float c0 = seed * 0.999f, c1 = seed * 1e-3f;
    const f2 pc0 = f2{c0, c0}, pc1 = f2{c1, c1};
    const double dc0 = (double)c0, dc1 = (double)c1;
    if constexpr (KIND == CNDMASK_VCC) asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");     // a defined condition mask
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
            const int i = k & 7;
#define X(id, label, grp, ...) \
    if constexpr (KIND == id) { __VA_ARGS__ }
            KINDS(X)
#undef X
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float s = (float)(m64 & 1) + (float)sx;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)u[i] + (float)q[i & 3] + (float)d[i & 3];
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

int main(int argc, char** argv)
{
    const char* only = argc > 1 ? argv[1] : nullptr;      // optional: only the kinds whose label contains this text
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
#define X(id, label, grp, ...) \
    if ((only == nullptr || std::strstr(label, only)) && run_kind<id>(cus, d_ticks, d_sink, h)) return 1;
    KINDS(X)
#undef X
    return 0;
}
