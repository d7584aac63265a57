/*
 * include/kde_test_hooks.h — entry points of tools/hooks/libkde_hooks.so and libkde_hip_stage.so.
 *
 * NOT part of the product ABI (include/kde_hip.h / libkde_hip.so export none of these): a measurement helper for
 * bench.py and a probe that lets a test call a device function of the product kernels on chosen arguments.
 */
#ifndef KDE_TEST_HOOKS_H
#define KDE_TEST_HOOKS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* float4 streaming copy (one 16-byte element per thread, the loop shape tools/hbm_microbench found fastest):
 * the empirical HBM ceiling bench.py quotes next to the 8 TB/s datasheet figure.  0 = ok. */
int kde_bench_copy(const void* src_dev, void* dst_dev, size_t bytes, void* stream);

/* packed-BGR copy of npix pixels with K0's access pattern (one unaligned dword load per pixel, 16-bit stores): run under the
 * same PMC passes as K0 it calibrates FETCH_SIZE / WRITE_SIZE for that pattern (exactly 3 bytes read and written per
 * pixel), which the 2 x FETCH_SIZE rule of 16-byte streams does not cover.  0 = ok. */
int kde_bench_bgr3_copy(const void* src_dev, void* dst_dev, size_t npix, void* stream);

/* out_dev[i] = the kernels' square root (csrc/kde_device_math.h, sqrt_int24) of the integer first + i
 * (i < n; first + n <= 2^24), so that the tests can prove it equal to sqrtf() on every argument calculateLD can
 * form (DepthAdaptiveSuperpixel.cu:213).  0 = ok. */
int kde_test_sqrt_int24(uint32_t first, uint32_t n, float* out_dev, void* stream);

/* out_dev[i] = fastdiv24(xs_dev[i], make_fastdiv24(d, max_dividend)) evaluated ON THE DEVICE with the very functions of
 * csrc/kde_device_math.h (the division-by-multiplication of K1's tile map and K7's grid-cell arithmetic), so that a test
 * can compare it with x / d -- including the fallback path (ok == 0: d or max_dividend >= 2^24).  m_sh_ok (host, 3 words,
 * may be NULL) receives the magic number, the shift and the ok flag make_fastdiv24 chose.  0 = ok. */
int kde_test_fastdiv24(uint32_t d, uint64_t max_dividend, uint32_t n, const uint32_t* xs_dev, uint32_t* out_dev,
                       uint32_t* m_sh_ok, void* stream);

/* ---- tools/hooks/libkde_hip_stage.so ------------------------------------------------------------------------------
 * The product library's own sources compiled with -DKDE_STAGE_HOOKS: every entry point of include/kde_hip.h (same
 * code, same flags) plus the one below.  It exists so that the parity tests can check K1 and K10 STAGE BY STAGE: the
 * composite filters are ill-conditioned only through their first-pass average (and K10's deviation), each stage by
 * itself is well-conditioned, so the tests take the GPU's own average / deviation, re-evaluate the last pass from
 * them in binary64 on the CPU and hold the GPU's final value to 1e-4 against that (oracle/oracle.py: stage_check_*).
 * A test first asserts that this library's final output is bit-identical to the product library's.
 *
 * kde_stage_set(jbf_avg, ers_avg, ers_dev, counters, force_full_rules) — process-wide, applies to later launches:
 *   jbf_avg   device float[n*H*W] or NULL: K1's first-pass average exactly as its second pass uses it
 *             (JointBilateralFilter.cu:38-41; NaN where the sum of weights is 0 and the output is therefore 0)
 *   ers_avg   device float[H*W] or NULL: K10's label-restricted average (EdgeRefinedSuperpixel.cu:139-141)
 *   ers_dev   device float[H*W] or NULL: K10's mean absolute deviation (EdgeRefinedSuperpixel.cu:143-156); only
 *             meaningful where ers_avg is not NaN
 *   counters  device unsigned[8] or NULL, incremented once per tile: [0..3] K1 packed kernels by the body the tile ran
 *             (bit 0: colour rule compiled in, bit 1: depth rule), [4]/[5] K10 tiles with / without the "adaptive
 *             sigma provably small" shortcut, [6]/[7] K10 tiles without / with the depth rule in the colour-free rows
 *   force_full_rules  != 0: no tile-level elision — every tile runs the body with both Q1 rules (K1) / the per-pixel
 *             deviation pass and the depth rule (K10); the output must not change by a bit                          */
int kde_stage_set(float* jbf_avg_dev, float* ers_avg_dev, float* ers_dev_dev, unsigned* counters_dev, int force_full_rules);

#ifdef __cplusplus
}
#endif
#endif
