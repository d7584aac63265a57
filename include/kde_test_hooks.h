/*
 * include/kde_test_hooks.h — entry points of tools/hooks/libkde_hooks.so.
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

/* out_dev[i] = the kernels' square root (csrc/kde_device_math.h, sqrt_int24) of the integer first + i
 * (i < n; first + n <= 2^24), so that the tests can prove it equal to sqrtf() on every argument calculateLD can
 * form (DepthAdaptiveSuperpixel.cu:213).  0 = ok. */
int kde_test_sqrt_int24(uint32_t first, uint32_t n, float* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
