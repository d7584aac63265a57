/* tests/sanitize/oracle_driver.c — runs the CPU oracle's kernels on small, ragged and degenerate inputs.  Built by
 * tests/test_sanitizers.py with -fsanitize=address,undefined (CPU build only): any out-of-bounds access, signed
 * overflow, bad shift or misaligned access in the restatement aborts the run.  Prints a checksum so that the work
 * cannot be optimised away. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/kde_oracle.h"

static unsigned rng_state = 12345u;
static unsigned rnd(void) { return rng_state = rng_state * 1664525u + 1013904223u; }

static void fill(int w, int h, uint8_t* bgr, float* depth, int mode)
{
    for (int i = 0; i < w * h; i++) {
        const unsigned r = rnd();
        float z = 1000.0f + (float)(r % 2000u);
        if (mode == 1 && (r >> 12) % 4u == 0) z = 0.0f;                 /* holes */
        if (mode == 2) z = (r >> 12) % 2u ? 1000.0f : 1400.0f;          /* steps beyond the Q1 jump */
        if (mode == 3) z = (r >> 12) % 50u == 0 ? NAN : z;              /* NaN depth is "not > 50" */
        depth[i] = z;
        for (int c = 0; c < 3; c++) bgr[3 * i + c] = mode == 2 ? (uint8_t)(60 * ((r >> (8 + c)) & 1u)) : (uint8_t)(r >> (8 * c));
    }
}

int main(void)
{
    double acc = 0.0;
    const int sizes[][2] = {{1, 1}, {2, 3}, {7, 5}, {33, 9}, {64, 48}, {97, 51}};
    for (unsigned si = 0; si < sizeof(sizes) / sizeof(sizes[0]); si++) {
        const int w = sizes[si][0], h = sizes[si][1];
        const size_t n = (size_t)w * h;
        uint8_t* bgr = (uint8_t*)malloc(n * 3);
        uint8_t* sm = (uint8_t*)malloc(n * 3);
        float* depth = (float*)malloc(n * sizeof(float));
        float* out = (float*)malloc(n * sizeof(float));
        okde_env env;
        env.flags = (uint8_t*)malloc(n);
        env.lo = (double*)malloc(n * sizeof(double));
        env.hi = (double*)malloc(n * sizeof(double));
        for (int mode = 0; mode < 4; mode++) {
            fill(w, h, bgr, depth, mode);
            const int wins[] = {1, 3, 5, 11, 31};
            for (unsigned k = 0; k < 5; k++) {
                okde_jbf_process(w, h, depth, bgr, wins[k], k == 4 ? 0.5f : 3.0f, mode == 2 ? 7.65f : 50.0f, 20.0f, 5, 30.0f, 30.0f, sm, out, &env);
                for (size_t i = 0; i < n; i++) acc += (out[i] == out[i] ? out[i] : 0.0) + env.flags[i] + sm[3 * i];
            }
            okde_jbf_process(w, h, depth, bgr, 5, 70.0f, 0.0f, 0.0f, 13, 20.0f, 4.0f, sm, out, NULL);   /* sigmas off, big K0 */
            okde_mrf_kernel(w, h, depth, bgr, 5, 50.0f, 150.0f, out);
            for (size_t i = 0; i < n; i++) acc += out[i] == out[i] ? out[i] : 0.0;
            /* DimensionConvertor + Buffer2D */
            okde_float3* pts = (okde_float3*)malloc(n * sizeof(okde_float3));
            okde_float3* pts2 = (okde_float3*)malloc(n * sizeof(okde_float3));
            okde_p2r_depth(w, h, 575.8f, 575.8f, w / 2, h / 2, depth, pts);
            okde_r2p(w, h, 575.8f, 575.8f, w / 2, h / 2, pts, pts2);
            okde_p2r_points(w, h, 575.8f, 575.8f, w / 2, h / 2, pts2, pts);
            okde_p2r_interp(w, h, 575.8f, 575.8f, w / 2, h / 2, depth, pts2);
            okde_weighted_d* buf = (okde_weighted_d*)malloc(n * sizeof(okde_weighted_d));
            okde_buf_init((int)n, buf);
            for (int f = 0; f < 4; f++) okde_buf_update((int)n, buf, depth);
            depth[0] = 3.0e9f;                                           /* float -> int saturation in the gate */
            okde_buf_update((int)n, buf, depth);
            okde_buf_get_depth((int)n, buf, out);
            for (size_t i = 0; i < n; i++) acc += out[i] == out[i] ? out[i] : 0.0;
            /* superpixels + ERS on frames large enough for the geometry guard */
            if (w >= 64 && h >= 48) {
                const float K9[9] = {575.8f, 0, w / 2.0f, 0, 575.8f, h / 2.0f, 0, 0, 1};
                int32_t *sp = (int32_t*)malloc(n * 4), *da = (int32_t*)malloc(n * 4), *rl = (int32_t*)malloc(n * 4);
                fill(w, h, bgr, depth, mode == 3 ? 1 : mode);
                okde_p2r_depth(w, h, 575.8f, 575.8f, w / 2, h / 2, depth, pts);
                okde_ers_set_env_sink(&env);
                if (okde_rgbf_process(w, h, 4, 5, K9, depth, pts, bgr, sp, da, rl, out) == 0)
                    for (size_t i = 0; i < n; i++) acc += (out[i] == out[i] ? out[i] : 0.0) + rl[i] + env.flags[i];
                okde_ers_set_env_sink(NULL);
                const double K9d[9] = {575.8, 0, w / 2.0, 0, 575.8, h / 2.0, 0, 0, 1};
                if (okde_spdsr_head(w, h, 4, 5, K9d, depth, pts, bgr, rl, out, pts2) == 0) {
                    float nd[20 * 4];
                    okde_spdsr_cluster_planes(w, h, 20, rl, pts2, nd);
                    okde_projection_plane(w, h, 575.8f, 575.8f, w / 2, h / 2, nd, 20, rl, pts2, pts, pts2, 3);
                    for (int i = 0; i < 80; i++) acc += nd[i] == nd[i] ? nd[i] : 0.0;
                }
                free(sp); free(da); free(rl);
            }
            free(pts); free(pts2); free(buf);
        }
        free(bgr); free(sm); free(depth); free(out); free(env.flags); free(env.lo); free(env.hi);
    }
    printf("oracle driver ok %.6e\n", acc);
    return 0;
}
