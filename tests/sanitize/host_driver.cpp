// tests/sanitize/host_driver.cpp — the host-side arithmetic of libkde_hip.so (csrc/kde_host_math.h: exp underflow
// threshold, spatial table, the "smallest value whose factor underflows" searches that implement the reference's
// Q1 rule) compiled WITHOUT HIP under -fsanitize=address,undefined and checked against the CPU oracle and against
// the defining properties.  Built and run by tests/test_sanitizers.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../kinectdepthmapenhancement_amd/csrc/kde_host_math.h"
#include "../../oracle/kde_oracle.h"

#define CHECK(c)                                                      \
    do {                                                              \
        if (!(c)) {                                                   \
            std::fprintf(stderr, "FAILED %s (line %d)\n", #c, __LINE__); \
            return 1;                                                 \
        }                                                             \
    } while (0)

int main()
{
    using namespace kde;
    // exp(-x) == 0 in binary32 exactly from the threshold on
    const float xz = exp_zero_threshold();
    CHECK(expf(-xz) == 0.0f);
    CHECK(expf(-std::nextafterf(xz, 0.0f)) != 0.0f);
    // the spatial table is the oracle's, bit for bit, for every window / sigma the ABI accepts
    for (int w = 1; w <= 31; w += 2)
        for (float sigma : {0.5f, 1.0f, 3.0f, 30.0f, 70.0f}) {
            std::vector<float> a((size_t)w * w), b((size_t)w * w);
            spatial_table(w, sigma, a.data());
            okde_spatial_table(w, sigma, b.data());
            for (size_t i = 0; i < a.size(); i++) CHECK(a[i] == b[i]);
        }
    // smallest q with q / den >= thr: the defining property on both sides of the answer, incl. degenerate dens
    for (float sigma : {0.5f, 4.0f, 20.0f, 70.0f, 1000.0f, 1.0e18f}) {
        const float den = 2.0f * (sigma * sigma);
        const float q = smallest_q_reaching(den, xz);
        if (std::isinf(q)) {
            CHECK(!(3.0e38f / den >= xz));
        } else {
            CHECK(q / den >= xz);
            CHECK(q == 0.0f || !(std::nextafterf(q, 0.0f) / den >= xz));
        }
    }
    CHECK(smallest_q_reaching(0.0f, xz) == 0.0f);
    CHECK(smallest_q_reaching(-1.0f, xz) == 0.0f);
    CHECK(smallest_q_reaching(NAN, xz) == 0.0f);
    // smallest integer colour distance whose factor underflows, against the literal float expression of the reference
    for (float sigma : {0.3f, 2.0f, 7.65f, 20.0f, 30.0f, 50.0f, 400.0f}) {
        const float den = 2 * (sigma * sigma);
        const int cd = smallest_cd_reaching(den, xz);
        CHECK(cd >= 0 && cd <= 195076);
        if (cd <= 195075) CHECK(expf(-(float)cd / den) == 0.0f);
        if (cd > 0) CHECK(expf(-(float)(cd - 1) / den) != 0.0f);
    }
    std::printf("host driver ok (exp_zero_threshold = %.9g)\n", (double)xz);
    return 0;
}
