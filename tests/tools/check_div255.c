// Test tool (tests/test_oracle.py::test_division_free_quality_term). Exhaustive check: for every float x with |x| <= 26000, (int)(x / 255.0f) (IEEE division, truncation)
// equals the division-free form used by k_pdq_hash64.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static inline int fast_form(float x) {
    /* multiplier = the float just BELOW 1/255 (RN(1/255) = 0x1.010102p-8 lies above it): the product,
     * even after its own rounding, stays below the true quotient, so trunc() is never too large and a
     * single "remainder >= 255" correction suffices. */
    const float c = 0x1.0101p-8f;
    float ax = fabsf(x);
    float m = truncf(ax * c);            /* candidate: floor(ax/255) or one less */
    float r = fmaf(-255.0f, m, ax);      /* exact remainder ax - 255*m */
    if (r >= 255.0f) m += 1.0f;
    int d = (int)m;
    return x < 0 ? -d : d;
}
int main(int argc, char** argv) {
    uint32_t hi; float lim = 26000.0f; memcpy(&hi, &lim, 4);
    uint64_t bad = 0, n = 0;
    uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1;  /* 1 = every float (about 25 s) */
    for (uint32_t u = 0; u <= hi; u += stride) {
        float x; memcpy(&x, &u, 4);
        for (int s = 0; s < 2; ++s) {
            float xs = s ? -x : x;
            int ref = (int)(xs / 255.0f);
            int got = fast_form(xs);
            if (ref != got) { if (bad < 10) printf("MISMATCH x=%a ref=%d got=%d\n", xs, ref, got); bad++; }
            n++;
        }
    }
    printf("checked %llu floats, %llu mismatches\n", (unsigned long long)n, (unsigned long long)bad);
    return bad != 0;
}
