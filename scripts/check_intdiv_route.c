// check_intdiv_route.c -- exhaustive check of the claim behind centroid_step4 (morna_amd/csrc/devutil.hpp):
// for every divisor m = 2 .. 202 and EVERY float significand of t, with y = RN32(1 / m),
//     q = RN(t * y);  e = fma(q, m, -t);  o = fma(-e, y, q)      equals      RN32(t / m)
// (bitwise, including the sign of zero), at several exponents of t: mid-range, next to FLT_MIN where the
// quotient is normal, and next to FLT_MAX.  Quotients below FLT_MIN are counted separately: there the route may
// differ from the division only by returning a non-zero subnormal (which the kernel detects and redoes).
//
//   gcc -O2 -march=native -fopenmp -ffp-contract=off scripts/check_intdiv_route.c -o /tmp/check_intdiv -lm && /tmp/check_intdiv
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float as_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t as_u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char **argv)
{
    // "quick": a subset for the test suite (7 divisors, 3 exponents: 3.5e8 cases); no argument: everything
    const int quick = argc > 1;
    static const int all_exps[] = {127, 100, 150, 9, 8, 7, 3, 2, 1, 253, 254};   // biased exponents of t
    static const int quick_exps[] = {127, 2, 254};
    static const int quick_m[] = {2, 3, 7, 10, 127, 201, 202};
    const int *exps = quick ? quick_exps : all_exps;
    const unsigned n_exps = quick ? 3 : 11;
    const int n_m = quick ? 7 : 201;
    long long bad = 0, bad_sub = 0, n = 0, n_sub = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : bad, bad_sub, n, n_sub)
    for (int mi = 0; mi < n_m; mi++) {
        const int m = quick ? quick_m[mi] : 2 + mi;
        const float fm = (float)m, y = 1.0f / fm;
        for (unsigned ei = 0; ei < n_exps; ei++)
            for (uint32_t sig = 0; sig < (1u << 23); sig++)
                for (int neg = 0; neg < 2; neg++) {
                    const float t = as_f(((uint32_t)neg << 31) | ((uint32_t)exps[ei] << 23) | sig);
                    const volatile float want = t / fm;
                    const float q = t * y, e = fmaf(q, fm, -t), o = fmaf(-e, y, q);
                    const int sub = fabsf(want) < 1.17549435e-38f;
                    if (sub) {
                        n_sub++;
                        // a wrong answer must be a non-zero subnormal (the guard of the kernel)
                        if (as_u(o) != as_u(want) && !(o != 0.f && fabsf(o) < 1.17549435e-38f)) bad_sub++;
                    } else {
                        n++;
                        if (as_u(o) != as_u(want)) bad++;
                    }
                }
    }
    // zeros keep their sign
    for (int m = 2; m <= 202; m++)
        for (int neg = 0; neg < 2; neg++) {
            volatile float tz = neg ? -0.0f : 0.0f;   // volatile: no compile-time folding of the signed zeros
            const float t = tz, fm = (float)m, y = 1.0f / fm;
            const float q = t * y;
            volatile float e = fmaf(q, fm, -t);   // volatile: gcc otherwise folds the negation below into the first fma,
            const float o = fmaf(-e, y, q);       // which is not the same operation on signed zeros
            if (as_u(o) != as_u(t / fm)) bad++;
        }
    // subnormal t (all of them, both signs)
#pragma omp parallel for schedule(dynamic) reduction(+ : bad_sub, n_sub)
    for (int mi = 0; mi < n_m; mi++) {
        const int m = quick ? quick_m[mi] : 2 + mi;
        const float fm = (float)m, y = 1.0f / fm;
        for (uint32_t sig = 1; sig < (1u << 23); sig += quick ? 7 : 1) {
            const float t = as_f(sig);
            const volatile float want = t / fm;
            const float q = t * y, e = fmaf(q, fm, -t), o = fmaf(-e, y, q);
            n_sub++;
            if (as_u(o) != as_u(want) && !(o != 0.f && fabsf(o) < 1.17549435e-38f)) bad_sub++;
        }
    }
    printf("normal quotients: %lld checked, %lld differ\n", n, bad);
    printf("quotients below FLT_MIN: %lld checked, %lld differ WITHOUT being a non-zero subnormal\n", n_sub, bad_sub);
    (void)argv;
    return bad || bad_sub ? 1 : 0;
}
