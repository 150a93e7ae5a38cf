// valu_rates.hip -- issue cost (cycles per wave64 instruction, ONE wave on its SIMD) of the instructions the
// two_means step is made of, measured with s_memtime around unrolled runs of independent and of dependent
// instructions.  Evidence for DESIGN.md's account of why two_means is bound by instruction issue, not by HBM.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 8
#define ITER 512

// 8 independent instructions per asm block (registers v[...] of 8 separate values), or 8 dependent ones
#define BENCH_KERNEL(NAME, DECL, BODY_INDEP, BODY_DEP, SINK)                                           \
    __global__ void NAME(uint64_t *out, float seed)                                                     \
    {                                                                                                   \
        DECL;                                                                                           \
        const uint64_t w0 = wall_clock64();                                                             \
        uint64_t t0 = __builtin_readcyclecounter();                                                     \
        for (int i = 0; i < ITER; i++) { BODY_INDEP; }                                                  \
        uint64_t t1 = __builtin_readcyclecounter();                                                     \
        for (int i = 0; i < ITER; i++) { BODY_DEP; }                                                    \
        uint64_t t2 = __builtin_readcyclecounter();                                                     \
        if (threadIdx.x == 0) {                                                                         \
            out[0] = t1 - t0;                                                                           \
            out[1] = t2 - t1;                                                                           \
            out[3] = wall_clock64() - w0;                                                               \
            out[4] = t2 - t0;                                                                           \
        }                                                                                               \
        SINK;                                                                                           \
    }

#define F8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7
#define D8 double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7
#define SINKF if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) out[2] = 1
#define SINKD if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 12345.0) out[2] = 1
#define SINKFD if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) == 12345.f) out[2] = 1

#define OP1F(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define OP1F_DEP(op) asm volatile(op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0" : "+v"(a0))
#define OP2F(op) asm volatile(op " %0, %0, %0\n" op " %1, %1, %1\n" op " %2, %2, %2\n" op " %3, %3, %3\n" op " %4, %4, %4\n" op " %5, %5, %5\n" op " %6, %6, %6\n" op " %7, %7, %7" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define OP2F_DEP(op) asm volatile(op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0" : "+v"(a0))
#define OP3F(op) asm volatile(op " %0, %0, %0, %0\n" op " %1, %1, %1, %1\n" op " %2, %2, %2, %2\n" op " %3, %3, %3, %3\n" op " %4, %4, %4, %4\n" op " %5, %5, %5, %5\n" op " %6, %6, %6, %6\n" op " %7, %7, %7, %7" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define OP3F_DEP(op) asm volatile(op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0" : "+v"(a0))
#define OP2D(op) asm volatile(op " %0, %0, %0\n" op " %1, %1, %1\n" op " %2, %2, %2\n" op " %3, %3, %3\n" op " %4, %4, %4\n" op " %5, %5, %5\n" op " %6, %6, %6\n" op " %7, %7, %7" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
#define OP2D_DEP(op) asm volatile(op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0\n" op " %0, %0, %0" : "+v"(d0))
#define OP3D(op) asm volatile(op " %0, %0, %0, %0\n" op " %1, %1, %1, %1\n" op " %2, %2, %2, %2\n" op " %3, %3, %3, %3\n" op " %4, %4, %4, %4\n" op " %5, %5, %5, %5\n" op " %6, %6, %6, %6\n" op " %7, %7, %7, %7" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
#define OP3D_DEP(op) asm volatile(op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0\n" op " %0, %0, %0, %0" : "+v"(d0))
#define OP1D(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" \
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
#define OP1D_DEP(op) asm volatile(op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0" : "+v"(d0))
// f32 -> f64 -> f32 round trips: 8 independent pairs (16 instructions), or one dependent chain of 8 pairs
#define CVT_PAIR asm volatile("v_cvt_f64_f32 %8, %0\nv_cvt_f64_f32 %9, %1\nv_cvt_f64_f32 %10, %2\nv_cvt_f64_f32 %11, %3\n"                 \
                              "v_cvt_f64_f32 %12, %4\nv_cvt_f64_f32 %13, %5\nv_cvt_f64_f32 %14, %6\nv_cvt_f64_f32 %15, %7\n"               \
                              "v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"                 \
                              "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15"                 \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1), \
                                "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
#define CVT_PAIR_DEP asm volatile("v_cvt_f64_f32 %1, %0\nv_cvt_f32_f64 %0, %1\nv_cvt_f64_f32 %1, %0\nv_cvt_f32_f64 %0, %1\n"               \
                                  "v_cvt_f64_f32 %1, %0\nv_cvt_f32_f64 %0, %1\nv_cvt_f64_f32 %1, %0\nv_cvt_f32_f64 %0, %1\n"               \
                                  "v_cvt_f64_f32 %1, %0\nv_cvt_f32_f64 %0, %1\nv_cvt_f64_f32 %1, %0\nv_cvt_f32_f64 %0, %1\n"               \
                                  "v_cvt_f64_f32 %1, %0\nv_cvt_f32_f64 %0, %1\nv_cvt_f64_f32 %1, %0\nv_cvt_f32_f64 %0, %1"                 \
                                  : "+v"(a0), "+v"(d0))
// the quotient route of centroid_step4 for 8 values: cvt, mul_f64 by a wave-uniform double, cvt back
#define QUOT asm volatile("v_cvt_f64_f32 %8, %0\nv_cvt_f64_f32 %9, %1\nv_cvt_f64_f32 %10, %2\nv_cvt_f64_f32 %11, %3\n"                     \
                          "v_mul_f64 %8, %8, %8\nv_mul_f64 %9, %9, %9\nv_mul_f64 %10, %10, %10\nv_mul_f64 %11, %11, %11\n"                 \
                          "v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\n"                     \
                          "v_cvt_f64_f32 %12, %4\nv_cvt_f64_f32 %13, %5\nv_cvt_f64_f32 %14, %6\nv_cvt_f64_f32 %15, %7\n"                   \
                          "v_mul_f64 %12, %12, %12\nv_mul_f64 %13, %13, %13\nv_mul_f64 %14, %14, %14\nv_mul_f64 %15, %15, %15\n"           \
                          "v_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15"                     \
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(d0), "+v"(d1),     \
                            "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
#define CLASS8 asm volatile("v_cmp_class_f32 vcc, %0, %0\nv_cmp_class_f32 vcc, %1, %1\nv_cmp_class_f32 vcc, %2, %2\nv_cmp_class_f32 vcc, %3, %3\n" \
                            "v_cmp_class_f32 vcc, %4, %4\nv_cmp_class_f32 vcc, %5, %5\nv_cmp_class_f32 vcc, %6, %6\nv_cmp_class_f32 vcc, %7, %7"  \
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc")
#define CLASS_OR8 asm volatile("v_cmp_class_f32 vcc, %0, %0\ns_or_b64 s[20:21], s[20:21], vcc\nv_cmp_class_f32 vcc, %1, %1\ns_or_b64 s[20:21], s[20:21], vcc\n" \
                               "v_cmp_class_f32 vcc, %2, %2\ns_or_b64 s[20:21], s[20:21], vcc\nv_cmp_class_f32 vcc, %3, %3\ns_or_b64 s[20:21], s[20:21], vcc"    \
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc", "s20", "s21")

BENCH_KERNEL(k_fma_f32, F8, OP3F("v_fma_f32"), OP3F_DEP("v_fma_f32"), SINKF)
BENCH_KERNEL(k_mul_f32, F8, OP2F("v_mul_f32"), OP2F_DEP("v_mul_f32"), SINKF)
BENCH_KERNEL(k_pk_fma_f32, D8, OP3D("v_pk_fma_f32"), OP3D_DEP("v_pk_fma_f32"), SINKD)
BENCH_KERNEL(k_pk_mul_f32, D8, OP2D("v_pk_mul_f32"), OP2D_DEP("v_pk_mul_f32"), SINKD)
BENCH_KERNEL(k_mul_f64, D8, OP2D("v_mul_f64"), OP2D_DEP("v_mul_f64"), SINKD)
BENCH_KERNEL(k_fma_f64, D8, OP3D("v_fma_f64"), OP3D_DEP("v_fma_f64"), SINKD)
BENCH_KERNEL(k_rcp_f64, D8, OP1D("v_rcp_f64"), OP1D_DEP("v_rcp_f64"), SINKD)
BENCH_KERNEL(k_sqrt_f32, F8, OP1F("v_sqrt_f32"), OP1F_DEP("v_sqrt_f32"), SINKF)
BENCH_KERNEL(k_rcp_f32, F8, OP1F("v_rcp_f32"), OP1F_DEP("v_rcp_f32"), SINKF)
BENCH_KERNEL(k_cvt_pair, F8; D8, CVT_PAIR, CVT_PAIR_DEP, SINKFD)
BENCH_KERNEL(k_quot_route, F8; D8, QUOT, QUOT, SINKFD)
BENCH_KERNEL(k_class, F8, CLASS8, CLASS_OR8, SINKF)

struct Case {
    const char *name;
    void (*fn)(uint64_t *, float);
    int n_indep, n_dep;   // instructions per loop iteration
};

int main()
{
    uint64_t *d;
    hipMalloc(&d, 64);
    Case cases[] = {
        {"v_fma_f32", k_fma_f32, 8, 8},       {"v_mul_f32", k_mul_f32, 8, 8},
        {"v_pk_fma_f32", k_pk_fma_f32, 8, 8}, {"v_pk_mul_f32", k_pk_mul_f32, 8, 8},
        {"v_mul_f64", k_mul_f64, 8, 8},       {"v_fma_f64", k_fma_f64, 8, 8},
        {"v_rcp_f64", k_rcp_f64, 8, 8},       {"v_sqrt_f32", k_sqrt_f32, 8, 8},
        {"v_rcp_f32", k_rcp_f32, 8, 8},       {"cvt f32->f64->f32 (pairs)", k_cvt_pair, 16, 16},
        {"cvt+mul_f64+cvt (8 values)", k_quot_route, 24, 24},
        {"v_cmp_class_f32 | +s_or_b64", k_class, 8, 8},
    };
    printf("%-32s %12s %12s   (cycles per instruction, one wave64 alone on its SIMD; s_memtime ticks)\n", "instruction",
           "independent", "dependent");
    for (auto &c : cases) {
        uint64_t h[5];
        for (int rep = 0; rep < 2; rep++) {   // second run: instruction cache warm
            hipLaunchKernelGGL(c.fn, dim3(1), dim3(64), 0, 0, d, 1.5f);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
        printf("%-32s %12.2f %12.2f   [%.1f counter ticks per us of the 100 MHz wall clock]\n", c.name,
               (double)h[0] / (ITER * c.n_indep), (double)h[1] / (ITER * c.n_dep), (double)h[4] / ((double)h[3] / 100.0));
    }
    hipFree(d);
    return 0;
}
