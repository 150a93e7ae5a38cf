// mfma_accum_probe.hip -- how the gfx950 matrix cores round while they accumulate, measured.
//
// Two filters of the library take dot products on MFMA instructions and reason about the distance between the
// MFMA result and the exact sum (splitmm.hip: v_mfma_f32_32x32x16_f16, EACC; knn.hip: v_mfma_f32_32x32x2_f32,
// exact_scan_eps).  The ISA manual does not state the internal summation order or rounding, so this program
// measures it: chains of L dependent MFMAs on random operands (wide dynamic range, mixed signs), compared on the
// host with (a) candidate bit-exact models and (b) the exact sum, the error expressed in units of
// u * sum|a_i b_i| (u = 2^-24) and per product.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/mfma_accum_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// A[32][K], B[K][32] row-major fp32; D[32][32]; one wave.  K = 2 L.
__global__ void chain_f32(const float *A, const float *B, int L, float *D)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    for (int s = 0; s < L; s++) {
        const int k = 2 * s + h;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 2 * L + k], B[k * 32 + r], acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; e++) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
}

// A[32][K], B[32][K] (B given transposed: B[n][k]) fp16; K = 16 L
__global__ void chain_f16(const _Float16 *A, const _Float16 *Bt, int L, float *D)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    for (int s = 0; s < L; s++) {
        const f16x8 a = *(const f16x8 *)(A + r * 16 * L + 16 * s + 8 * h);
        const f16x8 b = *(const f16x8 *)(Bt + r * 16 * L + 16 * s + 8 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; e++) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}
static float rnd_float(int spread)   // random sign, mantissa, exponent in [-spread, 0]
{
    const float m = 1.0f + (float)(rnd() & 0x7FFFFF) / 8388608.0f;
    const float v = ldexpf(m, -(int)(rnd() % (uint32_t)(spread + 1)));
    return (rnd() & 1) ? -v : v;
}

#define CK(e)                                                                    \
    do {                                                                         \
        hipError_t _e = (e);                                                     \
        if (_e != hipSuccess) {                                                  \
            printf("%s failed: %s\n", #e, hipGetErrorString(_e));                \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main()
{
    printf("# mfma_accum_probe: error of a chain of L dependent MFMAs against the exact sum\n");
    printf("# err/u/S = |mfma - exact| / (2^-24 * sum|a_i b_i|);  /prod = that divided by the number of products\n");
    const int Ls32[] = {1, 2, 8, 64, 512, 1536, 4096};
    printf("\n[v_mfma_f32_32x32x2_f32]  K = 2 L\n");
    printf("%6s %8s | %14s %14s | %10s %10s\n", "L", "products", "== fma k0,k1", "== fma k1,k0", "max err/u/S", "/prod");
    for (int L : Ls32) {
        const int K = 2 * L;
        std::vector<float> A(32 * K), B(K * 32), D(1024);
        for (auto &v : A) v = rnd_float(12);
        for (auto &v : B) v = rnd_float(12);
        float *dA, *dB, *dD;
        CK(hipMalloc(&dA, A.size() * 4));
        CK(hipMalloc(&dB, B.size() * 4));
        CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(chain_f32, dim3(1), dim3(64), 0, 0, dA, dB, L, dD);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        int eq01 = 0, eq10 = 0;
        double worst = 0;
        for (int m = 0; m < 32; m++)
            for (int n = 0; n < 32; n++) {
                float s01 = 0.f, s10 = 0.f;
                long double ex = 0, sabs = 0;
                for (int s = 0; s < L; s++) {
                    const float a0 = A[m * K + 2 * s], a1 = A[m * K + 2 * s + 1], b0 = B[(2 * s) * 32 + n], b1 = B[(2 * s + 1) * 32 + n];
                    s01 = fmaf(a1, b1, fmaf(a0, b0, s01));
                    s10 = fmaf(a0, b0, fmaf(a1, b1, s10));
                    ex += (long double)a0 * b0 + (long double)a1 * b1;
                    sabs += fabsl((long double)a0 * b0) + fabsl((long double)a1 * b1);
                }
                const float got = D[m * 32 + n];
                eq01 += memcmp(&got, &s01, 4) == 0;
                eq10 += memcmp(&got, &s10, 4) == 0;
                const double e = (double)(fabsl((long double)got - ex) / (sabs * 5.9604644775390625e-8L));
                if (e > worst) worst = e;
            }
        printf("%6d %8d | %9d/1024 %9d/1024 | %10.3f %10.5f\n", L, K, eq01, eq10, worst, worst / K);
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
    }

    const int Ls16[] = {1, 2, 8, 24, 96, 192, 256, 512};
    printf("\n[v_mfma_f32_32x32x16_f16]  K = 16 L; products of halfs are exact in fp32\n");
    printf("%6s %8s | %16s %16s | %10s %10s\n", "L", "products", "== RN(acc+sum16)", "== RZ(acc+sum16)", "max err/u/S", "/prod");
    for (int L : Ls16) {
        const int K = 16 * L;
        std::vector<_Float16> A(32 * K), B(32 * K);
        std::vector<float> D(1024);
        // operands as the library scales them: largest element of a row in [2^14, 2^15), the others up to 2^-10 of it
        for (auto &v : A) v = (_Float16)(rnd_float(10) * 16384.0f);
        for (auto &v : B) v = (_Float16)(rnd_float(10) * 16384.0f);
        _Float16 *dA, *dB;
        float *dD;
        CK(hipMalloc(&dA, A.size() * 2));
        CK(hipMalloc(&dB, B.size() * 2));
        CK(hipMalloc(&dD, 4096));
        CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(chain_f16, dim3(1), dim3(64), 0, 0, dA, dB, L, dD);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
        int eq_rn = 0, eq_rz = 0;
        double worst = 0;
        for (int m = 0; m < 32; m++)
            for (int n = 0; n < 32; n++) {
                float rn = 0.f, rz = 0.f;
                long double ex = 0, sabs = 0;
                for (int s = 0; s < L; s++) {
                    long double part_rn = rn, part_rz = rz;
                    for (int j = 0; j < 16; j++) {
                        const long double p = (long double)(float)A[m * K + 16 * s + j] * (long double)(float)B[n * K + 16 * s + j];
                        part_rn += p;
                        part_rz += p;
                        ex += p;
                        sabs += fabsl(p);
                    }
                    rn = (float)part_rn;   // x87 long double holds acc + 16 products of <= 22 bits exactly here
                    float z = (float)part_rz;
                    if (fabsl((long double)z) > fabsl(part_rz)) z = nextafterf(z, 0.f);
                    rz = z;
                }
                const float got = D[m * 32 + n];
                eq_rn += memcmp(&got, &rn, 4) == 0;
                eq_rz += memcmp(&got, &rz, 4) == 0;
                const double e = (double)(fabsl((long double)got - ex) / (sabs * 5.9604644775390625e-8L));
                if (e > worst) worst = e;
            }
        printf("%6d %8d | %11d/1024 %11d/1024 | %10.3f %10.5f\n", L, K, eq_rn, eq_rz, worst, worst / K);
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dD);
    }
    printf("\n# bounds the library assumes, same units: exact scan (fp32 MFMA) 1 per product of a chain;\n");
    printf("# split / query filters (fp16 MFMA) 4 per product of a chain (splitmm.hip EACC).\n");
    return 0;
}
