// mm16.hpp -- one 128 x 128 tile of C[m][n] = sum_k A[m][k] B[n][k] on v_mfma_f32_32x32x16_f16 (gfx950).
//
// Shared by the split of a forest level (splitmm.hip: A = hyperplanes, B = rows) and by the candidate filter of a
// batch of approximate queries (knn.hip: A = queries, B = rows).  Both only FILTER with the result: whatever it
// cannot decide is recomputed with the canonical fp32 dot, so the summation order here is free.
//
// Workgroup: 512 threads = 8 waves, wave (wm, wn) owns B-rows wm * 32 .. +31 and A-rows wn * 64 .. +63 as two
// 32 x 32 MFMA tiles; K is stepped by 64 halfs.  The operand slabs go from global memory straight into LDS
// (global_load_lds_dwordx4: no staging registers, no ds_write -- the VGPR -> LDS store path, ~80 B/clk, would take as
// long as the slab's MFMAs).  One wave instruction fills 1 KiB of LDS = 8 tile rows of 128 B, lane i at byte 16 i.
// The image is unpadded, so the 16-byte chunks of row r are XOR-swizzled with (r >> 1) & 7: lane i fetches chunk
// (i & 7) ^ f(row) of its row, a reader finds chunk c of row r at position c ^ f(r), and the 16 lanes ds_read_b128
// serves per cycle fall on 16 different bank groups.  Two buffers: the slab of step k+1 lands while step k is
// multiplied; one barrier per step.  The even and the odd K-steps go to two accumulators (two chains of dpad / 2
// products: the rounding bound the callers use), added at the end.
//
// Measured accumulation error of this instruction on MI355X: scripts/mfma_accum_probe.hip,
// profiles/r02_mfma_accum_probe.txt (< 3 units of 2^-24 * sum |a_i b_i| over chains of up to 8192 products; the
// callers allow 4 units per product).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace morna {

typedef _Float16 mm16_f16x8 __attribute__((ext_vector_type(8)));
typedef float mm16_f32x16 __attribute__((ext_vector_type(16)));

#define MM16_TILE 128
#define MM16_THREADS 512
#define MM16_BK 64
#define MM16_LDS (2 * 2 * MM16_TILE * MM16_BK * 2)   // 64 KiB: two buffers of two operand slabs

// smem: MM16_LDS bytes, 16-byte aligned.  b_row(rt) / a_row(rt): global row of the B / A operand for tile row rt
// (0..127), already clamped to a valid row.  On return acc[tn][e] holds C[m][n] with
//   n (B row of the tile) = wm * 32 + (lane & 31),  m (A row of the tile) = wn * 64 + tn * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
// and every wave has passed the barrier behind the last LDS read (smem may be reused).
template <typename BRow, typename ARow>
__device__ inline void mm16_tile(const _Float16 *__restrict__ B16, const _Float16 *__restrict__ A16, int32_t dpad,
                                 unsigned char *smem, BRow b_row, ARow a_row, mm16_f32x16 (&acc)[2])
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    auto dma = [&](int k0, int buf) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = w * 4 + u;                       // 32 instructions per step: 16 for the B rows, 16 for the A rows
            const bool is_b = j < 16;
            const int rt = ((is_b ? j : j - 16) << 3) + (lane >> 3);   // row of the tile
            const int chunk = (lane & 7) ^ ((rt >> 1) & 7);
            const _Float16 *src = (is_b ? B16 + (int64_t)b_row(rt) * dpad : A16 + (int64_t)a_row(rt) * dpad) + k0 + chunk * 8;
            unsigned char *dst = smem + buf * (2 * MM16_TILE * MM16_BK * 2) + (is_b ? 0 : MM16_TILE * MM16_BK * 2) +
                                 (is_b ? j : j - 16) * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    mm16_f32x16 acc_odd[2];
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[j][e] = acc_odd[j][e] = 0.f;

    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int lr = lane & 31, lh = lane >> 5;
    const int brow = wm * 32 + lr, bsw = (brow >> 1) & 7;
    auto kstep = [&](int k0, int buf, mm16_f32x16(&ac)[2]) {
        if (k0 + MM16_BK < dpad) dma(k0 + MM16_BK, buf ^ 1);   // next slab lands in the other buffer under the MFMAs
        const unsigned char *bs = smem + buf * (2 * MM16_TILE * MM16_BK * 2);
        const unsigned char *as = bs + MM16_TILE * MM16_BK * 2;
#pragma unroll
        for (int blk = 0; blk < MM16_BK / 16; blk++) {
            // 32x32x16: lane (r = l & 31, h = l >> 5) supplies A[m = r][k = 8h + j], B[k = 8h + j][n = r], j = 0..7,
            // i.e. the 16-byte chunk 2 blk + h of its row
            const int kc = 2 * blk + lh;
            const mm16_f16x8 b8 = *(const mm16_f16x8 *)(bs + brow * 128 + ((kc ^ bsw) << 4));
#pragma unroll
            for (int tn = 0; tn < 2; tn++) {
                const int arow = wn * 64 + tn * 32 + lr;
                const mm16_f16x8 a8 = *(const mm16_f16x8 *)(as + arow * 128 + ((kc ^ ((arow >> 1) & 7)) << 4));
                ac[tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, ac[tn], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the next slab is in LDS
        __syncthreads();
    };
    // dpad is a multiple of 256: an even number of K-steps; the buffer index is the step's parity
    for (int k0 = 0; k0 < dpad; k0 += 2 * MM16_BK) {
        kstep(k0, 0, acc);
        kstep(k0 + MM16_BK, 1, acc_odd);
    }
#pragma unroll
    for (int tn = 0; tn < 2; tn++) acc[tn] = acc[tn] + acc_odd[tn];
}

// EACC of the callers' bound: |sum y_i g_i (as computed) - exact| <= EACC |y| |g| for two chains of dpad / 2 products,
// 4 * 2^-24 per product, + the rounding of the canonical fp32 dot it stands in for (< 4e-6)
static inline float mm16_eacc(int32_t dpad) { return (2.f * (float)dpad + 2.f) * 5.9604645e-8f + 4.1e-6f; }

}  // namespace morna
