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

#define MM16_TILE 128    // A rows (hyperplanes / queries) of a workgroup tile in the 8-wave forms
#define MM16_THREADS 512
#define MM16_LDS_OF(BR, BK, NS) ((NS) * ((BR) + MM16_TILE) * (BK) * 2)   // NS stages of a BR-row and a 128-row slab of BK halfs
#define MM16_LDS MM16_LDS_OF(128, 64, 2)

// The loop shared by every form.  WM x WN waves; wave (wm, wn) owns NB 32-row MFMA tiles of B (rows wm * 32 NB ..) and
// two of A (rows wn * 64 ..); a stage = the B slab (BR = 32 NB WM rows), then the A slab (AR = 64 WN rows), rows of BK
// halfs, the 16-byte chunks of row r XOR-swizzled (lane i of a delivery fetches chunk (i % LPR) ^ f(row) of its row, a
// reader finds chunk c of row r at position c ^ f(r), so the lanes a ds_read_b128 serves together fall on different
// bank groups).  CH = 2: the even and the odd 16-wide K blocks go to two accumulators (two chains of dpad / 2
// products: the bound the callers use); CH = 1: one chain of dpad products.
//   * the source address of each delivery is fixed per lane for the whole loop: the row look-ups happen once, before it;
//   * the fragments of K block b + 1 are read from LDS while the MFMAs of block b run (round 1's loop waited for its two
//     or three ds_read_b128 in front of EVERY MFMA);
//   * NS stages, NS - 1 deliveries in flight (NS = 2: one).
// On return acc[tb][tn][e] holds C[m][n] with n (B row of the tile) = wm * 32 NB + tb * 32 + (lane & 31) and
// m (A row of the tile) = wn * 64 + tn * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); every wave has passed a barrier
// behind the last LDS read (smem may be reused).
#ifndef MM16_PROBE
#define MM16_PROBE 0   // experiments: 1 = no MFMAs (deliveries + fragment reads only), 2 = no deliveries after the first, 3 = neither reads nor MFMAs
#endif
template <int WM, int WN, int NB, int BK, int NS, int CH, typename BRow, typename ARow>
__device__ inline void mm16_core(const _Float16 *__restrict__ B16, const _Float16 *__restrict__ A16, int32_t dpad,
                                 unsigned char *smem, BRow b_row, ARow a_row, mm16_f32x16 (&acc)[NB][2])
{
    static_assert(BK == 32 || BK == 64, "K-step of 32 or 64 halfs");
    constexpr int BR = 32 * NB * WM, AR = 64 * WN, NW = WM * WN;
    constexpr int ROWB = BK * 2;                        // bytes of a tile row in LDS
    constexpr int RPI = 1024 / ROWB;                    // tile rows one delivery fills (1 KiB per wave instruction)
    constexpr int LPR = ROWB / 16;                      // lanes (16-byte chunks) per row
    constexpr int STAGE = (BR + AR) * ROWB;             // bytes of one stage
    constexpr int NDMA = (BR + AR) / RPI;               // deliveries that fill a stage
    constexpr int PW = NDMA / NW;                       // ... per wave
    constexpr int NBLK = BK / 16;
    static_assert(NDMA % NW == 0, "a stage is filled by all waves alike");
    static_assert(PW * (NS - 2) < 64, "vmcnt is a 6-bit counter");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;
    auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };
    const _Float16 *src[PW];
#pragma unroll
    for (int u = 0; u < PW; u++) {
        const int j = w * PW + u;
        const bool is_b = j < BR / RPI;
        const int rt = (is_b ? j : j - BR / RPI) * RPI + lane / LPR;   // row of the tile
        const int chunk = (lane % LPR) ^ swz(rt);
        src[u] = (is_b ? B16 + (int64_t)b_row(rt) * dpad : A16 + (int64_t)a_row(rt) * dpad) + chunk * 8;
    }
    auto dma = [&](int k0, int stage) {
#pragma unroll
        for (int u = 0; u < PW; u++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[u] + k0),
                                             (__attribute__((address_space(3))) void *)(smem + stage * STAGE + (w * PW + u) * 1024),
                                             16, 0, 0);
    };
    mm16_f32x16 acc_odd[CH == 2 ? NB : 1][2];
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                acc[i][j][e] = 0.f;
                if (CH == 2) acc_odd[i][j][e] = 0.f;
            }

    const int nsteps = dpad / BK;
#pragma unroll
    for (int st = 0; st < NS - 1; st++)
        if (st < nsteps) dma(st * BK, st);
    const int lr = lane & 31, lh = lane >> 5;
    // byte offsets of this lane's fragments inside a slab, K block 0; block b is chunk 2 b + lh: XOR with (2 b) << 4
    // commutes with the swizzle, so block b's offset is the block-0 offset ^ (b << 5)
    int boff[NB], aoff[2];
#pragma unroll
    for (int tb = 0; tb < NB; tb++) {
        const int brow = wm * 32 * NB + tb * 32 + lr;
        boff[tb] = brow * ROWB + ((lh ^ swz(brow)) << 4);
    }
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int arow = wn * 64 + tn * 32 + lr;
        aoff[tn] = BR * ROWB + arow * ROWB + ((lh ^ swz(arow)) << 4);
    }
    for (int k = 0; k < nsteps; k++) {
        // stage k has landed when at most the deliveries of the NS - 2 stages issued after it are outstanding (they
        // complete in order); near the end fewer were issued: drain
        if (NS > 2 && k + NS - 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // every wave's part of stage k is in LDS, and every wave is done reading stage k - 1
        if (MM16_PROBE != 2 && k + NS - 1 < nsteps) dma((k + NS - 1) * BK, (k + NS - 1) % NS);   // into the stage read in step k - 1
        if (MM16_PROBE == 3) continue;
        const unsigned char *st = smem + (k % NS) * STAGE;
        mm16_f16x8 b8[2][NB], a8[2][2];
#pragma unroll
        for (int tb = 0; tb < NB; tb++) b8[0][tb] = *(const mm16_f16x8 *)(st + boff[tb]);
#pragma unroll
        for (int tn = 0; tn < 2; tn++) a8[0][tn] = *(const mm16_f16x8 *)(st + aoff[tn]);
#pragma unroll
        for (int blk = 0; blk < NBLK; blk++) {
            // 32x32x16: lane (r = l & 31, h = l >> 5) supplies A[m = r][k = 8h + j], B[k = 8h + j][n = r], j = 0..7,
            // i.e. the 16-byte chunk 2 blk + h of its row
            const int cur = blk & 1, nxt = cur ^ 1;
            if (blk + 1 < NBLK) {
#pragma unroll
                for (int tb = 0; tb < NB; tb++) b8[nxt][tb] = *(const mm16_f16x8 *)(st + (boff[tb] ^ ((blk + 1) << 5)));
#pragma unroll
                for (int tn = 0; tn < 2; tn++) a8[nxt][tn] = *(const mm16_f16x8 *)(st + (aoff[tn] ^ ((blk + 1) << 5)));
            }
#pragma unroll
            for (int tb = 0; tb < NB; tb++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++) {
                    if (MM16_PROBE == 1) {   // keep the reads alive without the matrix cores
                        acc[tb][tn][0] += (float)a8[cur][tn][0] + (float)b8[cur][tb][0];
                    } else if (CH == 2 && (blk & 1)) acc_odd[tb][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8[cur][tn], b8[cur][tb], acc_odd[tb][tn], 0, 0, 0);
                    else acc[tb][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8[cur][tn], b8[cur][tb], acc[tb][tn], 0, 0, 0);
                }
            // pin the order: the next block's reads go out BEFORE this block's MFMAs (left alone, the scheduler puts every
            // read right in front of its use, in one register set, and each MFMA waits for LDS)
            if (MM16_PROBE == 0 || MM16_PROBE == 2) {
                if (blk + 1 < NBLK) __builtin_amdgcn_sched_group_barrier(0x100, NB + 2, 0);   // DS reads
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * NB, 0);                        // MFMAs
            }
        }
    }
    __syncthreads();   // the last stage has been read by everybody: smem is free
    if (CH == 2) {
#pragma unroll
        for (int tb = 0; tb < NB; tb++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) acc[tb][tn] = acc[tb][tn] + acc_odd[tb][tn];
    }
}

// 8 waves, BR x 128 tile (BR = 128: 32 x 64 per wave; BR = 256: 64 x 64 per wave), two chains
template <int BR, int BK, int NS, typename BRow, typename ARow>
__device__ inline void mm16_tile(const _Float16 *__restrict__ B16, const _Float16 *__restrict__ A16, int32_t dpad,
                                 unsigned char *smem, BRow b_row, ARow a_row, mm16_f32x16 (&acc)[BR / 128][2])
{
    mm16_core<4, 2, BR / 128, BK, NS, 2>(B16, A16, dpad, smem, b_row, a_row, acc);
}

// 16 waves (1024 threads), 256 x 256 tile, 64 x 64 per wave, ONE chain of dpad products (the register file holds no
// second set of accumulators at four waves per SIMD); smem: NS stages of 512 rows of BK halfs
template <int BK, int NS, typename BRow, typename ARow>
__device__ inline void mm16_tile_256x256(const _Float16 *__restrict__ B16, const _Float16 *__restrict__ A16, int32_t dpad,
                                         unsigned char *smem, BRow b_row, ARow a_row, mm16_f32x16 (&acc)[2][2])
{
    mm16_core<4, 4, 2, BK, NS, 1>(B16, A16, dpad, smem, b_row, a_row, acc);
}

// EACC of the callers' bound: |sum y_i g_i (as computed) - exact| <= EACC |y| |g| for two chains of dpad / 2 products,
// 4 * 2^-24 per product, + the rounding of the canonical fp32 dot it stands in for (< 4e-6)
static inline float mm16_eacc(int32_t dpad) { return (2.f * (float)dpad + 2.f) * 5.9604645e-8f + 4.1e-6f; }

}  // namespace morna
