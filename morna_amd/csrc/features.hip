// features.hip -- feature-hashed TF-IDF matrix build on gfx950.
//
// Replaces the add_junction loop and the add_item hand-off of the reference
// (commanderson/morna morna.py:344-388, 405-424).  Parity contract: the fp32
// matrix is BIT-IDENTICAL to the reference's, which sums fp64 contributions per
// (sample, column) cell in file order and rounds once to fp32.
//
// Data flow (all in HBM; the nnz stream is read once, coalesced):
//   hash_keys_kernel     J keys -> column, sign*idf; counts lines per column
//   col_scan / col_fill  lines bucketed by column, file order kept inside a bucket
//   line_flags_kernel    does a sample repeat inside a line? (LDS bitmap per line)
//   accumulate_kernel    one workgroup per (column, tile of 8192 samples): the
//                        tile's fp64 accumulators live in LDS; the lines of the
//                        column are walked in file order, the lanes pick the
//                        entries whose sample falls in the tile, one workgroup
//                        barrier between lines.  Different (column, tile) pairs
//                        never share a cell and a line touches a cell at most
//                        once (else it is replayed serially), so the per-cell
//                        order is the file order, with no atomics.
//   transpose_convert    fp32 [D][N] (each cell rounded once from its fp64 sum) -> fp32 [N][dpad] through an LDS tile
//   row_norms_kernel     canonical dot(x, x) per row
#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "common.hpp"
#include "devutil.hpp"

namespace morna {

// small clears as a kernel of our own: a hipMemsetAsync brings ~15 us of idle device with it
__global__ __launch_bounds__(256) void zero_i32_kernel(int32_t *__restrict__ p, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0;
}

// ------------------------------------------------------------------ hash pass

__global__ void hash_keys_kernel(const uint8_t *__restrict__ keys, const int64_t *__restrict__ key_off,
                                 int64_t J, int32_t dim, const double *__restrict__ idf,
                                 int32_t *__restrict__ col_out, double *__restrict__ sidf_out,
                                 int32_t *__restrict__ hash_out, int32_t *__restrict__ sign_out,
                                 int32_t *__restrict__ col_count)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= J) return;
    int64_t b = key_off[j], e = key_off[j + 1];
    int32_t h = (int32_t)murmur3_32(keys + b, e - b, 0u);   // mmh3.hash, morna.py:369
    int32_t col = floored_mod(h, dim);                      // morna.py:371
    col_out[j] = col;
    if (sidf_out) sidf_out[j] = h < 0 ? -idf[j] : idf[j];   // multiplier * (cov * idf): sign is exact
    if (hash_out) hash_out[j] = h;
    if (sign_out) sign_out[j] = h < 0 ? -1 : 1;             // morna.py:370, before the modulo
    if (col_count) atomicAdd(&col_count[col], 1);
}

// exclusive scan of the per-column line counts (one workgroup)
__global__ __launch_bounds__(1024) void col_scan_kernel(const int32_t *__restrict__ count, int32_t dim,
                                                        int32_t *__restrict__ off /* [dim + 1] */)
{
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    const int per = (dim + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < dim ? lo + per : dim;
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += count[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int v = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = tid ? s_part[tid - 1] : 0;
    for (int i = lo; i < hi; i++) {
        off[i] = run;
        run += count[i];
    }
    if (tid == 1023) off[dim] = s_part[1023];
}

#define ACC_THREADS 1024

// Lines bucketed by column, file order kept inside a bucket: a stable counting sort of the J lines by column.
//   col_rank_kernel   one workgroup per chunk of CR_LINES consecutive lines: rank of a line among the EARLIER lines
//                     of its chunk with the same column (compare against the chunk's columns in LDS), and the
//                     chunk's line count per column (chunk_cnt[column][chunk], zeroed before)
//   col_base_kernel   per column: exclusive prefix of those counts over the chunks, in place (a wave per column)
//   col_place_kernel  lines[off[column] + base[column][chunk] + rank] = line
// (The first version scanned all J lines once per column: 2.1e8 comparisons at C3, 0.4 ms.  Round 3: chunks of 256 lines
// instead of 1024 -- a shard of the row cut keeps all ~J lines of the file, so at 6250 rows this bucketing, 69 workgroups
// whose last thread compared against 1023 earlier lines, was 0.17 of the feature build's 0.29 ms -- and the counts by column
// first, so that a column's prefix is a coalesced wave scan: 0.22 -> 0.06 ms for the four kernels.)
#define CR_LINES 256
__global__ __launch_bounds__(CR_LINES) void col_rank_kernel(const int32_t *__restrict__ col, int64_t J, int32_t n_chunks,
                                                            int32_t *__restrict__ rank, int32_t *__restrict__ chunk_cnt)
{
    __shared__ __attribute__((aligned(16))) int32_t s_col[CR_LINES];
    const int i = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * CR_LINES + i;
    const int32_t my = j < J ? col[j] : -1;
    s_col[i] = my;
    __syncthreads();
    if (my < 0) return;
    int r = 0;
    const int4 *v = (const int4 *)s_col;
    int k4 = 0;
    for (; k4 < (i >> 2); k4++) {   // all lanes of a wave read the same address: broadcast
        const int4 c = v[k4];
        r += (c.x == my) + (c.y == my) + (c.z == my) + (c.w == my);
    }
    for (int k = k4 * 4; k < i; k++) r += s_col[k] == my;
    rank[j] = r;
    atomicAdd(&chunk_cnt[(int64_t)my * n_chunks + blockIdx.x], 1);
}

// one wave per column: chunk_cnt[column][0 .. n_chunks) -> its exclusive prefix
__global__ __launch_bounds__(256) void col_base_kernel(int32_t *__restrict__ chunk_cnt, int32_t n_chunks, int32_t dim)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int c = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
    if (c >= dim) return;
    int32_t *cnt = chunk_cnt + (int64_t)c * n_chunks;
    int run = 0;
    for (int k0 = 0; k0 < n_chunks; k0 += WAVE) {
        const int k = k0 + lane;
        const int t = k < n_chunks ? cnt[k] : 0;
        int incl = t;   // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const int o = __shfl_up(incl, off, WAVE);
            if (lane >= off) incl += o;
        }
        if (k < n_chunks) cnt[k] = run + incl - t;
        run += __shfl(incl, WAVE - 1, WAVE);
    }
}

__global__ __launch_bounds__(256) void col_place_kernel(const int32_t *__restrict__ col, int64_t J, int32_t n_chunks,
                                                        const int32_t *__restrict__ rank, const int32_t *__restrict__ chunk_base,
                                                        const int32_t *__restrict__ off, int32_t *__restrict__ lines)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    const int32_t c = col[j];
    lines[off[c] + chunk_base[(int64_t)c * n_chunks + j / CR_LINES] + rank[j]] = (int32_t)j;
}

// One workgroup per line: does any sample id repeat inside the line?  (Internal
// ids follow first-seen order, so a line's id list is generally NOT sorted even
// though the file's sample list is.)  Uses an LDS bitmap over the samples.
#define FLAG_MAX_WORDS 32768   // 128 KiB of LDS: up to 2^20 samples
#define LF_THREADS 1024   // a line holds ~1.4 k samples: one or two per thread
// upgrade != 0: only the lines whose flag is already non-zero are examined, and their flag becomes 1 + (a sample
// repeats) -- the lines line_prep_kernel found not ascending (accumulate_piece_kernel's flags 0 / 1 / 2).
__global__ __launch_bounds__(LF_THREADS) void line_flags_kernel(const int64_t *__restrict__ row_ptr,
                                                         const int32_t *__restrict__ ids, int64_t J,
                                                         int32_t n_words, uint8_t *__restrict__ flag_out, int32_t upgrade)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *bm = (uint32_t *)smem;
    __shared__ int s_dup;
    const int tid = threadIdx.x;
    for (int i = tid; i < n_words; i += LF_THREADS) bm[i] = 0u;
    __syncthreads();
    for (int64_t j = blockIdx.x; j < J; j += gridDim.x) {
        if (upgrade && flag_out[j] == 0) continue;   // uniform over the workgroup; the flag is only written below
        const int64_t b = row_ptr[j], e = row_ptr[j + 1];
        if (tid == 0) s_dup = 0;
        __syncthreads();
        int dup = 0;
        for (int64_t t = b + tid; t < e; t += LF_THREADS) {
            const uint32_t id = (uint32_t)ids[t];
            const uint32_t bit = 1u << (id & 31);
            dup |= (atomicOr(&bm[id >> 5], bit) & bit) != 0;
        }
        if (dup) s_dup = 1;
        __syncthreads();
        if (tid == 0) flag_out[j] = (uint8_t)(s_dup + (upgrade ? 1 : 0));
        for (int64_t t = b + tid; t < e; t += LF_THREADS) bm[(uint32_t)ids[t] >> 5] = 0u;   // un-set only what was set
        __syncthreads();
    }
}

// The same question with one WAVE per line and a bitmap of its own per wave: a line holds ~1.4 k ids, so a
// workgroup of 1024 threads per line spends its time in four barriers per line.  LDS operations of one wave are
// performed in order: the atomics of a line, the stores that clear its bits and the atomics of the wave's next line
// need no barrier between them.  Used while 16 bitmaps fit LDS (n_words <= LFW_MAX_WORDS: up to 65536 samples).
#define LFW_WAVES 16
#define LFW_MAX_WORDS 2048
#define LFW_UNROLL 8
#define LFW_HELD 32
__global__ __launch_bounds__(LFW_WAVES * WAVE) void line_flags_wave_kernel(const int64_t *__restrict__ row_ptr,
                                                                            const int32_t *__restrict__ ids, int64_t J,
                                                                            int32_t n_words, uint8_t *__restrict__ flag_out,
                                                                            int32_t upgrade)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    uint32_t *bm = (uint32_t *)smem + (size_t)w * n_words;
    for (int i = lane; i < n_words; i += WAVE) bm[i] = 0u;
    const int64_t n_waves = (int64_t)gridDim.x * LFW_WAVES;
    const uint8_t up = upgrade ? 1 : 0;
    for (int64_t j = (int64_t)blockIdx.x * LFW_WAVES + w; j < J; j += n_waves) {
        if (upgrade && flag_out[j] == 0) continue;   // uniform over the wave
        const int64_t b = row_ptr[j], e = row_ptr[j + 1];
        int dup = 0;
        if (e - b <= (int64_t)WAVE * LFW_HELD) {
            // the whole line in registers (up to 2048 ids): every load in flight at once, and the bits are cleared
            // without reading the line again
            uint32_t id[LFW_HELD];
#pragma unroll
            for (int u = 0; u < LFW_HELD; u++) id[u] = b + lane + u * WAVE < e ? (uint32_t)ids[b + lane + u * WAVE] : 0xffffffffu;
#pragma unroll
            for (int u = 0; u < LFW_HELD; u++)
                if (id[u] != 0xffffffffu) {
                    const uint32_t bit = 1u << (id[u] & 31);
                    dup |= (atomicOr(&bm[id[u] >> 5], bit) & bit) != 0;
                }
            const bool any_held = __ballot(dup) != 0;
#pragma unroll
            for (int u = 0; u < LFW_HELD; u++)
                if (id[u] != 0xffffffffu) bm[id[u] >> 5] = 0u;
            if (lane == 0) flag_out[j] = (uint8_t)(any_held + up);
            continue;
        }
        // LFW_UNROLL loads in flight per lane: the line's ids come from HBM, one dependent round trip per load otherwise
        for (int64_t t0 = b + lane; t0 < e; t0 += (int64_t)WAVE * LFW_UNROLL) {
            uint32_t id[LFW_UNROLL];
#pragma unroll
            for (int u = 0; u < LFW_UNROLL; u++) id[u] = t0 + u * WAVE < e ? (uint32_t)ids[t0 + u * WAVE] : 0xffffffffu;
#pragma unroll
            for (int u = 0; u < LFW_UNROLL; u++)
                if (id[u] != 0xffffffffu) {
                    const uint32_t bit = 1u << (id[u] & 31);
                    dup |= (atomicOr(&bm[id[u] >> 5], bit) & bit) != 0;   // two lanes with the same id: one of them sees the bit
                }
        }
        const bool any = __ballot(dup) != 0;
        for (int64_t t0 = b + lane; t0 < e; t0 += (int64_t)WAVE * LFW_UNROLL) {   // un-set only what was set
            uint32_t id[LFW_UNROLL];
#pragma unroll
            for (int u = 0; u < LFW_UNROLL; u++) id[u] = t0 + u * WAVE < e ? (uint32_t)ids[t0 + u * WAVE] : 0xffffffffu;
#pragma unroll
            for (int u = 0; u < LFW_UNROLL; u++)
                if (id[u] != 0xffffffffu) bm[id[u] >> 5] = 0u;
        }
        if (lane == 0) flag_out[j] = (uint8_t)(any + up);
    }
}

__global__ void flags_to_serial_kernel(uint8_t *__restrict__ flag, int64_t J)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < J && flag[j]) flag[j] = 2;
}

// ------------------------------------------- lines in item ORDER: tile extents
//
// The caller may hand over an ORDER of the items (morna_stage_item_order: any key in which the sample lists of the
// lines ascend -- for an intropolis file, the external sample id: its lines list their samples in ascending order,
// while the internal ids are handed out first-seen, morna.py:377-382).  Under that order p = rank[item] a line is a
// strictly ascending sequence, hence (a) no sample repeats in it and (b) the entries that fall into a tile of
// consecutive p are a contiguous piece of the line.  line_prep_kernel checks the line and records where each tile's
// piece begins; accumulate_piece_kernel then reads each entry of the nnz stream ONCE (the tile-by-tile form above reads
// every line once per tile).  A line that does not ascend is flagged and handled the old way, so the result never
// depends on the hint.
#define LP_HELD 32   // entries per lane held in registers: lines of up to 2048 entries in one piece
// One piece of a line (up to NU * 64 entries from entry c0): positions of its items, ascending test, tile starts.
// Branch-free over the NU register sets -- every load of a phase is in flight before the first result is used; entries
// past the end of the line re-read its first entry and are masked afterwards.
template <int TILE_SHIFT, int NU, typename RankT>
__device__ inline void line_prep_piece(const int32_t *__restrict__ ids, const RankT *__restrict__ rank, int32_t n_items, int64_t b,
                                       int32_t len, int32_t c0, int32_t n_tiles, int32_t *__restrict__ off, int32_t *__restrict__ pid,
                                       int lane, bool &asc, int32_t &prev_last)
{
    int32_t p[NU];
    bool bad = false;   // an id outside [0, n_items): such an entry has no cell (the tile-by-tile form skips it the same way)
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int32_t e = c0 + u * WAVE + lane;
        p[u] = ids[b + (e < len ? e : c0)];
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const bool ok = (uint32_t)p[u] < (uint32_t)n_items;
        bad |= !ok;
        const int32_t r = (int32_t)rank[ok ? p[u] : 0];
        p[u] = ok ? r : -1;
    }
    if (__any(bad)) asc = false;   // the line takes the scanned path, where a position of -1 falls into no tile
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int32_t e = c0 + u * WAVE + lane;
        if (e < len) pid[b + e] = p[u];
        else p[u] = INT32_MAX;   // past the end: in no tile
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int32_t e = c0 + u * WAVE + lane;
        int32_t prv = __shfl_up(p[u], 1, WAVE);
        const int32_t carry = u == 0 ? prev_last : __builtin_amdgcn_readlane(p[u > 0 ? u - 1 : 0], WAVE - 1);
        if (lane == 0) prv = carry;
        const bool valid = e < len;
        asc = asc && (!valid || p[u] > prv);
        const int32_t tc = p[u] >> TILE_SHIFT, tp = prv < 0 ? -1 : prv >> TILE_SHIFT;
        const bool edge = valid && tc != tp;   // a tile (or several) begins at this entry: ~n_tiles times per line
        if (__any(edge)) {
            if (edge)
                for (int32_t t = tp + 1; t <= tc && t <= n_tiles; t++) off[t] = e;
        }
    }
    const int32_t in_piece = len - c0 < WAVE * NU ? len - c0 : WAVE * NU;
    const int32_t last_e = in_piece - 1;
    int32_t v = -1;
#pragma unroll
    for (int u = 0; u < NU; u++)
        if (u == last_e / WAVE) v = p[u];
    prev_last = __shfl(v, last_e % WAVE, WAVE);
}

// the lines j = wave, wave + n_waves, ... of one wave
template <int TILE_SHIFT, typename RankT>
__device__ inline void line_prep_lines(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ ids,
                                       const RankT *__restrict__ rank, int32_t n_items, int64_t J, int32_t n_tiles,
                                       int32_t *__restrict__ tile_off, int32_t *__restrict__ pid, uint8_t *__restrict__ flag,
                                       int64_t wave, int64_t n_waves, int lane)
{
    for (int64_t j = wave; j < J; j += n_waves) {
        const int64_t b = row_ptr[j];
        const int32_t len = (int32_t)(row_ptr[j + 1] - b);
        int32_t *off = tile_off + j * (n_tiles + 1);
        bool asc = true;
        int32_t prev_last = -1;   // position of the entry before this piece
        for (int32_t c0 = 0; c0 < len;) {
            const int32_t left = len - c0;   // uniform: the piece takes the smallest register set that holds what is left
            if (left > WAVE * 24) {
                line_prep_piece<TILE_SHIFT, 32>(ids, rank, n_items, b, len, c0, n_tiles, off, pid, lane, asc, prev_last);
                c0 += WAVE * 32;
            } else if (left > WAVE * 16) {
                line_prep_piece<TILE_SHIFT, 24>(ids, rank, n_items, b, len, c0, n_tiles, off, pid, lane, asc, prev_last);
                c0 += WAVE * 24;
            } else if (left > WAVE * 8) {
                line_prep_piece<TILE_SHIFT, 16>(ids, rank, n_items, b, len, c0, n_tiles, off, pid, lane, asc, prev_last);
                c0 += WAVE * 16;
            } else {
                line_prep_piece<TILE_SHIFT, 8>(ids, rank, n_items, b, len, c0, n_tiles, off, pid, lane, asc, prev_last);
                c0 += WAVE * 8;
            }
        }
        const bool all_asc = __all(asc);
        const int32_t tl = prev_last < 0 ? -1 : prev_last >> TILE_SHIFT;
        for (int32_t t = tl + 1 + lane; t <= n_tiles; t += WAVE) off[t] = len;   // tiles behind the last entry (and the end)
        if (lane == 0) flag[j] = all_asc ? 0 : 1;
    }
}

template <int TILE_SHIFT>
__global__ __launch_bounds__(256) void line_prep_kernel(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ ids,
                                                        const int32_t *__restrict__ rank, int32_t n_items, int64_t J, int32_t n_tiles,
                                                        int32_t *__restrict__ tile_off /* [J][n_tiles + 1] */,
                                                        int32_t *__restrict__ pid /* [nnz] position of each entry's item */,
                                                        uint8_t *__restrict__ flag)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) / WAVE, n_waves = (int64_t)gridDim.x * (256 / WAVE);
    line_prep_lines<TILE_SHIFT>(row_ptr, ids, rank, n_items, J, n_tiles, tile_off, pid, flag, wave, n_waves, lane);
}

// The same with the item -> position table in LDS as 16-bit entries (up to 65536 items: 128 KB), one workgroup per CU.
// The look-up rank[id] is a gather of 64 scattered words per instruction: served by the L2 it is the kernel's bound
// (one cache line per lane), served by LDS it costs a few clocks.
#define LPL_THREADS 768
template <int TILE_SHIFT>
__global__ __launch_bounds__(LPL_THREADS) void line_prep_lds_kernel(const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ ids,
                                                                    const int32_t *__restrict__ rank, int32_t n_items, int64_t J,
                                                                    int32_t n_tiles, int32_t *__restrict__ tile_off,
                                                                    int32_t *__restrict__ pid, uint8_t *__restrict__ flag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *tab = (uint16_t *)smem;
    for (int i = threadIdx.x; i < n_items; i += LPL_THREADS) tab[i] = (uint16_t)rank[i];
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * LPL_THREADS + threadIdx.x) / WAVE, n_waves = (int64_t)gridDim.x * (LPL_THREADS / WAVE);
    line_prep_lines<TILE_SHIFT>(row_ptr, ids, tab, n_items, J, n_tiles, tile_off, pid, flag, wave, n_waves, lane);
}

// One workgroup of four waves per (column, tile of TILE consecutive positions of the item order): the tile's fp64
// accumulators live in LDS, the column's lines are walked in file order, and of each line only the piece that falls
// into the tile is read -- a thread has at most one entry of a piece (pieces longer than the workgroup take more rounds),
// so ALL the pieces of up to AW_LINES lines are fetched into registers at once, and the line loop itself is LDS
// read-modify-writes with one barrier per line (a cell may be touched by different threads in consecutive lines; a line
// touches a cell at most once: it ascends, or line_flags checked it).
//   flag 0  the line ascends in the item order: its piece [tile_off[t], tile_off[t + 1])
//   flag 1  it does not, but no sample repeats: the whole line is scanned for entries of this tile
//   flag 2  a sample repeats: thread 0 replays the line in its own order
#define AW_THREADS 256
#define AW_LINES 32    // lines per pass: their extents staged in LDS, their pieces held in registers
template <int TILE>
__global__ __launch_bounds__(AW_THREADS) void accumulate_piece_kernel(
    int32_t n_tiles, int32_t n_cols, const int32_t *__restrict__ col_off, const int32_t *__restrict__ col_lines,
    const double *__restrict__ sidf, const uint8_t *__restrict__ flags, const int64_t *__restrict__ row_ptr,
    const int32_t *__restrict__ tile_off, const int32_t *__restrict__ pid, const int32_t *__restrict__ cov,
    int64_t n_items, float *__restrict__ colacc /* [D][n_items], by position in the order */,
    int32_t min_lines /* columns with fewer lines are left to accumulate_wave_kernel */)
{
    __shared__ double acc[TILE];
    __shared__ int64_t s_b[AW_LINES];
    __shared__ double s_w[AW_LINES];
    __shared__ int32_t s_lo[AW_LINES], s_hi[AW_LINES], s_len[AW_LINES], s_fl[AW_LINES];
    const int tid = threadIdx.x;
    const int c = (int)(blockIdx.x / n_tiles), t = (int)(blockIdx.x % n_tiles);   // consecutive tiles of a column: neighbours
    const int32_t p_lo = t * TILE;
    const uint32_t span = (uint32_t)((int64_t)p_lo + TILE < n_items ? TILE : n_items - p_lo);
    const int nl = col_off[c + 1] - col_off[c];
    if (nl < min_lines) return;
    for (int i = tid; i < TILE; i += AW_THREADS) acc[i] = 0.0;
    const int32_t *lines = col_lines + col_off[c];
    const int tstride = n_tiles + 1;

    auto rmw = [&](int32_t p, int32_t cv, double wq) {
        // tf_idf = cov * idf (one rounding), then += (one rounding): morna.py:384-388
        acc[p - p_lo] = __dadd_rn(acc[p - p_lo], __dmul_rn((double)cv, wq));
    };
    for (int l0 = 0; l0 < nl; l0 += AW_LINES) {
        const int nlc = nl - l0 < AW_LINES ? nl - l0 : AW_LINES;
        __syncthreads();   // the previous pass is done with the staged extents (and the zeroing above is complete)
        if (tid < nlc) {   // one thread per line: every look-up of the pass in flight at once
            const int32_t j = lines[l0 + tid];
            const int64_t b = row_ptr[j];
            const int32_t f = (int32_t)flags[j];
            const int32_t lo = tile_off[(int64_t)j * tstride + t], hi = tile_off[(int64_t)j * tstride + t + 1];
            s_b[tid] = b;
            s_len[tid] = (int32_t)(row_ptr[j + 1] - b);
            s_fl[tid] = f;
            s_lo[tid] = f == 0 ? lo : 0;   // the extents of a line that does not ascend mean nothing
            s_hi[tid] = f == 0 ? hi : 0;
            s_w[tid] = sidf[j];
        }
        __syncthreads();
        // every load unconditional (a thread with no entry in a piece re-reads entry 0 of the arrays; the line loop knows
        // which threads count): a select on the loaded value would make each load wait for itself
        // ... but a WAVE none of whose threads has an entry in a piece (at C3 a piece is ~110 entries: the third and fourth
        // wave, and every wave for the slots past the column's last line) skips the pair of loads: a scalar branch
        int32_t P[AW_LINES], C[AW_LINES];
        const int wbase = tid & ~(WAVE - 1);
#pragma unroll
        for (int i = 0; i < AW_LINES; i++) {
            const int ii = i < nlc ? i : 0;
            const int32_t lo_i = __builtin_amdgcn_readfirstlane(s_lo[ii]), hi_i = __builtin_amdgcn_readfirstlane(s_hi[ii]);
            P[i] = 0;
            C[i] = 0;
            if (i < nlc && lo_i + wbase < hi_i) {
                const int32_t e = lo_i + tid;
                const int64_t at = e < hi_i ? s_b[ii] + e : 0;
                P[i] = pid[at];
                C[i] = cov[at];
            }
        }
#pragma unroll
        for (int i = 0; i < AW_LINES; i++) {
            if (i < nlc) {   // uniform
                const int fl = __builtin_amdgcn_readfirstlane(s_fl[i]);
                const double wq = s_w[i];
                const int64_t b = s_b[i];
                if (fl == 0) {
                    const int32_t lo = __builtin_amdgcn_readfirstlane(s_lo[i]), hi = __builtin_amdgcn_readfirstlane(s_hi[i]);
                    if (lo + tid < hi) rmw(P[i], C[i], wq);
                    for (int32_t e = lo + AW_THREADS + tid; e < hi; e += AW_THREADS)   // a piece longer than the workgroup
                        rmw(pid[b + e], cov[b + e], wq);
                } else {
                    const int32_t len = __builtin_amdgcn_readfirstlane(s_len[i]);
                    if (fl == 1) {
                        for (int32_t e = tid; e < len; e += AW_THREADS) {
                            const int32_t p = pid[b + e];
                            if ((uint32_t)(p - p_lo) < span) rmw(p, cov[b + e], wq);
                        }
                    } else if (tid == 0) {
                        for (int32_t e = 0; e < len; e++) {
                            const int32_t p = pid[b + e];
                            if ((uint32_t)(p - p_lo) < span) rmw(p, cov[b + e], wq);
                        }
                    }
                }
                __syncthreads();   // the next line of this column may touch the same cells from other threads
            }
        }
    }
    // the cell's fp64 sum is complete: the one rounding to fp32 (add_item's cast, morna.py:405-424) happens here
    float *out = colacc + (int64_t)c * n_items + p_lo;
    for (int i = tid; i < (int)span; i += AW_THREADS) out[i] = __double2float_rn(acc[i]);
}

// ------------------------------------------------------------ accumulate pass

#define ACC_TILE 8192     // samples per tile: 64 KiB of fp64 accumulators in LDS
#define ACC_LINES 256     // lines whose extents are staged per pass
#define ACC_PF 2          // entries per thread prefetched from a coming line
#define ACC_PFD 4         // lines ahead (8: over 64 VGPRs, one 1024-thread workgroup per CU, 2.6 ms instead of 2.0)

__global__ __launch_bounds__(ACC_THREADS) void accumulate_kernel(
    int32_t n_tiles, int32_t n_cols, const int32_t *__restrict__ col_off, const int32_t *__restrict__ col_lines, const double *__restrict__ sidf,
    const uint8_t *__restrict__ flags, const int64_t *__restrict__ row_ptr, const int32_t *__restrict__ ids,
    const int32_t *__restrict__ cov, int64_t n_items, float *__restrict__ colacc /* [D][n_items] */)
{
    __shared__ double acc[ACC_TILE];
    __shared__ int64_t s_b[ACC_LINES], s_e[ACC_LINES];
    __shared__ double s_w[ACC_LINES];
    __shared__ uint8_t s_f[ACC_LINES];
    // block b = ((column / 8) * n_tiles + tile) * 8 + column % 8
    const int c = (int)(blockIdx.x >> 3) / n_tiles * 8 + (int)(blockIdx.x & 7);
    const int tile = (int)(blockIdx.x >> 3) % n_tiles;
    if (c >= n_cols) return;
    const int64_t r_lo = (int64_t)tile * ACC_TILE;
    const int64_t r_hi = r_lo + ACC_TILE < n_items ? r_lo + ACC_TILE : n_items;
    const uint32_t span = (uint32_t)(r_hi - r_lo);
    const int tid = threadIdx.x;
    for (int i = tid; i < ACC_TILE; i += ACC_THREADS) acc[i] = 0.0;
    const int nl = col_off[c + 1] - col_off[c];
    const int32_t *lines = col_lines + col_off[c];

    for (int l0 = 0; l0 < nl; l0 += ACC_LINES) {
        const int nb = nl - l0 < ACC_LINES ? nl - l0 : ACC_LINES;
        if (tid < nb) {
            const int32_t j = lines[l0 + tid];
            s_b[tid] = row_ptr[j]; s_e[tid] = row_ptr[j + 1]; s_w[tid] = sidf[j]; s_f[tid] = flags[j];
        }
        __syncthreads();
        // The first ACC_PF * ACC_THREADS entries of a line are fetched into registers ACC_PFD lines ahead (a ring of
        // register sets, the line loop unrolled ACC_PFD times): with one line of cover, every line step lasted one
        // memory round trip (1.4 us at C3) whatever its own work.
        int32_t idr[ACC_PFD][ACC_PF], cvr[ACC_PFD][ACC_PF];
        auto fetch = [&](int i, int32_t(&idd)[ACC_PF], int32_t(&cvd)[ACC_PF]) {
            const bool live = i < nb;
            const int64_t b = live ? s_b[i] : 0, e = live ? s_e[i] : 0;
#pragma unroll
            for (int m = 0; m < ACC_PF; m++) {
                const int64_t t = b + tid + (int64_t)m * ACC_THREADS;
                idd[m] = t < e ? ids[t] : -1;
                cvd[m] = t < e ? cov[t] : 0;
            }
        };
#pragma unroll
        for (int d = 0; d < ACC_PFD; d++) fetch(d, idr[d], cvr[d]);
        for (int i0 = 0; i0 < nb; i0 += ACC_PFD) {
#pragma unroll
            for (int d = 0; d < ACC_PFD; d++) {
                const int i = i0 + d;
                if (i < nb) {   // uniform over the workgroup
                    const int64_t b = s_b[i], e = s_e[i];
                    const double wgt = s_w[i];
                    if (s_f[i]) {
                        // a sample repeats inside this line: keep the line's own order
                        if (tid == 0)
                            for (int64_t t = b; t < e; t++) {
                                const uint32_t off = (uint32_t)((int64_t)ids[t] - r_lo);
                                if (off < span) acc[off] = __dadd_rn(acc[off], __dmul_rn((double)cov[t], wgt));
                            }
                    } else {
                        // every sample at most once: the lanes touch distinct cells of the tile
#pragma unroll
                        for (int m = 0; m < ACC_PF; m++) {
                            const uint32_t off = (uint32_t)((int64_t)idr[d][m] - r_lo);   // -1 (no entry) is out of range
                            // tf_idf = cov * idf (one rounding), then += (one rounding): morna.py:384-388
                            if (idr[d][m] >= 0 && off < span) acc[off] = __dadd_rn(acc[off], __dmul_rn((double)cvr[d][m], wgt));
                        }
                        for (int64_t t = b + tid + (int64_t)ACC_PF * ACC_THREADS; t < e; t += ACC_THREADS) {
                            const uint32_t off = (uint32_t)((int64_t)ids[t] - r_lo);
                            if (off < span) acc[off] = __dadd_rn(acc[off], __dmul_rn((double)cov[t], wgt));
                        }
                    }
                    fetch(i + ACC_PFD, idr[d], cvr[d]);   // this register set is free: the line ACC_PFD ahead
                    __syncthreads();   // the next line of this column may touch the same cells
                }
            }
        }
    }
    // the cell's fp64 sum is complete: the one rounding to fp32 (add_item's cast, morna.py:405-424) happens here, and
    // the column image that transpose_kernel turns into rows is fp32 (half the bytes written and read again)
    float *out = colacc + (int64_t)c * n_items + r_lo;
    for (int i = tid; i < (int)span; i += ACC_THREADS) out[i] = __double2float_rn(acc[i]);
}

// --------------------------------------------------- fp32 [D][N] -> fp32 [N][dpad]

#define TT 64
// item_of: the item whose value stands at position n of a column image (null: the identity)
__global__ __launch_bounds__(256) void transpose_convert_kernel(const float *__restrict__ colacc,
                                                                int64_t n_items, int32_t dim, int32_t dpad,
                                                                const int32_t *__restrict__ item_of, float *__restrict__ X)
{
    __shared__ float tile[TT][TT + 1];
    __shared__ int32_t s_item[TT];
    const int64_t n0 = (int64_t)blockIdx.x * TT;
    if (threadIdx.x < TT) {
        const int64_t n = n0 + threadIdx.x;
        s_item[threadIdx.x] = n < n_items ? (item_of ? item_of[n] : (int32_t)n) : 0;
    }
    const int32_t c0 = blockIdx.y * TT;
    const int tx = threadIdx.x & (TT - 1), ty = threadIdx.x / TT;   // 64 x 4
    for (int r = ty; r < TT; r += 4) {
        int32_t c = c0 + r;
        int64_t n = n0 + tx;
        float v = 0.f;
        if (c < dim && n < n_items) v = colacc[(int64_t)c * n_items + n];   // already rounded to fp32 by accumulate_kernel
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < TT; r += 4) {
        int64_t n = n0 + r;
        int32_t c = c0 + tx;
        if (n < n_items && c < dpad) X[(int64_t)s_item[r] * dpad + c] = tile[tx][r];
    }
}

// ------------------------------- fp32 [D][N] -> fp32 [N][dpad] AND the rows' canonical norms, in one pass
//
// One workgroup = 64 positions of the column image, ALL column tiles in ascending order: the cells pass through the LDS
// tile on their way to the rows (as above), and on the way each thread squares what it holds into the chains of the
// canonical dot(x, x) -- lane l, component c of wave_dot owns elements 256 k + 4 l + c, one fmaf chain over k: column
// tile j covers k = j / 4 and lanes 16 (j % 4) .. +15, and thread (tx, ty) reads cells of ONE component c = ty % 4 of row
// n0 + tx, so it keeps 32 chains (4 quarters x 8 lanes) that receive their terms in ascending k as j ascends.  At the end
// the four components of a lane meet in LDS, are folded (a0 + a1) + (a2 + a3), and the 64 lane values are added in the butterfly's
// order (32, 16, 8, 4, 2, 1): bit for bit what row_norms_kernel computes from the finished row, without reading the
// 0.6 GB of rows again.
#define TN_TY 4   // waves per workgroup: thread (tx, ty) handles tile rows ty, ty + 4, ... (16 per tile, 64 chains); 8 waves of 8 cells: 1.38 ms against 1.28 for the feature group at C3
__global__ __launch_bounds__(TT * TN_TY) void transpose_norms_kernel(const float *__restrict__ colacc, int64_t n_items, int32_t dim,
                                                                     int32_t dpad, const int32_t *__restrict__ item_of,
                                                                     float *__restrict__ X, float *__restrict__ norm2,
                                                                     RowInfo *__restrict__ info)
{
    constexpr int NI = TT / TN_TY;        // cells of a tile per thread
    __shared__ float tile[TT][TT + 1];
    __shared__ float s_ch[4][TT][17];     // one quarter's chains: [component][row][lane of the quarter]
    __shared__ float s_f[TT][TT + 1];     // folded lane values of every row
    __shared__ float s_mn[TN_TY][TT];
    __shared__ int32_t s_item[TT];
    const int64_t n0 = (int64_t)blockIdx.x * TT;
    const int tx = threadIdx.x & (TT - 1);
    const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x / TT);   // a wave = one ty (the column addresses below are scalar)
    if (threadIdx.x < TT) {
        const int64_t n = n0 + threadIdx.x;
        s_item[threadIdx.x] = n < n_items ? (item_of ? item_of[n] : (int32_t)n) : 0;
    }
    // tile row r = ty + TN_TY i is column 64 j + r: component c = r % 4 = ty % 4 (the same for all of the thread's cells),
    // lane 16 (j % 4) + r / 4 = 16 (j % 4) + (ty >> 2) + (TN_TY / 4) i
    float ch[4][NI];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < NI; i++) ch[q][i] = 0.f;
    float mn = INFINITY;   // smallest non-zero |x| this thread has seen
    const int64_t n = n0 + tx;
    const int n_tiles = dpad / TT;
    const uint32_t n_off = (uint32_t)(n < n_items ? n : n_items - 1);   // (a position past the end re-reads the last one; nothing of it is stored)
    float cur[NI], nxt[NI];
    auto fetch = [&](int j, float(&v)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int32_t c = j * TT + ty + TN_TY * i;                  // uniform over the wave
            const float *colp = colacc + (int64_t)(c < dim ? c : 0) * n_items;
            const float x = colp[n_off];
            v[i] = c < dim ? x : 0.f;
        }
    };
    fetch(0, cur);
    for (int j0 = 0; j0 < n_tiles; j0 += 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) {   // dpad is a multiple of 256: four tiles per k
            const int j = j0 + q;
            if (j + 1 < n_tiles) fetch(j + 1, nxt);
            else {
#pragma unroll
                for (int i = 0; i < NI; i++) nxt[i] = 0.f;
            }
            __syncthreads();   // the previous tile has been written out
#pragma unroll
            for (int i = 0; i < NI; i++) {
                tile[ty + TN_TY * i][tx] = cur[i];
                ch[q][i] = fmaf(cur[i], cur[i], ch[q][i]);
                mn = fminf(mn, cur[i] != 0.f ? fabsf(cur[i]) : INFINITY);
            }
            __syncthreads();
            const int32_t c0 = j * TT;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const int r = ty + TN_TY * i;   // uniform over the wave: the row's base address is scalar
                float *rowp = X + (int64_t)__builtin_amdgcn_readfirstlane(s_item[r]) * dpad + c0;
                if (n0 + r < n_items) rowp[tx] = tile[tx][r];
            }
#pragma unroll
            for (int i = 0; i < NI; i++) cur[i] = nxt[i];
        }
    }
    // fold: lane l of row tx = (c0 + c1) + (c2 + c3); component c of the quarter's 16 lanes sits in the threads with ty % 4 = c
#pragma unroll
    for (int q = 0; q < 4; q++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; i++) s_ch[ty & 3][tx][(ty >> 2) + (TN_TY / 4) * i] = ch[q][i];   // lane of the quarter = tile row / 4
        __syncthreads();
        // thread (tx, ty) folds 16 / TN_TY lanes of this quarter
#pragma unroll
        for (int u = 0; u < 16 / TN_TY; u++) {
            const int i = (16 / TN_TY) * ty + u;
            s_f[tx][16 * q + i] = (s_ch[0][tx][i] + s_ch[1][tx][i]) + (s_ch[2][tx][i] + s_ch[3][tx][i]);
        }
    }
    s_mn[ty][tx] = mn;
    __syncthreads();
    if (ty == 0 && n < n_items) {
        // the butterfly as lane 0 sees it: at stage s lane l < s adds lane l + s (in place, in the row's own LDS line)
#pragma unroll 1
        for (int st = 32; st >= 1; st >>= 1)
#pragma unroll 1
            for (int l = 0; l < st; l++) s_f[tx][l] = s_f[tx][l] + s_f[tx][l + st];
        const float d = s_f[tx][0];
        float m = s_mn[0][tx];
#pragma unroll
        for (int y = 1; y < TN_TY; y++) m = fminf(m, s_mn[y][tx]);
        const int32_t item = s_item[tx];
        norm2[item] = d;
        const float norm = sqrtf(d);
        const bool sub = (double)m < (double)norm * 0x1p-125;   // as row_norms_kernel
        RowInfo ri;
        ri.norm2 = d;
        ri.norm = sub ? -norm : norm;
        ri.rnorm = 1.0 / (double)norm;
        info[item] = ri;
    }
}

// ------------------------------------------------------------------ row norms

__global__ __launch_bounds__(256) void row_norms_kernel(const float *__restrict__ X, int64_t n_items,
                                                        int32_t dpad, float *__restrict__ norm2,
                                                        RowInfo *__restrict__ info)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) / WAVE;
    const int nvec = dpad / 4;
    for (int64_t r = wave; r < n_items; r += nwaves) {
        const float4 *x = (const float4 *)(X + r * dpad);
        // wave_dot(x, x) with the row read once, and the smallest non-zero |x_i| taken on the way
        Acc4 s = acc4_zero();
        float mn = INFINITY;
        auto low = [&](const float4 &v) {
            mn = fminf(mn, v.x != 0.f ? fabsf(v.x) : INFINITY);
            mn = fminf(mn, v.y != 0.f ? fabsf(v.y) : INFINITY);
            mn = fminf(mn, v.z != 0.f ? fabsf(v.z) : INFINITY);
            mn = fminf(mn, v.w != 0.f ? fabsf(v.w) : INFINITY);
        };
        int i = lane;
        for (; i + 3 * WAVE < nvec; i += 4 * WAVE) {
            const float4 v0 = x[i], v1 = x[i + WAVE], v2 = x[i + 2 * WAVE], v3 = x[i + 3 * WAVE];
            fma4(s, v0, v0);
            fma4(s, v1, v1);
            fma4(s, v2, v2);
            fma4(s, v3, v3);
            low(v0);
            low(v1);
            low(v2);
            low(v3);
        }
        for (; i < nvec; i += WAVE) {
            const float4 v = x[i];
            fma4(s, v, v);
            low(v);
        }
        const float d = acc4_finish(s);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mn = fminf(mn, __shfl_xor(mn, off, WAVE));
        if (lane == 0) {
            norm2[r] = d;
            // what every two_means step on this row would compute from norm2 (forest.hip), computed here once
            const float norm = sqrtf(d);
            // x_i / norm rounds to a subnormal only if |x_i| < norm * 2^-126 * (1 + 2^-24); a factor 2 of slack
            const bool sub = (double)mn < (double)norm * 0x1p-125;
            RowInfo ri;
            ri.norm2 = d;
            ri.norm = sub ? -norm : norm;
            ri.rnorm = 1.0 / (double)norm;
            info[r] = ri;
        }
    }
}

// ------------------------------------------------------------------ host side

int compute_norms(morna_index *h)
{
    MORNA_TRY(h->norm2.alloc((size_t)h->n_items));
    MORNA_TRY(h->rowinfo.alloc((size_t)h->n_items));
    if (h->n_items > 0) {
        int64_t waves = h->n_items;
        int blocks = (int)std::min<int64_t>((waves + 3) / 4, 256 * 16);
        hipLaunchKernelGGL(row_norms_kernel, dim3(blocks), dim3(256), 0, h->stream, h->X.p, h->n_items,
                           h->dpad, h->norm2.p, h->rowinfo.p);
        HIP_TRY(hipGetLastError());
    }
    h->norms_valid = true;
    h->half_valid = false;   // the rows changed: their fp16 image (splitmm.hip) is made again when next needed
    return MORNA_OK;
}

int upload_host_rows(morna_index *h)
{
    if (!h->host_dirty) return MORNA_OK;
    const int64_t n = h->host_n;
    MORNA_TRY(h->X.alloc((size_t)std::max<int64_t>(n, 1) * h->dpad));
    HIP_TRY(hipMemsetAsync(h->X.p, 0, (size_t)std::max<int64_t>(n, 1) * h->dpad * sizeof(float), h->stream));
    if (n > 0)
        HIP_TRY(hipMemcpy2DAsync(h->X.p, (size_t)h->dpad * sizeof(float), h->host_rows.data(),
                                 (size_t)h->dim * sizeof(float), (size_t)h->dim * sizeof(float), (size_t)n,
                                 hipMemcpyHostToDevice, h->stream));
    if (h->n_items != n) h->comm_sizes_valid = false;   // (comm.hip: the shard offsets follow the row count)
    h->n_items = n;
    h->host_dirty = false;
    h->built = false;
    MORNA_TRY(compute_norms(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MORNA_OK;
}

int hash_keys_device(morna_index *h, const uint8_t *key_bytes, const int64_t *key_off, int64_t J,
                     int32_t *hash_out, int32_t *col_out, int32_t *sign_out)
{
    if (J <= 0) return MORNA_OK;
    const int64_t nbytes = key_off[J];
    DevBuf<uint8_t> dk;
    DevBuf<int64_t> doff;
    DevBuf<int32_t> dh, dc, ds;
    MORNA_TRY(dk.alloc((size_t)nbytes));
    MORNA_TRY(doff.alloc((size_t)J + 1));
    MORNA_TRY(dh.alloc((size_t)J));
    MORNA_TRY(dc.alloc((size_t)J));
    MORNA_TRY(ds.alloc((size_t)J));
    HIP_TRY(hipMemcpyAsync(dk.p, key_bytes, (size_t)nbytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(doff.p, key_off, (size_t)(J + 1) * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(hash_keys_kernel, dim3((unsigned)((J + 255) / 256)), dim3(256), 0, h->stream, dk.p, doff.p,
                       J, h->dim, (const double *)nullptr, dc.p, (double *)nullptr, dh.p, ds.p, (int32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    if (hash_out) HIP_TRY(hipMemcpyAsync(hash_out, dh.p, (size_t)J * 4, hipMemcpyDeviceToHost, h->stream));
    if (col_out) HIP_TRY(hipMemcpyAsync(col_out, dc.p, (size_t)J * 4, hipMemcpyDeviceToHost, h->stream));
    if (sign_out) HIP_TRY(hipMemcpyAsync(sign_out, ds.p, (size_t)J * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MORNA_OK;
}

int build_features(morna_index *h, int64_t n_items)
{
    if (!h->staged) {
        set_error("build_features: no junction lines staged");
        return MORNA_E_STATE;
    }
    if (n_items <= 0) {
        set_error("no internal ids were assigned: no sample passes the sample threshold");
        return MORNA_E_EMPTY;
    }
    const int64_t J = h->J;
    const int32_t D = h->dim;
    if (J > INT32_MAX) {
        set_error("build_features: %lld junction lines exceed the 2^31 limit", (long long)J);
        return MORNA_E_INVALID;
    }
    // scratch lives in the handle (released by morna_unstage_junctions): a rebuild does no hipMalloc
    ScratchRef<int32_t> col(h->scratch[0]), col_count(h->scratch[1]), col_off(h->scratch[2]), col_lines(h->scratch[3]);
    ScratchRef<double> sidf(h->scratch[4]);
    ScratchRef<float> colacc(h->scratch[5]);
    ScratchRef<uint8_t> flags(h->scratch[6]);
    ScratchRef<int32_t> bucket_aux(h->scratch[7]);   // [J] rank of a line in its chunk, then [n_chunks][D] chunk counts / bases
    // with an item order whose length matches: the wave-per-tile form (each entry of the nnz stream read once)
    static const bool wave_on = !(getenv("MORNA_FEATURES_WAVE") && atoi(getenv("MORNA_FEATURES_WAVE")) == 0);
    const bool by_order = wave_on && h->order_n == n_items && J > 0 && h->nnz > 0;   // (the kernels read entry 0 as a dummy)
    constexpr int AW_TILE_SHIFT = 12, AW_TILE = 1 << AW_TILE_SHIFT;   // 4096 positions: 0.61 ms at C3 (2048: 0.76, 8192: 0.83)
    const int32_t aw_tiles = (int32_t)((n_items + AW_TILE - 1) / AW_TILE);
    ScratchRef<int32_t> tile_off(h->scratch[24]);        // [J][aw_tiles + 1] where each tile's piece of a line begins
    ScratchRef<int32_t> pid(h->scratch[25]);             // [nnz] position in the order of every entry's item
    if (by_order) MORNA_TRY(tile_off.alloc((size_t)J * (size_t)(aw_tiles + 1)));
    if (by_order) MORNA_TRY(pid.alloc((size_t)h->nnz));
    // algorithmic bytes of this pass: 8*nnz + keys + 8*J + 4*N*D
    const int64_t alg_bytes = 8 * h->nnz + h->key_bytes_n + 8 * J + 4 * n_items * (int64_t)D;
    MORNA_TRY(col.alloc((size_t)J));
    MORNA_TRY(sidf.alloc((size_t)J));
    MORNA_TRY(flags.alloc((size_t)J));
    MORNA_TRY(col_count.alloc((size_t)D));
    MORNA_TRY(col_off.alloc((size_t)D + 1));
    MORNA_TRY(col_lines.alloc((size_t)J));
    const int64_t n_line_chunks = (J + CR_LINES - 1) / CR_LINES;
    MORNA_TRY(bucket_aux.alloc((size_t)J + (size_t)n_line_chunks * (size_t)D));
    int32_t *const line_rank = bucket_aux.p, *const chunk_cnt = bucket_aux.p + J;
    MORNA_TRY(colacc.alloc((size_t)D * (size_t)n_items));
    MORNA_TRY(h->X.alloc((size_t)n_items * h->dpad));
    {
        ScopedTimer tm(h, MORNA_T_FEATURES, alg_bytes);
        hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, h->stream, col_count.p, (int64_t)D);
        if (J > 0) {
            hipLaunchKernelGGL(hash_keys_kernel, dim3((unsigned)((J + 255) / 256)), dim3(256), 0, h->stream,
                               h->s_keys.p, h->s_key_off.p, J, D, h->s_idf.p, col.p, sidf.p, (int32_t *)nullptr,
                               (int32_t *)nullptr, col_count.p);
            // the duplicate flags of the lines depend only on the staged ids: side stream, beside the hashing and
            // the bucketing of the lines by column
            HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
            HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
            const int64_t n_words = (n_items + 31) / 32;
            const int32_t upgrade = by_order ? 1 : 0;
            if (by_order) {
                // ascending check + tile extents of every line; the flag kernels below then look at the lines that do
                // not ascend only (none, for a file whose lines are sorted and an order that says so)
                static const bool lds_rank = !(getenv("MORNA_PREP_LDS") && atoi(getenv("MORNA_PREP_LDS")) == 0);
                if (lds_rank && n_items <= 65536) {
                    const size_t lds = ((size_t)n_items * 2 + 15) / 16 * 16;
                    HIP_TRY(hipFuncSetAttribute((const void *)line_prep_lds_kernel<AW_TILE_SHIFT>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    const int lp_blocks = (int)std::max<int64_t>(1, std::min<int64_t>((J + 11) / 12, (int64_t)h->n_cus));
                    hipLaunchKernelGGL(line_prep_lds_kernel<AW_TILE_SHIFT>, dim3(lp_blocks), dim3(LPL_THREADS), lds, h->stream2,
                                       h->s_row_ptr.p, h->s_ids.p, h->item_rank.p, (int32_t)n_items, J, aw_tiles, tile_off.p, pid.p,
                                       flags.p);
                } else {
                    const int lp_blocks = (int)std::max<int64_t>(1, std::min<int64_t>((J + 3) / 4, (int64_t)h->n_cus * 8));
                    hipLaunchKernelGGL(line_prep_kernel<AW_TILE_SHIFT>, dim3(lp_blocks), dim3(256), 0, h->stream2, h->s_row_ptr.p,
                                       h->s_ids.p, h->item_rank.p, (int32_t)n_items, J, aw_tiles, tile_off.p, pid.p, flags.p);
                }
            }
            if (n_words <= LFW_MAX_WORDS) {
                const size_t lds = (size_t)LFW_WAVES * (size_t)n_words * 4;
                HIP_TRY(hipFuncSetAttribute((const void *)line_flags_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                const int64_t per_wg = (J + LFW_WAVES - 1) / LFW_WAVES;
                const int fl_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(per_wg, (int64_t)h->n_cus * 2));
                hipLaunchKernelGGL(line_flags_wave_kernel, dim3(fl_blocks), dim3(LFW_WAVES * WAVE), lds, h->stream2,
                                   h->s_row_ptr.p, h->s_ids.p, J, (int32_t)n_words, flags.p, upgrade);
            } else if (n_words <= FLAG_MAX_WORDS) {
                const int fl_blocks = (int)std::min<int64_t>(J, 256 * 4);
                hipLaunchKernelGGL(line_flags_kernel, dim3(fl_blocks), dim3(LF_THREADS), (size_t)n_words * 4, h->stream2,
                                   h->s_row_ptr.p, h->s_ids.p, J, (int32_t)n_words, flags.p, upgrade);
            } else {
                // the sample bitmap does not fit LDS: the order-preserving serial path for every line (by_order: for
                // every line that does not ascend -- flag 1 -> 2; ascending lines need no such test)
                if (by_order)
                    hipLaunchKernelGGL(flags_to_serial_kernel, dim3((unsigned)((J + 255) / 256)), dim3(256), 0, h->stream2, flags.p, J);
                else
                    HIP_TRY(hipMemsetAsync(flags.p, 1, (size_t)J, h->stream2));
            }
            HIP_TRY(hipEventRecord(h->ev_join, h->stream2));
        }
        hipLaunchKernelGGL(col_scan_kernel, dim3(1), dim3(1024), 0, h->stream, col_count.p, D, col_off.p);
        if (J > 0) {
            const int32_t n_chunks = (int32_t)((J + CR_LINES - 1) / CR_LINES);
            hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)n_chunks * D + 255) / 256, 2048)), dim3(256), 0,
                               h->stream, chunk_cnt, (int64_t)n_chunks * D);
            hipLaunchKernelGGL(col_rank_kernel, dim3((unsigned)n_chunks), dim3(CR_LINES), 0, h->stream, col.p, J, n_chunks,
                               line_rank, chunk_cnt);
            hipLaunchKernelGGL(col_base_kernel, dim3((unsigned)((D + 256 / WAVE - 1) / (256 / WAVE))), dim3(256), 0, h->stream,
                               chunk_cnt, n_chunks, (int32_t)D);
            hipLaunchKernelGGL(col_place_kernel, dim3((unsigned)((J + 255) / 256)), dim3(256), 0, h->stream, col.p, J, n_chunks,
                               line_rank, chunk_cnt, col_off.p, col_lines.p);
        }
        if (J > 0) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        if (by_order) {
            hipLaunchKernelGGL(accumulate_piece_kernel<AW_TILE>, dim3((unsigned)D * (unsigned)aw_tiles), dim3(AW_THREADS), 0, h->stream, aw_tiles,
                               (int32_t)D, col_off.p, col_lines.p, sidf.p, flags.p, h->s_row_ptr.p, tile_off.p, pid.p,
                               h->s_cov.p, n_items, colacc.p, 0);
        } else {
            const unsigned tiles = (unsigned)((n_items + ACC_TILE - 1) / ACC_TILE);
            // workgroup b runs on XCD b % 8: the sample tiles of one column are dealt to ONE XCD, back to back, so the
            // column's lines come from HBM once and from that XCD's L2 for the other tiles
            hipLaunchKernelGGL(accumulate_kernel, dim3(8u * tiles * (unsigned)((D + 7) / 8)), dim3(ACC_THREADS), 0, h->stream,
                               (int32_t)tiles, (int32_t)D, col_off.p, col_lines.p, sidf.p, flags.p, h->s_row_ptr.p, h->s_ids.p,
                               h->s_cov.p, n_items, colacc.p);
        }
        // the column image becomes the rows, and the rows' canonical norms come out of the same pass (MORNA_FUSED_NORMS=0:
        // transpose, then row_norms_kernel over the finished rows)
        // (the fused pass is one workgroup per 64 positions: below one workgroup per CU -- a 6250-row shard of the 8-way cut --
        // the 2-D transpose and a norm pass of their own are faster: 0.34 -> 0.29 ms for that shard's feature build)
        static const int fused_env = getenv("MORNA_FUSED_NORMS") ? atoi(getenv("MORNA_FUSED_NORMS")) : -1;
        const bool fused = fused_env >= 0 ? fused_env != 0 : (n_items + TT - 1) / TT >= (int64_t)h->n_cus;
        const int32_t *item_of = by_order ? (const int32_t *)h->item_at.p : (const int32_t *)nullptr;
        if (fused) {
            MORNA_TRY(h->norm2.alloc((size_t)n_items));
            MORNA_TRY(h->rowinfo.alloc((size_t)n_items));
            hipLaunchKernelGGL(transpose_norms_kernel, dim3((unsigned)((n_items + TT - 1) / TT)), dim3(TT * TN_TY), 0, h->stream, colacc.p,
                               n_items, D, h->dpad, item_of, h->X.p, h->norm2.p, h->rowinfo.p);
        } else {
            dim3 tg((unsigned)((n_items + TT - 1) / TT), (unsigned)((h->dpad + TT - 1) / TT));
            hipLaunchKernelGGL(transpose_convert_kernel, tg, dim3(256), 0, h->stream, colacc.p, n_items, D, h->dpad, item_of, h->X.p);
        }
        HIP_TRY(hipGetLastError());
        if (h->n_items != n_items) h->comm_sizes_valid = false;   // (comm.hip: the shard offsets follow the row count)
        h->n_items = n_items;
        h->host_n = 0;
        h->host_rows.clear();
        h->host_dirty = false;
        h->built = false;
        if (fused) {
            h->norms_valid = true;
            h->half_valid = false;   // the rows changed: their fp16 image (splitmm.hip) is made again when next needed
        } else {
            MORNA_TRY(compute_norms(h));
        }
    }
    // No synchronisation here: everything above is ordered on the handle's stream and reads no caller memory, and
    // whatever the caller does next with the handle (build, queries, get_items) is ordered behind it or synchronises
    // itself -- a forest build that follows starts without the device draining first.  Blocking copies settle() first.
    h->unsettled = true;
    return MORNA_OK;
}

}  // namespace morna
