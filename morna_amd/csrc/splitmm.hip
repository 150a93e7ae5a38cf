// splitmm.hip -- the split of a whole forest level as ONE contraction on the matrix cores.
//
// Counterpart of the inner loop of annoy's _make_tree (called from AnnoyIndex.build,
// commanderson/morna morna.py:425): every row of a split node gets side = sign(dot(row, hyperplane)).
// Over a level that is C[row][node] = sum_d X[row][d] H[node][d] for all rows and all split nodes of
// all trees; the entry a row needs is the one of ITS node in each tree.  While a tree has few split
// nodes, computing the whole product on v_mfma_f32_32x32x16_f16 is far cheaper than streaming each
// row once per tree through the vector ALUs (C3, fourth level: 2.5e11 MAC against 120 GB of L2 -> CU
// traffic).
//
// The product is taken on fp16 COPIES of the rows and hyperplanes (each scaled by a power of two so
// that its largest element lies in [2^14, 2^15)), and it only FILTERS.  With y = fp16(s x) = s x - d and
// g = fp16(t h) = t h - f (d, f: the rounding errors, whose Euclidean norms are MEASURED when the copies are made:
// about 2^-11 / sqrt(3) of |y|, not the worst case 2^-11),
//   |sum y_i g_i (as computed) - s t wave_dot(x, h)|  <=  (EACC |y| + |d|) |g| + (|y| + |d|) |f|,
//   EACC = 2 * dpad * 2^-24 (fp32 accumulation in any order, with room for adders that truncate or align to the
//          largest addend: the even and the odd K-steps go to two accumulators of dpad / 2 products each,
//          4 * 2^-24 per product, added once at the end) + the rounding of the canonical fp32 dot itself (< 4e-6),
// so whenever |C| exceeds that bound the sign of C IS the sign of the canonical wave_dot the split is defined
// by.  The (row, node) pairs the filter cannot decide -- about 0.5 % -- are listed and recomputed by
// split_amb_kernel with wave_dot on the fp32 data, including the dot == 0 coin flip.  The sides written
// are therefore exactly those of split_kernel; the forest stays bit-identical to the oracle.
#include <cstdio>
#include <cstdlib>

#include "common.hpp"
#include "devutil.hpp"
#include "mm16.hpp"

namespace morna {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline float sm_eps(int32_t dpad) { return mm16_eacc(dpad); }   // EACC

// ---- fp16 image of a row-major fp32 matrix: one wave per row ---------------------------------------
// dst row = fp16(2^e * src row) with max |2^e x_i| in [2^14, 2^15); norm[row] = an upper bound of the
// Euclidean norm of the fp16 row (+inf when the row cannot be scaled into range: it is then never
// decided by the filter).
__global__ __launch_bounds__(256) void rows_to_half_kernel(const float *__restrict__ src, int64_t rows, int32_t dpad,
                                                           _Float16 *__restrict__ dst, float *__restrict__ norm,
                                                           float *__restrict__ err /* |s x - y| per row (upper bound) */,
                                                           float *__restrict__ inv_scale /* 2^-e per row, or null */,
                                                           unsigned int *__restrict__ zero_me /* a counter to reset, or null */)
{
    if (zero_me && blockIdx.x == 0 && threadIdx.x == 0) *zero_me = 0u;
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t r = (int64_t)blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
    if (r >= rows) return;
    const float4 *x = (const float4 *)(src + r * dpad);
    const int nvec = dpad / 4;
    float m = 0.f;
    bool bad = false;
    for (int i = lane; i < nvec; i += WAVE) {
        const float4 v = x[i];
        const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        bad |= !(a <= 3.0e38f);   // inf or NaN
        m = fmaxf(m, a);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, WAVE));
    bad = __any(bad);
    int e = 0;
    if (m > 0.f) e = 14 - ilogbf(m);
    if (e > 126 || e < -126) bad = true;   // the scale itself must be a normal float
    const float s = bad ? 0.f : ldexpf(1.f, e);
    float sum = 0.f, sume = 0.f;
    _Float16 *d = dst + r * dpad;
    for (int i = lane; i < nvec; i += WAVE) {
        const float4 v = x[i];
        const _Float16 h0 = (_Float16)(v.x * s), h1 = (_Float16)(v.y * s), h2 = (_Float16)(v.z * s), h3 = (_Float16)(v.w * s);
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        *(f16x4 *)(d + 4 * i) = (f16x4){h0, h1, h2, h3};
        const float f0 = (float)h0, f1 = (float)h1, f2 = (float)h2, f3 = (float)h3;
        sum += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
        // the rounding error of each element, exactly: s x_i is exact (power of two), y_i has 11 bits of it
        const float e0 = v.x * s - f0, e1 = v.y * s - f1, e2 = v.z * s - f2, e3 = v.w * s - f3;
        sume += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
    }
    sum = wave_sum_xor(sum);
    sume = wave_sum_xor(sume);
    // fp32 sum of <= 8192 squares: relative error < 1e-3; the bound is widened by that much
    if (lane == 0) {
        norm[r] = bad ? INFINITY : sqrtf(sum) * 1.002f;
        err[r] = bad ? 0.f : sqrtf(sume) * 1.002f + 1e-30f;   // (elements of s x below the fp32 normal range: < 1e-38 each)
        if (inv_scale) inv_scale[r] = bad ? 0.f : ldexpf(1.f, -e);
    }
}

// ---- the contraction with the side decision fused into its epilogue --------------------------------
#define SM_OPEN 1408             // open pairs a tile keeps in LDS (11 KB; ~500 expected of 16384 at the root level; the 128 x 128 form must stay
                                 // within 16 KB of static LDS beside its 64 KB of slabs for two workgroups to share a CU)

// BIG = false: 128 rows x 128 hyperplanes per workgroup, 8 waves, two accumulation chains; two workgroups per CU.
// BIG = true : 256 x 256, 16 waves, one chain (mm16.hpp): half the operand bytes delivered to LDS per product -- the
//              loop is bound by those deliveries -- for levels with enough hyperplane tiles to fill the chip evenly.
//              Its bound (EACC for one chain of dpad products) is wider, so it leaves more pairs open; the sides written
//              are the same either way.
template <bool BIG>
__global__ __launch_bounds__(BIG ? 1024 : 512) void split_mm_kernel(
    const _Float16 *__restrict__ X16, const float *__restrict__ xn, const float *__restrict__ xe, int64_t n_items, int32_t dpad,
    const _Float16 *__restrict__ H16, const float *__restrict__ hn, const float *__restrict__ he, int32_t n_tasks,
    const SplitTask *__restrict__ tasks, const int32_t *__restrict__ inv /* [tree][row] position in the tree's permutation */,
    const int32_t *__restrict__ item_at /* the item each ROW of the contraction stands for (split_mm_order_rows), or null: row = item */,
    float eps, uint8_t *__restrict__ side, int32_t *__restrict__ ones, unsigned int *__restrict__ amb_count,
    int2 *__restrict__ amb, unsigned int amb_cap,
    const int32_t *__restrict__ col_list /* [row tiles][n_tasks] the tasks each row tile needs, ascending; null: all of them */,
    const int32_t *__restrict__ col_count /* [row tiles] */,
    const int32_t *__restrict__ col_first /* [row tiles + 1] workgroups (chunks of COLS tasks) before each row tile */)
{
    constexpr int ROWS = BIG ? 256 : 128, COLS = ROWS, THREADS = BIG ? 1024 : 512;
    // dynamic LDS: during the contraction the stages of operand slabs, afterwards the result, 128 hyperplanes x ROWS rows at
    // a time (BIG: the two halves of the hyperplane tile in turn)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Cs = (float *)smem;
    __shared__ int s_ones[COLS], s_tree[COLS], s_start[COLS], s_end[COLS], s_task[COLS], s_slot[COLS];
    __shared__ float s_hn[COLS], s_he[COLS];
    __shared__ int2 s_open[SM_OPEN];   // pairs this tile's filter left open
    __shared__ int s_nopen;
    __shared__ unsigned int s_obase;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    // Workgroup b runs on XCD b % 8.  Without lists an XCD takes every 8th row tile and walks that tile's hyperplane tiles
    // back to back: the row tile (786 KB at D = 3000) is fetched from HBM once and then served by that XCD's
    // L2; the level's hyperplanes (a few MB) stay in the Infinity Cache for everybody.  With lists the (row tile, chunk
    // of its list) pairs, in row-tile order, are cut into 8 runs of equal length, one per XCD (col_first[tile] = pairs
    // before the tile): the same locality, and the XCDs finish together although the lists differ in length.
    const int n_ct = (n_tasks + COLS - 1) / COLS;
    int64_t row_tile;
    int c0;
    if (col_list) {
        const int n_rt = (int)((n_items + ROWS - 1) / ROWS);
        const int total = col_first[n_rt], run = (total + 7) / 8;
        const int slot = (int)(blockIdx.x >> 3), g = (int)(blockIdx.x & 7) * run + slot;
        if (slot >= run || g >= total) return;
        int lo = 0, hi = n_rt - 1;   // the last tile with col_first[tile] <= g
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (col_first[mid] <= g) lo = mid;
            else hi = mid - 1;
        }
        row_tile = lo;
        c0 = (g - col_first[lo]) * COLS;
    } else {
        row_tile = (int64_t)((blockIdx.x >> 3) / n_ct) * 8 + (blockIdx.x & 7);
        c0 = (int)((blockIdx.x >> 3) % n_ct) * COLS;
    }
    const int64_t r0 = row_tile * ROWS;
    if (r0 >= n_items) return;
    // the tasks (hyperplanes) of this workgroup: COLS consecutive ones of the level, or of the row tile's own list -- the
    // tasks that hold at least one of its rows (split_active_*_kernel).  Either way they ascend: a tree's tasks are
    // neighbours and sorted by start.
    const int n_mine = col_list ? col_count[row_tile] : n_tasks;
    if (c0 >= n_mine) return;   // uniform
    const int n_valid = n_mine - c0 < COLS ? n_mine - c0 : COLS;
    if (tid < COLS) {
        const int k = c0 + (tid < n_valid ? tid : n_valid - 1);
        const int col = col_list ? col_list[row_tile * n_tasks + k] : k;
        const SplitTask t = tasks[col];
        s_ones[tid] = 0;
        if (tid == 0) s_nopen = 0;
        s_task[tid] = col;
        s_slot[tid] = t.slot;   // row of the level's hyperplane image (= the task's number on a first attempt)
        s_tree[tid] = t.tree;
        s_start[tid] = t.start;
        s_end[tid] = t.start + t.count;
        s_hn[tid] = hn[t.slot];
        s_he[tid] = he[t.slot];
    }
    __syncthreads();   // the delivery addresses below come from s_task
    // the contraction (mm16.hpp): hyperplanes are the A side (m) and the rows the B side (n), so the result has a
    // row of X on the lane and the epilogue's look-ups and side bytes of a wave run along consecutive rows; rows
    // past the end repeat the last one: their products are never looked up
    constexpr int NB = BIG ? 2 : 1;
    f32x16 acc[NB][2];
    auto b_row = [&](int rt) {
        const int64_t r = r0 + rt < n_items ? r0 + rt : n_items - 1;
        return item_at ? (int64_t)item_at[r] : r;   // (a lane asks once: the delivery addresses are kept in registers)
    };
    auto a_row = [&](int rt) { return (int64_t)s_slot[rt]; };
    if constexpr (BIG) mm16_tile_256x256<64, 2>(X16, H16, dpad, smem, b_row, a_row, acc);   // (4 stages of 32 halfs, three deliveries in flight: 0.62 / 0.73 ms against 0.57 / 0.66 at C3)
    else mm16_tile<128, 64, 2>(X16, H16, dpad, smem, b_row, a_row, acc);
    const int lr = lane & 31, lh = lane >> 5;
    const int wm = BIG ? w >> 2 : w >> 1, wn = BIG ? w & 3 : w & 1;   // the wave's part of the tile: B rows wm * 32 NB .., A rows wn * 64 ..

    const int64_t row = r0 + (tid & (ROWS - 1));
    const bool row_ok = row < n_items;
    // bound of a pair = (EACC |y| + |d|) |g| + (|y| + |d|) |f|, 0.5 % of slack for its own roundings
    const int64_t item = row_ok ? (item_at ? (int64_t)item_at[row] : row) : 0;
    const float xn_r = xn[item], xe_r = xe[item];
    const float xa = (eps * xn_r + xe_r) * 1.005f, xb = (xn_r + xe_r) * 1.005f;
#pragma unroll
    for (int half = 0; half < COLS / 128; half++) {
        // The result goes to LDS as C[hyperplane][row], 128 hyperplanes at a time.  C/D layout of the 32x32 MFMA: n = lane & 31
        // (a row of X), m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (a hyperplane = a task of the level).
        if (half) __syncthreads();   // the first half has been looked up
        if ((wn >> 1) == half) {
#pragma unroll
            for (int tb = 0; tb < NB; tb++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        Cs[((wn & 1) * 64 + tn * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * ROWS + wm * 32 * NB + tb * 32 + lr] = acc[tb][tn][e];
        }
        __syncthreads();
        const int c_lo = half * 128;   // (indices into the workgroup's task tables)
        if (c_lo >= n_valid) break;   // uniform
        // A row needs ONE entry per tree: that of its node.  A wave takes 64 consecutive rows and one tree at a
        // time: the task look-up, the position look-up and the side bytes all run along consecutive rows.
        const int c_hi = (c_lo + 128 < n_valid ? c_lo + 128 : n_valid);   // tasks [c_lo, c_hi) are this pass's
        const int t_first = s_tree[c_lo], t_last = s_tree[c_hi - 1];
        constexpr int EU = BIG ? 4 : 8, TSTEP = THREADS / ROWS;   // trees per batch and thread (BIG: the other half of the result is still in registers); trees between a thread's steps
        for (int tb = t_first + tid / ROWS; tb <= t_last; tb += EU * TSTEP) {
            // the look-ups of a batch are issued together: where the row stands in each tree's permutation
            int a[EU], pos[EU], pp[EU];
#pragma unroll
            for (int u = 0; u < EU; u++) {
                const int tr = tb + u * TSTEP;
                pp[u] = (row_ok && tr <= t_last) ? inv[(int64_t)tr * n_items + row] : -1;
            }
#pragma unroll
            for (int u = 0; u < EU; u++) {
                // the node of the row in tree tr: among the tree's tasks in this tile (sorted by start), the one whose
                // segment holds the position; none: the row sits in a leaf, or in a node of another tile
                const int tr = tb + u * TSTEP;
                a[u] = -1;
                pos[u] = 0;
                // (the tree is the same for the whole wave: its tasks in the tile are found with one ballot per 64 slots,
                // and every lane counts the starts that do not exceed its position -- independent LDS broadcasts, no chain)
                int lo = -1, n_t = 0;
#pragma unroll
                for (int k = 0; k < COLS / WAVE; k++) {
                    const uint64_t mk = __ballot(k * WAVE + lane < n_valid && s_tree[k * WAVE + lane] == tr);
                    if (mk && lo < 0) lo = k * WAVE + (int)__builtin_ctzll(mk);
                    n_t += (int)__builtin_popcountll(mk);
                }
                int before = 0;
                for (int j = 0; j < n_t; j++) before += s_start[lo + j] <= pp[u] ? 1 : 0;
                if (pp[u] >= 0 && before > 0 && pp[u] < s_end[lo + before - 1]) {
                    a[u] = lo + before - 1;
                    pos[u] = pp[u] - s_start[lo + before - 1];
                }
            }
#pragma unroll
            for (int u = 0; u < EU; u++) {
                const int tr = tb + u * TSTEP;
                const bool mine = a[u] >= c_lo && a[u] < c_hi;
                const int cl = mine ? a[u] : 0;   // index into the workgroup's task tables
                const float c = Cs[(mine ? a[u] - c_lo : 0) * ROWS + (tid & (ROWS - 1))];   // bank = row: no conflict whatever the nodes are
                // false for NaN and for rows / hyperplanes that could not be scaled (norm = +inf)
                const bool decided = mine && fabsf(c) > xa * s_hn[cl] + xb * s_he[cl];
                const bool one = decided && c > 0.f;
                if (decided) side[(int64_t)tr * n_items + s_start[cl] + pos[u]] = (uint8_t)one;
                // right-side counts: one LDS atomic per wave when its rows share the node (the rule at shallow levels)
                const uint64_t mm = __ballot(mine);
                if (mm) {
                    const int cl0 = __builtin_amdgcn_readlane(cl, __builtin_ctzll(mm));
                    const uint64_t onem = __ballot(one);
                    if (__ballot(mine && cl != cl0) == 0) {
                        if (onem && lane == __builtin_ctzll(mm)) atomicAdd(&s_ones[cl0], (int)__builtin_popcountll(onem));
                    } else if (one) {
                        atomicAdd(&s_ones[cl], 1);
                    }
                }
                // open pairs are collected in LDS: all workgroups adding to the ONE global counter pair by pair (or
                // wave by wave) is what the kernel would otherwise wait for (~350 M same-address atomics per second)
                const bool open = mine && !decided;
                const uint64_t om = __ballot(open);
                if (om) {
                    int base = 0;
                    if (lane == __builtin_ctzll(om)) base = atomicAdd(&s_nopen, (int)__builtin_popcountll(om));
                    base = __builtin_amdgcn_readlane(base, __builtin_ctzll(om));
                    const int idx = base + (int)__builtin_popcountll(om & ((1ull << lane) - 1ull));
                    if (open) {
                        if (idx < SM_OPEN) {
                            s_open[idx] = make_int2((int)row, s_task[cl]);
                        } else {   // more than the LDS list holds: straight to the global list
                            const unsigned int g = atomicAdd(amb_count, 1u);
                            if (g < amb_cap) amb[g] = make_int2((int)row, s_task[cl]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    {
        const int n_open = s_nopen < SM_OPEN ? s_nopen : SM_OPEN;
        if (tid == 0 && n_open) s_obase = atomicAdd(amb_count, (unsigned int)n_open);
        __syncthreads();
        for (int i = tid; i < n_open; i += THREADS)
            if (s_obase + i < amb_cap) amb[s_obase + i] = s_open[i];
    }
    __syncthreads();
    if (tid < n_valid && s_ones[tid]) atomicAdd(&ones[s_task[tid]], s_ones[tid]);
}

// ---- which tasks does a row tile need? ------------------------------------------------------------
// A row is in ONE node per tree, and rows that are neighbours by id tend to be neighbours in the data (ids are
// handed out in the order samples first appear in the junction lines), so deep in the forest a tile of 256 rows meets only
// part of the level's split nodes.  active[tile][task] = 1 when some row of the tile is in the task's node; the lists
// made from it (ascending task numbers) are what split_mm_kernel multiplies the tile with.  One workgroup per task walks
// the node's slice of the permutation; the byte stores need no atomics.
__global__ __launch_bounds__(256) void split_active_mark_kernel(const SplitTask *__restrict__ tasks,
                                                                const int32_t *__restrict__ perm, int64_t n_items,
                                                                const int32_t *__restrict__ rank /* row of each item, or null: its id */,
                                                                int tile_shift, int32_t n_tasks, uint8_t *__restrict__ active)
{
    const SplitTask t = tasks[blockIdx.x];
    const int32_t *items = perm + TASK_ITEMS_AT(t, n_items);
    // neighbours in a node's list are often in the same tile (ids ascend along the list; ordered rows keep much of that):
    // only the first item of a run in the same tile stores.  The previous item's tile comes from the lane below.
    const int lane = threadIdx.x & (WAVE - 1);
    for (int i0 = 0; i0 < t.count; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool valid = i < t.count;
        const int32_t it = valid ? items[i] : 0;
        const int tile = (rank ? rank[it] : it) >> tile_shift;
        int prev = __shfl_up(tile, 1, WAVE);
        if (lane == 0) prev = -1;   // (the first lane of a wave always stores: cheaper than fetching its neighbour's row)
        if (valid && prev != tile) active[(int64_t)tile * n_tasks + blockIdx.x] = 1;
    }
}

// one workgroup per row tile: the marked tasks in ascending order, and how many
__global__ __launch_bounds__(256) void split_active_list_kernel(const uint8_t *__restrict__ active, int32_t n_tasks,
                                                                int32_t *__restrict__ col_list, int32_t *__restrict__ col_count)
{
    __shared__ int s_wave[4];
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const uint8_t *a = active + (int64_t)blockIdx.x * n_tasks;
    int32_t *out = col_list + (int64_t)blockIdx.x * n_tasks;
    int base = 0;
    for (int c = 0; c < n_tasks; c += 256) {
        const bool on = c + (int)threadIdx.x < n_tasks && a[c + threadIdx.x] != 0;
        const uint64_t m = __ballot(on);
        if (lane == 0) s_wave[w] = (int)__builtin_popcountll(m);
        __syncthreads();
        int before = base;
        for (int k = 0; k < w; k++) before += s_wave[k];
        if (on) out[before + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = c + (int)threadIdx.x;
        base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) col_count[blockIdx.x] = base;
}

// col_first[tile] = chunks of `cols` tasks in the lists of the tiles before it (one workgroup; n_rt is a few hundred)
__global__ __launch_bounds__(1024) void split_active_scan_kernel(const int32_t *__restrict__ col_count, int32_t n_rt, int32_t cols,
                                                                 int32_t *__restrict__ col_first, unsigned long long *__restrict__ launched)
{
    __shared__ int s_part[1024];
    const int per = (n_rt + 1023) / 1024, b = threadIdx.x * per;
    int sum = 0;
    for (int i = b; i < b + per && i < n_rt; i++) sum += (col_count[i] + cols - 1) / cols;
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = s_part[threadIdx.x] - sum;
    for (int i = b; i < b + per && i < n_rt; i++) {
        col_first[i] = run;
        run += (col_count[i] + cols - 1) / cols;
    }
    if (threadIdx.x == 1023) {
        col_first[n_rt] = s_part[1023];
        if (launched) atomicAdd(launched, (unsigned long long)s_part[1023]);   // (tile, chunk) pairs the contraction will execute
    }
}

// ---- the pairs the filter left open: canonical fp32 dot, one wave per pair ------------------------
__global__ __launch_bounds__(256) void split_amb_kernel(const float *__restrict__ X, int64_t n_items, int32_t dpad,
                                                        const SplitTask *__restrict__ tasks,
                                                        const int32_t *__restrict__ inv, uint32_t seed,
                                                        const float *__restrict__ hp,
                                                        const unsigned int *__restrict__ amb_count,
                                                        const int2 *__restrict__ amb, unsigned int amb_cap,
                                                        uint8_t *__restrict__ side, int32_t *__restrict__ ones,
                                                        const int32_t *__restrict__ item_at /* item of each row of the contraction, or null: the row */)
{
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int nvec = dpad / 4;
    unsigned int n = *amb_count;
    if (n > amb_cap) n = amb_cap;   // cannot happen: the list holds one entry per (row, tree) at most
    for (unsigned int i = blockIdx.x * (256 / WAVE) + w; i < n; i += gridDim.x * (256 / WAVE)) {
        const int2 pr = amb[i];   // (row of the contraction -- `inv` is indexed by it --, task)
        const SplitTask t = tasks[pr.y];
        const int64_t item = item_at ? item_at[pr.x] : pr.x;
        const float d = wave_dot((const float4 *)(X + item * dpad), (const float4 *)(hp + (int64_t)t.slot * dpad), nvec, lane);
        if (lane == 0) {
            const int pos = inv[(int64_t)t.tree * n_items + pr.x] - t.start;
            const uint32_t nseed = node_seed(seed, (uint32_t)t.tree, (uint32_t)t.level, (uint32_t)t.start, (uint32_t)t.attempt);
            // Angular::side: dot != 0 ? dot > 0 : coin flip
            const int s = d != 0.f ? (d > 0.f) : pos_flip(nseed, (uint32_t)pos);
            side[(int64_t)t.tree * n_items + t.start + pos] = (uint8_t)s;
            if (s) atomicAdd(&ones[pr.y], 1);
        }
    }
}

// ---- an order of the rows for the deep levels ---------------------------------------------------------------------
// The lists above are short when the rows of a tile are alike: in every tree they then sit in the same few nodes.  After the
// root level the forest itself says which rows are alike: key(item) = the side the item took at the root of trees 0 .. 11,
// one bit per tree.  Rows sorted by that key (ties by id: a counting sort with a stable in-block rank, deterministic) make
// the ROWS of the contraction from the second level on; `rank` (row of an item) and `item_at` (item of a row) translate.
// C3, blocks of 256 rows x 256 tasks the lists leave at the levels with 800 / 1600 / 1936 split nodes: 494 / 650 / 546,
// against 640 / 971 / 964 with the rows in id order and 784 / 1372 / 1568 without lists (scripts/active_pairs_probe.py).
// The order changes which products are computed, never a side: tests compare forests with MORNA_SPLIT_ORDER=0.
#define ORD_BITS 12
#define ORD_BINS (1 << ORD_BITS)
#define ORD_BLOCK 256

__global__ __launch_bounds__(256) void order_key_kernel(const uint8_t *__restrict__ side /* [tree][position = item at the root] */,
                                                        int64_t n_items, int n_bits, uint16_t *__restrict__ key)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    uint32_t k = 0;
    for (int t = 0; t < n_bits; t++) k = (k << 1) | (side[(int64_t)t * n_items + i] ? 1u : 0u);
    key[i] = (uint16_t)k;
}

__global__ __launch_bounds__(ORD_BLOCK) void order_hist_kernel(const uint16_t *__restrict__ key, int64_t n_items, int32_t n_blocks,
                                                               int32_t *__restrict__ cnt /* [ORD_BINS][blocks] */)
{
    __shared__ int s_cnt[ORD_BINS];
    for (int k = threadIdx.x; k < ORD_BINS; k += ORD_BLOCK) s_cnt[k] = 0;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * ORD_BLOCK + threadIdx.x;
    if (i < n_items) atomicAdd(&s_cnt[key[i]], 1);
    __syncthreads();
    for (int k = threadIdx.x; k < ORD_BINS; k += ORD_BLOCK) cnt[(int64_t)k * n_blocks + blockIdx.x] = s_cnt[k];
}

// one wave per key: cnt[k][b] <- items with key k in the blocks before b; total[k]
__global__ __launch_bounds__(256) void order_scan_bins_kernel(int32_t *__restrict__ cnt, int32_t n_blocks, int32_t *__restrict__ total)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int k = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
    int32_t *row = cnt + (int64_t)k * n_blocks;
    int run = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += WAVE) {
        const int v = b0 + lane < n_blocks ? row[b0 + lane] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int u = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += u;
        }
        if (b0 + lane < n_blocks) row[b0 + lane] = run + incl - v;
        run += __shfl(incl, WAVE - 1, WAVE);
    }
    if (lane == 0) total[k] = run;
}

// total[k] <- items with a smaller key (one workgroup, four keys per thread)
__global__ __launch_bounds__(1024) void order_scan_keys_kernel(int32_t *__restrict__ total)
{
    static_assert(ORD_BINS == 4096, "four bins per thread");
    __shared__ int s_part[1024];
    int t[4];
#pragma unroll
    for (int u = 0; u < 4; u++) t[u] = total[threadIdx.x * 4 + u];
    const int mine = t[0] + t[1] + t[2] + t[3];
    s_part[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = threadIdx.x >= off ? s_part[threadIdx.x - off] : 0;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    int base = s_part[threadIdx.x] - mine;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        total[threadIdx.x * 4 + u] = base;
        base += t[u];
    }
}

__global__ __launch_bounds__(ORD_BLOCK) void order_rank_kernel(const uint16_t *__restrict__ key, int64_t n_items, int32_t n_blocks,
                                                               const int32_t *__restrict__ cnt, const int32_t *__restrict__ base,
                                                               int32_t *__restrict__ rank, int32_t *__restrict__ item_at)
{
    __shared__ uint16_t s_key[ORD_BLOCK];
    const int64_t i = (int64_t)blockIdx.x * ORD_BLOCK + threadIdx.x;
    const uint16_t k = i < n_items ? key[i] : (uint16_t)0xFFFF;
    s_key[threadIdx.x] = k;
    __syncthreads();
    if (i >= n_items) return;
    int same = 0;   // earlier items of the block with the same key
    for (int j = 0; j < (int)threadIdx.x; j++) same += s_key[j] == k ? 1 : 0;
    const int32_t r = base[k] + cnt[(int64_t)k * n_blocks + blockIdx.x] + same;
    rank[i] = r;
    item_at[r] = (int32_t)i;
}

// inv by row from inv by item: inv_r[tree][rank[item]] = inv[tree][item]
__global__ __launch_bounds__(256) void order_inv_kernel(const int32_t *__restrict__ inv_by_item, const int32_t *__restrict__ rank,
                                                        int64_t n_items, int64_t total, int32_t *__restrict__ inv)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t t = i / n_items, item = i - t * n_items;
    inv[t * n_items + rank[item]] = inv_by_item[i];
}

// scratch slots 30..32: [rank | item_at], inv by row, the counting sort's table (+ keys).  The fp16 rows stay where they are:
// a lane of the contraction fetches its row's address through item_at once (a copy of the image in row order cost 0.17 ms
// of HBM time beside the second level's two_means).  Enqueued on `stream` (the side stream, under that two_means).
int split_mm_order_rows(morna_index *h, const uint8_t *side, const int32_t *inv_by_item, int32_t n_trees, hipStream_t stream,
                        const int32_t **rank_out, int32_t **inv_out)
{
    const int64_t N = h->n_items;
    ScratchRef<int32_t> maps(h->scratch[30]), invr(h->scratch[31]), table(h->scratch[32]);
    const int32_t n_blocks = (int32_t)((N + ORD_BLOCK - 1) / ORD_BLOCK);
    MORNA_TRY(maps.alloc((size_t)N * 2));
    MORNA_TRY(invr.alloc((size_t)n_trees * N));
    MORNA_TRY(table.alloc((size_t)n_blocks * ORD_BINS + ORD_BINS + (size_t)(N + 1) / 2));
    int32_t *key_base = table.p + (size_t)n_blocks * ORD_BINS;   // [ORD_BINS] items with a smaller key
    uint16_t *key = (uint16_t *)(key_base + ORD_BINS);
    int32_t *rank = maps.p, *item_at = maps.p + N;
    const int n_bits = n_trees < ORD_BITS ? n_trees : ORD_BITS;
    hipLaunchKernelGGL(order_key_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, side, N, n_bits, key);
    hipLaunchKernelGGL(order_hist_kernel, dim3((unsigned)n_blocks), dim3(ORD_BLOCK), 0, stream, key, N, n_blocks, table.p);
    hipLaunchKernelGGL(order_scan_bins_kernel, dim3(ORD_BINS / 4), dim3(256), 0, stream, table.p, n_blocks, key_base);
    hipLaunchKernelGGL(order_scan_keys_kernel, dim3(1), dim3(1024), 0, stream, key_base);
    hipLaunchKernelGGL(order_rank_kernel, dim3((unsigned)n_blocks), dim3(ORD_BLOCK), 0, stream, key, N, n_blocks, table.p, key_base, rank,
                       item_at);
    const int64_t total = (int64_t)n_trees * N;
    hipLaunchKernelGGL(order_inv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, inv_by_item, rank, N, total, invr.p);
    HIP_TRY(hipGetLastError());
    h->ord_valid = true;
    *rank_out = rank;
    *inv_out = invr.p;
    return MORNA_OK;
}

// scratch slots 19..23 of the handle: fp16 rows, their norms, fp16 hyperplanes of the level, their norms,
// the open-pair list (first 16 bytes: its counter)
int split_mm_prepare_rows(morna_index *h, hipStream_t stream)
{
    if (h->half_valid) return MORNA_OK;
    ScratchRef<_Float16> x16(h->scratch[19]);
    ScratchRef<float> xn(h->scratch[20]);   // [0, N): norms; [N, 2N): 2^-e per row (the query filter unscales with it); [2N, 3N): |s x - y|
    MORNA_TRY(x16.alloc((size_t)h->n_items * h->dpad));
    MORNA_TRY(xn.alloc((size_t)h->n_items * 3));
    hipLaunchKernelGGL(rows_to_half_kernel, dim3((unsigned)((h->n_items + 3) / 4)), dim3(256), 0, stream, h->X.p,
                       h->n_items, h->dpad, x16.p, xn.p, xn.p + 2 * h->n_items, xn.p + h->n_items, (unsigned int *)nullptr);
    HIP_TRY(hipGetLastError());
    h->half_valid = true;
    return MORNA_OK;
}

// fp16 image of any row-major fp32 matrix with the library's row stride (the query vectors of a batch, knn.hip):
// dst[rows][dpad], per row an upper bound of the image's norm, of its rounding error's norm, and 2^-e
int split_mm_convert_rows(morna_index *h, const float *src, int64_t rows, _Float16 *dst, float *norm, float *err,
                          float *inv_scale, hipStream_t stream)
{
    hipLaunchKernelGGL(rows_to_half_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, src, rows, h->dpad, dst,
                       norm, err, inv_scale, (unsigned int *)nullptr);
    HIP_TRY(hipGetLastError());
    return MORNA_OK;
}

// n_slots: hyperplanes of the level (hp_level[slot]); the tasks are all of them in order (a first attempt: slot = task
// number) or the retried ones among them
int split_mm_level(morna_index *h, const SplitTask *d_tasks, int32_t n_tasks, int32_t n_slots, const float *hp_level,
                   const int32_t *perm, const int32_t *inv, uint32_t seed, uint8_t *side, int32_t *ones)
{
    const int64_t N = h->n_items;
    ScratchRef<_Float16> x16(h->scratch[19]), h16(h->scratch[21]);
    ScratchRef<float> xn(h->scratch[20]), hn(h->scratch[22]);
    ScratchRef<int32_t> maps(h->scratch[30]);
    ScratchRef<uint8_t> ambuf(h->scratch[23]);
    // the rows of the contraction: the items in id order, or in the order split_mm_order_rows made (`inv` is then by row too)
    const bool ord = h->ord_valid;
    const _Float16 *rows16 = x16.p;
    const float *rows_n = xn.p, *rows_e = xn.p + 2 * N;
    const int32_t *rank = ord ? maps.p : nullptr, *item_at = ord ? maps.p + N : nullptr;
    const size_t cap = (size_t)N * h->n_trees;   // a row is in at most one split node per tree
    if (cap > 0xFFFFFFF0u) {
        set_error("split_mm_level: %lld x %d (row, tree) pairs exceed the open-pair list", (long long)N, h->n_trees);
        return MORNA_E_INVALID;
    }
    MORNA_TRY(h16.alloc((size_t)n_slots * h->dpad));
    MORNA_TRY(hn.alloc((size_t)n_slots * 2));   // norms, then rounding-error norms
    MORNA_TRY(ambuf.alloc(16 + cap * sizeof(int2)));
    unsigned int *amb_count = (unsigned int *)ambuf.p;
    int2 *amb = (int2 *)(ambuf.p + 16);
    // the open-pair counter is reset by the kernel that converts the level's hyperplanes (same stream, just before)
    hipLaunchKernelGGL(rows_to_half_kernel, dim3((unsigned)((n_slots + 3) / 4)), dim3(256), 0, h->stream, hp_level,
                       (int64_t)n_slots, h->dpad, h16.p, hn.p, hn.p + n_slots, (float *)nullptr, amb_count);
    // 256 x 256 tiles once the level has hyperplane tiles enough for them to fill the chip in even rounds (MORNA_SPLIT_BIG=0:
    // the 128 x 128 form everywhere)
    static const bool big_on = !(getenv("MORNA_SPLIT_BIG") && atoi(getenv("MORNA_SPLIT_BIG")) == 0);
    // the tasks each row tile needs, from the second level of a tree on (MORNA_SPLIT_LISTS=0: every tile x every task)
    static const bool lists_on = !(getenv("MORNA_SPLIT_LISTS") && atoi(getenv("MORNA_SPLIT_LISTS")) == 0);
    if (big_on && (n_tasks >= 1024 || (ord && lists_on && n_tasks >= 512))) {
        const unsigned n_rt = (unsigned)((N + 255) / 256), n_ct = (unsigned)((n_tasks + 255) / 256);
        const int32_t *col_list = nullptr, *col_count = nullptr, *col_first = nullptr;
        if (lists_on) {
            ScratchRef<uint8_t> active(h->scratch[27]);
            ScratchRef<int32_t> lists(h->scratch[28]);
            MORNA_TRY(active.alloc((size_t)n_rt * n_tasks));
            MORNA_TRY(lists.alloc((size_t)n_rt * n_tasks + 2 * n_rt + 1));
            HIP_TRY(hipMemsetAsync(active.p, 0, (size_t)n_rt * n_tasks, h->stream));
            hipLaunchKernelGGL(split_active_mark_kernel, dim3((unsigned)n_tasks), dim3(256), 0, h->stream, d_tasks, perm, N, rank, 8,
                               n_tasks, active.p);
            hipLaunchKernelGGL(split_active_list_kernel, dim3(n_rt), dim3(256), 0, h->stream, active.p, n_tasks, lists.p,
                               lists.p + (size_t)n_rt * n_tasks);
            col_list = lists.p;
            col_count = lists.p + (size_t)n_rt * n_tasks;
            // d_stat[1]: 256 x 256 tiles launched through the lists while the MORNA_T_SPLIT_MM timer is on (resolve_timers prices them)
            const bool count_tiles = h->timing && (h->timing_mask & (1u << MORNA_T_SPLIT_MM));
            hipLaunchKernelGGL(split_active_scan_kernel, dim3(1), dim3(1024), 0, h->stream, col_count, (int32_t)n_rt, 256,
                               lists.p + (size_t)n_rt * n_tasks + n_rt, count_tiles ? h->d_stat.p + 1 : (unsigned long long *)nullptr);
            col_first = col_count + n_rt;
        }
        // executed flops: every launched (row tile, chunk of 256 tasks) pair is a 256 x 256 x dpad product; through the lists
        // their number is only known on the device (counted above)
        ScopedTimer tmm(h, MORNA_T_SPLIT_MM, lists_on ? 0 : 2ll * 256 * 256 * h->dpad * (int64_t)n_rt * n_ct);
        const float eps1 = (4.f * (float)h->dpad + 2.f) * 5.9604645e-8f + 4.1e-6f;   // EACC for ONE chain of dpad products
        HIP_TRY(hipFuncSetAttribute((const void *)split_mm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 256 * 4));
        hipLaunchKernelGGL(split_mm_kernel<true>, dim3(8u * ((n_rt + 7) / 8) * n_ct), dim3(1024), 128 * 256 * 4, h->stream, rows16, rows_n,
                           rows_e, N, h->dpad, h16.p, hn.p, hn.p + n_slots, n_tasks, d_tasks, inv, item_at, eps1, side, ones,
                           amb_count, amb, (unsigned int)cap, col_list, col_count, col_first);
    } else {
        const unsigned n_rt = (unsigned)((N + 127) / 128), n_ct = (unsigned)((n_tasks + 127) / 128);
        static_assert(MM16_LDS == 128 * 128 * 4, "the slabs and the result tile share the dynamic LDS");
        HIP_TRY(hipFuncSetAttribute((const void *)split_mm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 128 * 4));
        ScopedTimer tmm(h, MORNA_T_SPLIT_MM, 2ll * 128 * 128 * h->dpad * (int64_t)n_rt * n_ct);
        hipLaunchKernelGGL(split_mm_kernel<false>, dim3(8u * ((n_rt + 7) / 8) * n_ct), dim3(512), 128 * 128 * 4, h->stream, rows16, rows_n,
                           rows_e, N, h->dpad, h16.p, hn.p, hn.p + n_slots, n_tasks, d_tasks, inv, item_at, sm_eps(h->dpad), side,
                           ones, amb_count, amb, (unsigned int)cap, (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr);
    }
    hipLaunchKernelGGL(split_amb_kernel, dim3((unsigned)(4 * h->n_cus)), dim3(256), 0, h->stream, h->X.p, N, h->dpad, d_tasks,
                       inv, seed, hp_level, amb_count, amb, (unsigned int)cap, side, ones, item_at);
    HIP_TRY(hipGetLastError());
    // MORNA_DEBUG_OPEN=1: how many (row, tree) pairs the filter of this level left to the canonical dot (stderr;
    // costs a synchronisation, measurement only)
    static const bool debug_open = getenv("MORNA_DEBUG_OPEN") && atoi(getenv("MORNA_DEBUG_OPEN")) != 0;
    if (debug_open) {
        unsigned int n_open = 0;
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(&n_open, amb_count, 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "[morna] split_mm level: D=%d split nodes=%d (row, tree) pairs<=%lld open=%u (%.4f %%)\n", h->dim,
                n_tasks, (long long)cap, n_open, 100.0 * n_open / (double)cap);
    }
    return MORNA_OK;
}

}  // namespace morna
