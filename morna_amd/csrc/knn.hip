// knn.hip -- batched nearest-neighbour queries on gfx950.
//
// query_kernel   : annoy's _get_all_nns (get_nns_by_vector / get_nns_by_item,
//                  reference call sites morna.py:651, 659, 762, 769) -- one
//                  workgroup per query fuses the best-first forest traversal
//                  with the brute-force angular refine of the candidate rows and
//                  the (distance, id) top-k selection.
// exact_*_kernel : MornaSearch.exact_search_nn + cosine_distance (morna.py:
//                  681-716, 101-114) -- an fp32 scan of all rows picks every row
//                  that can be in the top k, then those rows are re-evaluated in
//                  the reference's sequential fp64 order and ranked with the
//                  bisect_left tie rule, so ids and distances are bit-exact.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <type_traits>

#include "common.hpp"
#include "devutil.hpp"
#include "mm16.hpp"

namespace morna {

#define Q_THREADS 256
#define Q_WAVES (Q_THREADS / WAVE)

struct QueryParams {
    const float *X;
    const float *norm2;
    int64_t n_items;
    int32_t dpad, n_trees, K;
    const int32_t *node_rec;   // [n_nodes][4] child0, child1, start, count
    const int32_t *node_tree;
    const int32_t *node_hp;
    const float *hp;
    const int32_t *perm;
    int64_t n_nodes;
    const float *Q;            // [nq][dpad] or null
    const int32_t *items;      // [nq] or null
    int32_t n_inline;          // spread form, by item, up to 16 queries: the item numbers travel with the launch (no copy)
    int32_t inline_items[16];
    int32_t k, search_k;
    int32_t cap;               // candidate capacity per query
    int32_t bm_words;
    // workspace, one slice per query
    uint64_t *pq;              // [nq][n_nodes]
    int32_t *cand;             // [nq][cap]
    int64_t *cand_off;         // [nq] one-wave descent: >= 0: the candidates are perm[cand_off .. + ncand) (one leaf, not copied); -1: cand[]
    int32_t zero_copy;         // the consumer reads the candidates through cand_off (query_cand_dots_kernel); 0: always copied to cand[]
    uint64_t *keys;            // [nq][cap]
    float *low;                // [nq][cap] lower bound of a candidate's distance (filter modes)
    int32_t *ncand;            // [nq] unique candidates found by the traversal (may exceed cap)
    float *qpp;                // [nq] canonical dot(q, q)
    uint32_t *bm_global;       // [nq][bm_words] when the bitmap does not fit LDS
    int32_t *ids_out;          // [nq][k]
    float *dist_out;           // [nq][k]
    int32_t *count_out;        // [nq]
    unsigned long long *stat;  // rows read: hyperplane dots + unique candidates + query
    // Candidate filter on the fp16 image of the rows (splitmm.hip); X16 null: every candidate gets the fp32 dot.
    // scores null: the refine kernel takes the filter dots itself, row by row (fp16 row x fp32 query), and every
    // filter distance is within `delta` of the canonical one.  scores set: they were taken for ALL rows at once on
    // the matrix cores (query_scores_kernel) from the fp16 images of rows AND queries; the bound is then per pair,
    // from the measured norms and rounding-error norms of the two images.
    const _Float16 *X16;       // [n_items][dpad], row r scaled by 2^e(r)
    const float *xscale;       // [n_items] 2^-e(r); 0: the row has no fp16 image
    const float *xn16, *xe16;  // [n_items] upper bounds of |y_r| and |2^e x_r - y_r|
    float delta;
    const float *scores;       // [nq][n_items]  sum_i y_r[i] g_q[i]
    const float *qscale, *qn16, *qe16;   // per query image: indexed by the item id (by-item) or the query number
    float eacc;       // bound of the fp32 accumulation of a filter dot: two chains (128 x 128 form) ...
    float eacc_big;   // ... one chain: the rows below big_rows went through the 256 x 256 form
    int64_t big_rows;
};

__device__ inline uint64_t pq_key(float d, int32_t node)
{
    return ((uint64_t)f32_orderable(d) << 32) | (uint32_t)node;
}

// the query's row (spread kernels): a stored row -- its number from the launch arguments or from the items array -- or a
// row of the staged query vectors
__device__ inline const float4 *query_row(const QueryParams &P, int64_t qi)
{
    if (P.n_inline) return (const float4 *)(P.X + (int64_t)P.inline_items[qi] * P.dpad);
    return P.items ? (const float4 *)(P.X + (int64_t)P.items[qi] * P.dpad) : (const float4 *)(P.Q + qi * P.dpad);
}

// block-wide min of a uint64 (all threads get the result)
__device__ inline uint64_t block_min_u64(uint64_t v, uint64_t *s_red, int tid)
{
    v = wave_min_u64(v);
    __syncthreads();
    if ((tid & (WAVE - 1)) == 0) s_red[tid / WAVE] = v;
    __syncthreads();
    uint64_t r = s_red[0];
#pragma unroll
    for (int i = 1; i < Q_WAVES; i++) r = s_red[i] < r ? s_red[i] : r;
    return r;
}

// k-th smallest of n distinct keys (~0 when n < k), the same value in every thread.  While a thread's share of the keys
// fits KTH_HELD registers and k <= KTH_MAX_K: every wave extracts the k smallest of ITS keys in k rounds of register compares
// and a wave minimum -- no barrier -- and the k-th smallest of the waves' k-lists is found by counting, one key per lane.
// Otherwise (and as round 1 did throughout): k rounds of a block-wide minimum over the keys in memory, two barriers each.
#define KTH_HELD 16
#define KTH_MAX_K 32
__device__ inline uint64_t block_kth_smallest(const uint64_t *keys, int n, int k, int tid, uint64_t *s_red, uint64_t *s_top)
{
    const int lane = tid & (WAVE - 1), w = tid / WAVE;
    if (n <= Q_THREADS * KTH_HELD && k <= KTH_MAX_K) {
        uint64_t held[KTH_HELD];
#pragma unroll
        for (int u = 0; u < KTH_HELD; u++) held[u] = tid + Q_THREADS * u < n ? keys[tid + Q_THREADS * u] : ~0ull;
        uint64_t prev = 0;
        for (int r = 0; r < k; r++) {
            uint64_t best = ~0ull;
#pragma unroll
            for (int u = 0; u < KTH_HELD; u++)
                if ((r == 0 || held[u] > prev) && held[u] < best) best = held[u];
            prev = wave_min_u64(best);
            if (lane == 0) s_top[w * KTH_MAX_K + r] = prev;   // ~0: this wave has run out of keys
        }
        __syncthreads();
        // Q_WAVES * k keys, k-th smallest: a key's rank is the number of keys below it (the real ones are distinct)
        uint64_t found = ~0ull;
        for (int i = tid; i < Q_WAVES * k; i += Q_THREADS) {
            const uint64_t v = s_top[(i / k) * KTH_MAX_K + i % k];
            int below = 0;
            for (int j = 0; j < Q_WAVES * k; j++) below += s_top[(j / k) * KTH_MAX_K + j % k] < v ? 1 : 0;
            if (v != ~0ull && below == k - 1) found = v;
        }
        return block_min_u64(found, s_red, tid);
    }
    uint64_t kth = 0;
    for (int r = 0; r < k; r++) {
        uint64_t best = ~0ull;
        for (int c = tid; c < n; c += Q_THREADS) {
            const uint64_t kk = keys[c];
            if ((r == 0 || kk > kth) && kk < best) best = kk;
        }
        kth = block_min_u64(best, s_red, tid);
    }
    return kth;
}

// A canonical dot with the query held in the wave's registers: the whole row is requested at once, all its 1-KiB pieces in
// flight together (through wave_dot(), four loads in flight, a lone wave moves a 12-KB hyperplane in ~2 us).  NV = float4
// per lane of a row (dpad / 256).  The same fmaf chains as wave_dot().
template <int NV>
__device__ inline float wave_dot_held(const float4 *__restrict__ row, const float4 (&qr)[NV > 0 ? NV : 1], int lane)
{
    float4 x[NV > 0 ? NV : 1];
#pragma unroll
    for (int k = 0; k < NV; k++) x[k] = row[lane + WAVE * k];
    Acc4 s = acc4_zero();
#pragma unroll
    for (int k = 0; k < NV; k++) fma4(s, x[k], qr[k]);
    return acc4_finish(s);
}

// ---- traversal: annoy's _get_all_nns up to the candidate set (oracle/annoy_oracle.c:453-504) -------------------
// One workgroup per query: all root margins (every wave takes trees), then the best-first descent by wave 0 (array
// priority queue, bitmap de-duplication of the leaves' ids).  Leaves the unique candidates in cand[].
template <bool BM_LDS>
__global__ __launch_bounds__(Q_THREADS) void query_traverse_kernel(QueryParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *qv = (float4 *)smem;
    uint32_t *bm = BM_LDS ? (uint32_t *)(smem + (size_t)P.dpad * sizeof(float))
                          : P.bm_global + (size_t)blockIdx.x * P.bm_words;
    __shared__ int s_ncand;

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int64_t qi = blockIdx.x;
    const int nvec = P.dpad / 4;
    const int T = P.n_trees;

    const float4 *src = P.items ? (const float4 *)(P.X + (int64_t)P.items[qi] * P.dpad)
                                : (const float4 *)(P.Q + qi * P.dpad);
    for (int i = tid; i < nvec; i += Q_THREADS) qv[i] = src[i];
    for (int i = tid; i < P.bm_words; i += Q_THREADS) bm[i] = 0u;
    if (tid == 0) s_ncand = 0;
    __syncthreads();

    uint64_t *pq = P.pq + qi * P.n_nodes;
    int32_t *cand = P.cand + qi * P.cap;
    const bool roots_split = P.n_items > P.K;   // every root is a split node (count = N > K)

    // ---- phase 1: the roots all sit at +inf and are expanded before anything else;
    //      their margins are independent, so every wave takes a share.
    if (w == Q_WAVES - 1) {
        float pp = wave_dot(qv, qv, nvec, lane);
        if (lane == 0) P.qpp[qi] = pp;
    }
    if (roots_split) {
        for (int t = w; t < T; t += Q_WAVES) {
            const float m = wave_dot((const float4 *)(P.hp + (int64_t)P.node_hp[t] * P.dpad), qv, nvec, lane);
            // slot s is only ever touched by lane s % 64 of wave 0 later on; written here once
            if (lane == 0) {
                pq[2 * t] = pq_key(m, P.node_rec[4 * t + 1]);       // min(+inf, margin), children[1]
                pq[2 * t + 1] = pq_key(-m, P.node_rec[4 * t + 0]);  // min(+inf, -margin), children[0]
            }
        }
    } else {
        for (int t = tid; t < T; t += Q_THREADS) pq[t] = pq_key(INFINITY, t);
    }
    __syncthreads();

    // ---- phase 2: best-first traversal by wave 0 (max-heap semantics on (bound, node id));
    //      the queue is an unsorted array scanned by the wave, popped slots are zeroed.
    if (w == 0) {
        int ndots = roots_split ? T : 0;
        int hn = roots_split ? 2 * T : T;
        int64_t nn = 0;
        const int64_t search_k = P.search_k;
        while (nn < search_k) {
            uint64_t best = 0;
            int bestpos = -1;
            for (int i = lane; i < hn; i += WAVE) {
                uint64_t kk = pq[i];
                if (kk > best) { best = kk; bestpos = i; }
            }
            const uint64_t top = wave_max_u64(best);
            if (top == 0) break;                       // queue empty
            if (best == top && bestpos >= 0) pq[bestpos] = 0;   // exactly one lane owns it
            const int32_t node = (int32_t)(uint32_t)top;
            const float d = f32_from_orderable((uint32_t)(top >> 32));
            const int32_t c0 = P.node_rec[4 * node + 0], c1 = P.node_rec[4 * node + 1];
            const int32_t start = P.node_rec[4 * node + 2], count = P.node_rec[4 * node + 3];
            if (c0 < 0) {
                // leaf: nns.insert(all ids); duplicates across trees are dropped by the bitmap
                const int32_t *src_ids = P.perm + (int64_t)P.node_tree[node] * P.n_items + start;
                for (int i = lane; i < count; i += WAVE) {
                    const int32_t id = src_ids[i];
                    const uint32_t bit = 1u << (id & 31);
                    const uint32_t old = atomicOr(&bm[id >> 5], bit);
                    if (!(old & bit)) {
                        const int slot = atomicAdd(&s_ncand, 1);
                        if (slot < P.cap) cand[slot] = id;
                    }
                }
                nn += count;
            } else {
                const float m = wave_dot((const float4 *)(P.hp + (int64_t)P.node_hp[node] * P.dpad), qv, nvec, lane);
                if (lane == (hn & (WAVE - 1))) pq[hn] = pq_key(d < m ? d : m, c1);
                if (lane == ((hn + 1) & (WAVE - 1))) pq[hn + 1] = pq_key(d < -m ? d : -m, c0);
                hn += 2;
                ndots++;
            }
        }
        if (lane == 0) {   // (LDS operations of a wave complete in order: the count includes the last leaf's atomics)
            const int nc = s_ncand;
            P.ncand[qi] = nc;
            atomicAdd(P.stat, (unsigned long long)(ndots + (nc < P.cap ? nc : P.cap) + 1));
        }
    }
}

// ---- small batches: ONE query spread over the chip ----------------------------------------------------------------
// The reference answers one query per process (morna.py:1345-1484 -> 651-665 / 762-774).  With one workgroup per query
// (above) a lone query has one CU of 256: its ~200 root margins and the ~2000 candidate rows of its leaf are read by four
// waves, 0.78 ms at C3 against 2 us for the bytes at the HBM rate.  For batches too small to fill the chip the same
// arithmetic is dealt out: (A) a wave per (query, tree) for the root margins, (B) one wave per query for the best-first
// descent -- a chain of dependent pops, each a node record, a hyperplane and one canonical dot --, (C) a wave per two
// candidates for the CANONICAL fp32 dot of every candidate (no fp16 filter: twice the bytes of the filter, but one
// dependent stage less: no k-th smallest bound, no compaction, no second gather), (D) the k smallest (distance, id) of
// the keys by one workgroup.  Same operations on the same values as the batch forms: identical answers.
__global__ __launch_bounds__(Q_THREADS) void query_roots_kernel(QueryParams P)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int64_t qi = blockIdx.y;
    const int nvec = P.dpad / 4;
    const float4 *q = query_row(P, qi);
    if (blockIdx.x == 0 && w == Q_WAVES - 1) {
        const float pp = wave_dot(q, q, nvec, lane);
        if (lane == 0) P.qpp[qi] = pp;
    }
    const int t = blockIdx.x * Q_WAVES + w;
    if (t >= P.n_trees || !(P.n_items > P.K)) return;   // (roots that are leaves: the descent seeds the queue itself)
    uint64_t *pq = P.pq + qi * P.n_nodes;
    const float m = wave_dot((const float4 *)(P.hp + (int64_t)P.node_hp[t] * P.dpad), q, nvec, lane);
    if (lane == 0) {
        pq[2 * t] = pq_key(m, P.node_rec[4 * t + 1]);       // min(+inf, margin), children[1]
        pq[2 * t + 1] = pq_key(-m, P.node_rec[4 * t + 0]);  // min(+inf, -margin), children[0]
    }
}

// Root margins of a BATCH: a workgroup takes QR_Q queries (their rows in LDS) and a share of the trees; a wave fetches a
// root's hyperplane ONCE into registers, all its pieces in flight, and takes its canonical dot with each of the QR_Q queries
// (one workgroup per query fetched every hyperplane once per query: 200 k dots x 12 KB out of L2 for 1000 queries, and a
// wave's 50 dots one after the other, 100 us of a workgroup's life).  The descent then runs as in the spread form, one wave
// per query (query_descend_kernel).  Same fmaf chains as wave_dot(): a product's factors commute.
#ifndef QR_Q
#define QR_Q 4         // (8 queries and / or 512 threads, 2 queries: the same time within 2 % at 50k and at 6250 rows)
#endif
#ifndef QR_THREADS
#define QR_THREADS 256
#endif
#define QR_WAVES (QR_THREADS / WAVE)
#define QR_TREES 48   // trees per workgroup
template <int NV>
__global__ __launch_bounds__(QR_THREADS) void query_roots_batch_kernel(QueryParams P, int32_t nq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *qs = (float4 *)smem;   // [QR_Q][nvec]
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int nvec = P.dpad / 4;
    const int64_t q0 = (int64_t)blockIdx.x * QR_Q;
    const int nqs = (int)(nq - q0 < QR_Q ? nq - q0 : QR_Q);
    for (int i = tid; i < QR_Q * nvec; i += QR_THREADS) {
        const int qq = i / nvec, v = i - qq * nvec;
        qs[i] = qq < nqs ? query_row(P, q0 + qq)[v] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (blockIdx.y == 0)   // the queries' own canonical norms, once
        for (int qq = w; qq < nqs; qq += QR_WAVES) {
            const float pp = wave_dot(qs + qq * nvec, qs + qq * nvec, nvec, lane);
            if (lane == 0) P.qpp[q0 + qq] = pp;
        }
    const int t_end = (int)(blockIdx.y + 1) * QR_TREES < P.n_trees ? (int)(blockIdx.y + 1) * QR_TREES : P.n_trees;
    for (int t = (int)blockIdx.y * QR_TREES + w; t < t_end; t += QR_WAVES) {
        const float4 *hrow = (const float4 *)(P.hp + (int64_t)P.node_hp[t] * P.dpad);
        const int32_t c0 = P.node_rec[4 * t + 0], c1 = P.node_rec[4 * t + 1];
        float4 hr[NV];
#pragma unroll
        for (int k = 0; k < NV; k++) hr[k] = hrow[lane + WAVE * k];
#pragma unroll
        for (int qq = 0; qq < QR_Q; qq++) {
            if (qq < nqs) {   // uniform
                Acc4 s4 = acc4_zero();
#pragma unroll
                for (int k = 0; k < NV; k++) fma4(s4, hr[k], qs[qq * nvec + lane + WAVE * k]);
                const float m = acc4_finish(s4);
                if (lane == 0) {
                    uint64_t *pq = P.pq + (q0 + qq) * P.n_nodes;
                    pq[2 * t] = pq_key(m, c1);        // min(+inf, margin), children[1]
                    pq[2 * t + 1] = pq_key(-m, c0);   // min(+inf, -margin), children[0]
                }
            }
        }
    }
}

#define PQ_LDS 1024   // queue entries kept in LDS by the one-wave descent; entries beyond stay in the global array
// One wave per query.  A pop is a chain -- the entry with the largest bound, its node's record and hyperplane, one dot, two
// pushes -- and the descent is a chain of pops (C3 at morna's defaults: 14 internal nodes and one leaf; annoy's queue hops
// between the trees), so what counts is the length of ONE pop.  A lone wave issues an instruction every ~4.3 cycles and a
// trip to HBM takes ~1000, so:
//   * the queue sits in LDS, bounds and nodes as two 32-bit arrays: the maximum is taken on the 32-bit bounds (8 x v_max +
//     a 6-step cross-lane maximum); only when two entries share the largest bound do the 64-bit keys decide (annoy's
//     (bound, node) order), by the slow scan;
//   * the record and the hyperplane of a popped node are requested together (the hyperplane slot of every entry sits beside
//     it: looked up when the node is PUSHED, under its parent's hyperplane fetch): one round trip per pop; the query stays in
//     registers (NV > 0) and a hyperplane's 1-KiB pieces are all requested at once;
//   * (tried: the margins, records and child slots of the roots' children -- most of the pops -- made ahead by the root-margin
//     kernel, so that those pops touch LDS only: the descent 22 -> 18 us, the root kernel 4.9 -> 7.3 us with three times the
//     dots, the launch no shorter; dropped)
//   * the first leaf, when it ends the search (search_k = 100, leaves of ~K ids), is not copied: its slice of the tree's
//     permutation IS the candidate list (distinct ids: a tree lists an item once).
// NV: float4 per lane of a row (dpad / 256) when the query fits the wave's registers; 0: any row length, through
// wave_dot().  Same fmaf chains either way.
// the wave's maximum of a 32-bit value (0 = nothing), every lane gets it: cross-lane moves only
__device__ inline uint32_t wave_max_u32_fast(uint32_t v)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = v > r[1] ? v : r[1];
    r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = v > r[1] ? v : r[1];
    uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x108, 0xf, 0xf, true);
    v = v > o ? v : o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0xf, true);
    v = v > o ? v : o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xf, 0xf, true);
    v = v > o ? v : o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);
    v = v > o ? v : o;
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

template <bool BM_LDS, int NV>
__global__ __launch_bounds__(WAVE) void query_descend_kernel(QueryParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // a queue entry = (bound as an orderable 32-bit word, node); an empty slot has bound word 0 (no real bound maps to 0:
    // f32_orderable() of a non-NaN float is >= 0x00800000 ... and the words of the negative floats are the complements)
    __shared__ uint32_t s_bound[PQ_LDS];
    __shared__ int32_t s_node[PQ_LDS];
    __shared__ int32_t s_slot[PQ_LDS];    // hyperplane slot of the entry's node, -1: a leaf
    __shared__ int s_ncand;
    const int lane = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const int nvec = P.dpad / 4;
    const int T = P.n_trees;
    uint32_t *bm = BM_LDS ? (uint32_t *)smem : P.bm_global + (size_t)qi * P.bm_words;
    const float4 *q = query_row(P, qi);
    uint64_t *gpq = P.pq + qi * P.n_nodes;
    int32_t *cand = P.cand + qi * P.cap;
    const bool roots_split = P.n_items > P.K;
    int hn = roots_split ? 2 * T : T;
#ifdef MORNA_DESCEND_PROBE
    long long tp[40];
    int ntp = 0;
#define TP() do { if (ntp < 40) tp[ntp++] = clock64(); } while (0)
#else
#define TP() do { } while (0)
#endif
    TP();
    float4 qr[NV > 0 ? NV : 1];
#pragma unroll
    for (int k = 0; k < NV; k++) qr[k] = q[lane + WAVE * k];
    {
        // the queue as the root margins left it, and the hyperplane slot of every entry's node: all the loads of a phase in
        // flight together (written as one loop, every iteration waited for its two dependent round trips)
        auto init = [&](auto NU) {
            constexpr int nu = decltype(NU)::value;
            uint64_t k16[nu];
            int32_t s16[nu];
#pragma unroll
            for (int u = 0; u < nu; u++) {
                const int i = lane + WAVE * u;
                k16[u] = i < hn ? (roots_split ? gpq[i] : pq_key(INFINITY, i)) : 0;
            }
#pragma unroll
            for (int u = 0; u < nu; u++) s16[u] = lane + WAVE * u < hn ? P.node_hp[(int32_t)(uint32_t)k16[u]] : -1;
#pragma unroll
            for (int u = 0; u < nu; u++) {
                s_bound[lane + WAVE * u] = (uint32_t)(k16[u] >> 32);   // (0 beyond hn: an empty slot)
                s_node[lane + WAVE * u] = (int32_t)(uint32_t)k16[u];
                s_slot[lane + WAVE * u] = s16[u];
            }
        };
        if (hn <= 8 * WAVE) {
            init(std::integral_constant<int, 8>());
            for (int i = 8 * WAVE + lane; i < PQ_LDS; i += WAVE) s_bound[i] = 0;   // slots the pushes will fill
        } else {
            init(std::integral_constant<int, PQ_LDS / WAVE>());
        }
    }
    if (!roots_split)
        for (int i = PQ_LDS + lane; i < hn; i += WAVE) gpq[i] = pq_key(INFINITY, i);
    for (int i = lane; i < P.bm_words; i += WAVE) bm[i] = 0u;
    if (lane == 0) s_ncand = 0;
    __syncthreads();   // (one wave: orders the LDS / global initialisation in front of the loop)
    int ndots = roots_split ? T : 0, n_leaves = 0;
    int64_t nn = 0, cand_off = -1;
    const int64_t search_k = P.search_k;
    TP();
    while (nn < search_k) {
        // ---- the entry with the largest (bound, node)
        int pos = -1;
        uint32_t top_b = 0;
        {
            const int hl = hn < PQ_LDS ? hn : PQ_LDS;
            auto pick = [&](auto NU) {
                constexpr int nu = decltype(NU)::value;
                uint32_t bb[nu], mb = 0;
#pragma unroll
                for (int u = 0; u < nu; u++) bb[u] = s_bound[lane + WAVE * u];   // (slots beyond hn hold 0)
#pragma unroll
                for (int u = 0; u < nu; u++) mb = bb[u] > mb ? bb[u] : mb;
                for (int i = PQ_LDS + lane; i < hn; i += WAVE) {   // (entries beyond the LDS arrays: global)
                    const uint32_t b = (uint32_t)(gpq[i] >> 32);
                    mb = b > mb ? b : mb;
                }
                top_b = wave_max_u32_fast(mb);
                if (top_b == 0) return;               // queue empty
                // who holds it?  One entry as a rule; several (equal bounds, e.g. both children behind a zero margin): annoy's
                // pair order pops the larger node id first
                int cnt = 0, mypos = -1;
#pragma unroll
                for (int u = 0; u < nu; u++)
                    if (bb[u] == top_b) { cnt++; mypos = lane + WAVE * u; }
                const unsigned long long holders = __ballot(cnt > 0);
                if (hn <= PQ_LDS && __popcll(holders) == 1 && __builtin_amdgcn_readlane(cnt, __ffsll((long long)holders) - 1) == 1) {
                    pos = __builtin_amdgcn_readlane(mypos, __ffsll((long long)holders) - 1);
                    return;
                }
                int32_t mynode = -1;                  // the slow way: the largest node among the holders
                mypos = -1;
                for (int i = lane; i < hn; i += WAVE) {
                    const bool il = i < PQ_LDS;
                    const uint32_t b = il ? s_bound[i] : (uint32_t)(gpq[i] >> 32);
                    if (b == top_b) {
                        const int32_t nd = il ? s_node[i] : (int32_t)(uint32_t)gpq[i];
                        if (nd > mynode) { mynode = nd; mypos = i; }
                    }
                }
                const uint32_t best_node = wave_max_u32_fast(mypos >= 0 ? (uint32_t)mynode + 1u : 0u) - 1u;   // node ids are >= 0
                const unsigned long long w2 = __ballot(mypos >= 0 && (uint32_t)mynode == best_node);
                pos = __builtin_amdgcn_readlane(mypos, __ffsll((long long)w2) - 1);
            };
            if (hl <= 8 * WAVE) pick(std::integral_constant<int, 8>());
            else pick(std::integral_constant<int, PQ_LDS / WAVE>());
            if (top_b == 0) break;
        }
        TP();
        const bool in_lds = pos < PQ_LDS;
        const int32_t node = in_lds ? s_node[pos] : (int32_t)(uint32_t)gpq[pos];
        const float d = f32_from_orderable(top_b);
        if (lane == 0) {
            if (in_lds) s_bound[pos] = 0;
            else gpq[pos] = 0;
        }
        const int32_t slot = in_lds ? s_slot[pos] : P.node_hp[node];
        if (slot < 0) {
            // leaf: nns.insert(all ids); duplicates across trees are dropped by the bitmap
            const int4 rec = *(const int4 *)(P.node_rec + 4 * (int64_t)node);
            const int32_t *src_ids = P.perm + (int64_t)P.node_tree[node] * P.n_items + rec.z;
            const int count = rec.w;
            if (P.zero_copy && n_leaves == 0 && nn + count >= search_k) {
                // the first leaf ends the search: its ids ARE the candidates -- not copied
                if (lane == 0) {
                    s_ncand = count;
                    cand_off = (int64_t)P.node_tree[node] * P.n_items + rec.z;
                }
            } else {
                for (int base = 0; base < count; base += 8 * WAVE) {
                    int32_t idv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int i = base + u * WAVE + lane;
                        idv[u] = i < count ? src_ids[i] : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        bool fresh = false;
                        if (idv[u] >= 0) {
                            const uint32_t bit = 1u << (idv[u] & 31);
                            fresh = !(atomicOr(&bm[idv[u] >> 5], bit) & bit);
                        }
                        const unsigned long long mask = __ballot(fresh);
                        if (mask) {   // one counter update per 64 ids: the fresh ones take consecutive slots
                            int first = 0;
                            if (lane == 0) first = atomicAdd(&s_ncand, __popcll(mask));
                            first = __builtin_amdgcn_readfirstlane(first);
                            const int slot_c = first + __popcll(mask & ((1ull << lane) - 1ull));
                            if (fresh && slot_c < P.cap) cand[slot_c] = idv[u];
                        }
                    }
                }
            }
            nn += count;
            n_leaves++;
        } else {
            const int4 rec = *(const int4 *)(P.node_rec + 4 * (int64_t)node);
            // the children's slots are wanted when they are popped, not now: requested here, under the hyperplane fetch
            const int32_t slot_c = lane < 2 ? P.node_hp[lane == 0 ? rec.y : rec.x] : 0;
            const float4 *hrow = (const float4 *)(P.hp + (int64_t)slot * P.dpad);
            const float m = NV > 0 ? wave_dot_held<NV>(hrow, qr, lane) : wave_dot(hrow, q, nvec, lane);
            if (lane < 2) {
                const uint64_t key = lane == 0 ? pq_key(d < m ? d : m, rec.y) : pq_key(d < -m ? d : -m, rec.x);
                if (hn + lane < PQ_LDS) {
                    s_bound[hn + lane] = (uint32_t)(key >> 32);
                    s_node[hn + lane] = (int32_t)(uint32_t)key;
                    s_slot[hn + lane] = slot_c;
                } else {
                    gpq[hn + lane] = key;
                }
            }
            hn += 2;
            ndots++;
        }
        __syncthreads();   // one wave: the queue writes above are seen by the next scan
        TP();
    }
    if (lane == 0) {
        const int nc = s_ncand;
        P.ncand[qi] = nc;
        P.cand_off[qi] = cand_off;
        atomicAdd(P.stat, (unsigned long long)(ndots + (nc < P.cap ? nc : P.cap) + 1));
    }
#ifdef MORNA_DESCEND_PROBE
    TP();
    if (lane == 0 && qi == 0) {
        float *o = P.low + 2 * gridDim.x + 16;
        o[0] = (float)ntp;
        for (int i = 1; i < ntp; i++) o[i] = (float)(tp[i] - tp[i - 1]);
    }
#endif
#undef TP
}

#define QS_CPW 1   // candidates per wave of query_cand_dots_kernel
__global__ __launch_bounds__(Q_THREADS) void query_cand_dots_kernel(QueryParams P)
{
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int64_t qi = blockIdx.y;
    const int nvec = P.dpad / 4;
    const int ncand = P.ncand[qi] < P.cap ? P.ncand[qi] : P.cap;
    const int c0 = (blockIdx.x * Q_WAVES + w) * QS_CPW;
    if (c0 >= ncand) return;
    const float4 *q = query_row(P, qi);
    const int64_t off = P.cand_off[qi];
    const int32_t *cand = off >= 0 ? P.perm + off : P.cand + qi * P.cap;
    uint64_t *keys = P.keys + qi * P.cap;
    const float pp = P.qpp[qi];
#pragma unroll
    for (int j = 0; j < QS_CPW; j++) {
        const int c = c0 + j;
        if (c < ncand) {
            const int32_t id = cand[c];
            const float pqv = wave_dot((const float4 *)(P.X + (int64_t)id * P.dpad), q, nvec, lane);
            if (lane == 0) keys[c] = ((uint64_t)f32_orderable(ang_dist(pp, P.norm2[id], pqv)) << 32) | (uint32_t)id;
        }
    }
}

// The k smallest (distance, id) keys of a query's candidates, in order.  Two looks at the keys instead of k rounds of a
// block-wide minimum: every thread's own minimum -- the k-th smallest of those 256 bounds the k-th smallest key from above --,
// then the few keys at or below that bound are collected and ranked by counting (the keys are distinct: the id is in them).
#define QT_HELD 16
#define QT_LIST 1024
__global__ __launch_bounds__(Q_THREADS) void query_topk_kernel(QueryParams P)
{
    __shared__ uint64_t s_red[Q_WAVES];
    __shared__ uint64_t s_min[Q_THREADS];
    __shared__ uint64_t s_list[QT_LIST];
    __shared__ uint64_t s_bound;
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const uint64_t *keys = P.keys + qi * P.cap;
    const int nsel = P.ncand[qi] < P.cap ? P.ncand[qi] : P.cap;
    const int kout = P.k < nsel ? P.k : nsel;
    auto emit = [&](int r, uint64_t key) {
        P.ids_out[qi * P.k + r] = (int32_t)(uint32_t)key;
        const float d = f32_from_orderable((uint32_t)(key >> 32));
        P.dist_out[qi * P.k + r] = sqrtf(d > 0.f ? d : 0.f);   // normalized_distance
    };
    bool done = false;
    if (nsel <= Q_THREADS * QT_HELD && kout > 0 && kout <= Q_THREADS) {
        uint64_t held[QT_HELD], mine = ~0ull;
#pragma unroll
        for (int u = 0; u < QT_HELD; u++) {
            held[u] = tid + Q_THREADS * u < nsel ? keys[tid + Q_THREADS * u] : ~0ull;
            mine = held[u] < mine ? held[u] : mine;
        }
        s_min[tid] = mine;
        if (tid == 0) { s_n = 0; s_bound = ~0ull; }
        __syncthreads();
        if (nsel <= Q_THREADS) {
            if (tid == 0) s_bound = ~0ull - 1;   // a key per thread at most: all of them are collected
        } else {
            int below = 0;
            for (int j = 0; j < Q_THREADS; j++) below += s_min[j] < mine ? 1 : 0;
            if (mine != ~0ull && below == kout - 1) s_bound = mine;   // kout threads hold a key at or below it
        }
        __syncthreads();
        const uint64_t bound = s_bound;
#pragma unroll
        for (int u = 0; u < QT_HELD; u++)
            if (held[u] <= bound && held[u] != ~0ull) {
                const int slot = atomicAdd(&s_n, 1);
                if (slot < QT_LIST) s_list[slot] = held[u];
            }
        __syncthreads();
        const int m = s_n;
        if (m <= QT_LIST) {
            for (int i = tid; i < m; i += Q_THREADS) {
                const uint64_t v = s_list[i];
                int below = 0;
                for (int j = 0; j < m; j++) below += s_list[j] < v ? 1 : 0;
                if (below < kout) emit(below, v);
            }
            done = true;
        }
    }
    if (!done) {   // more keys than the registers hold, or a tie of thousands at the bound: k rounds of a block-wide minimum
        uint64_t prev = 0;
        for (int r = 0; r < kout; r++) {
            uint64_t best = ~0ull;
            for (int c = tid; c < nsel; c += Q_THREADS) {
                const uint64_t kk = keys[c];
                if ((r == 0 || kk > prev) && kk < best) best = kk;
            }
            best = block_min_u64(best, s_red, tid);
            if (tid == 0) emit(r, best);
            prev = best;
        }
    }
    for (int r = kout + tid; r < P.k; r += Q_THREADS) {
        P.ids_out[qi * P.k + r] = -1;
        P.dist_out[qi * P.k + r] = INFINITY;
    }
    if (tid == 0) P.count_out[qi] = kout;
}

// ---- filter dots of a whole batch on the matrix cores ----------------------------------------------------------
// scores[q][r] = sum_i g_q[i] y_r[i] for ALL rows r and all queries q of the batch, from the fp16 images: a
// nq x N x dpad contraction that reads the fp16 matrix once (0.3 GB at 50k x 3000) where the per-query form gathers
// every query's ~K candidate rows separately (12 GB for 1000 queries).  It does ~N / K times the arithmetic the
// candidates need, which the matrix cores deliver in less time than the gathers take while
// nq * K >> N.  Queries are the A side (m), rows the B side (n): 32 lanes hold 32 consecutive rows of one query, so
// the result goes out in 128-byte runs.  qrow: row of Q16 that holds query q (by-item queries point into the
// matrix's own image), or null for the identity.
template <int BR, int BK, int NS>   // rows of the matrix per workgroup tile, halfs per K-step, stages in flight (mm16.hpp)
__global__ __launch_bounds__(MM16_THREADS) void query_scores_kernel(const _Float16 *__restrict__ X16, int64_t n_items, int32_t dpad,
                                                                    const _Float16 *__restrict__ Q16,
                                                                    const int32_t *__restrict__ qrow, int32_t nq,
                                                                    float *__restrict__ scores, int64_t r_begin /* rows [r_begin, n_items) */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int32_t s_qrow[MM16_TILE];
    constexpr int NB = BR / 128;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int wm = w >> 1, wn = w & 1;
    // workgroup b runs on XCD b % 8: an XCD takes every 8th row tile and walks that tile's query tiles back to back,
    // so the row tile comes from HBM once and then from that XCD's L2
    const int n_ct = (nq + MM16_TILE - 1) / MM16_TILE;
    const int64_t row_tile = (int64_t)((blockIdx.x >> 3) / n_ct) * 8 + (blockIdx.x & 7);
    const int64_t r0 = r_begin + row_tile * BR;
    const int c0 = (int)((blockIdx.x >> 3) % n_ct) * MM16_TILE;
    if (r0 >= n_items) return;
    if (tid < MM16_TILE) {
        const int q = c0 + tid < nq ? c0 + tid : nq - 1;
        s_qrow[tid] = qrow ? qrow[q] : q;
    }
    __syncthreads();
    mm16_f32x16 acc[NB][2];
    mm16_tile<BR, BK, NS>(X16, Q16, dpad, smem, [&](int rt) { return r0 + rt < n_items ? r0 + rt : n_items - 1; },
                          [&](int rt) { return (int64_t)s_qrow[rt]; }, acc);
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int tb = 0; tb < NB; tb++) {
        const int64_t r = r0 + wm * 32 * NB + tb * 32 + lr;
        if (r < n_items) {
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int q = c0 + wn * 64 + tn * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (q < nq) scores[(int64_t)q * n_items + r] = acc[tb][tn][e];
                }
        }
    }
}

// The same product in 256 x 256 tiles, 16 waves, ONE accumulation chain (mm16.hpp): half the operand bytes delivered to LDS
// per product, which is what bounds the loop.  Its workgroups take the whole CU, so it covers only as many 256-row tiles as
// fill the chip in whole rounds (the host picks n_row_tiles); the 128 x 128 form above takes the rows behind them.  The
// bound on its accumulation is the one-chain EACC (QueryParams::eacc_big).
template <int BK, int NS>
__global__ __launch_bounds__(1024) void query_scores_big_kernel(const _Float16 *__restrict__ X16, int64_t n_items, int32_t dpad,
                                                                const _Float16 *__restrict__ Q16, const int32_t *__restrict__ qrow,
                                                                int32_t nq, float *__restrict__ scores, int32_t n_row_tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int32_t s_qrow[256];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int wm = w >> 2, wn = w & 3;   // the wave's part of the tile: rows wm * 64 .., queries wn * 64 ..
    const int n_ct = (nq + 255) / 256;
    const int64_t row_tile = (int64_t)((blockIdx.x >> 3) / n_ct) * 8 + (blockIdx.x & 7);   // XCD b % 8 keeps its row tiles
    if (row_tile >= n_row_tiles) return;
    const int64_t r0 = row_tile * 256;   // (+ 255 < n_items: the host covers whole tiles only)
    const int c0 = (int)((blockIdx.x >> 3) % n_ct) * 256;
    if (tid < 256) {
        const int q = c0 + tid < nq ? c0 + tid : nq - 1;
        s_qrow[tid] = qrow ? qrow[q] : q;
    }
    __syncthreads();
    mm16_f32x16 acc[2][2];
    mm16_tile_256x256<BK, NS>(X16, Q16, dpad, smem, [&](int rt) { return r0 + rt; }, [&](int rt) { return (int64_t)s_qrow[rt]; }, acc);
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int tb = 0; tb < 2; tb++) {
        const int64_t r = r0 + wm * 64 + tb * 32 + lr;
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int q = c0 + wn * 64 + tn * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (q < nq) scores[(int64_t)q * n_items + r] = acc[tb][tn][e];
            }
    }
}

// ---- refine: angular distance to the candidates, k smallest (distance, id) --------------------------------------
// FILTER (modes 1 and 2): a distance taken from fp16 data is within delta of the canonical fp32 one, so the k
// smallest canonical distances are among the candidates whose lower bound does not exceed the k-th smallest upper
// bound; only those get the canonical wave_dot, and the ranking is done on canonical values alone.  Candidates
// without a usable fp16 value (inf, NaN, unscalable rows or queries) are never filtered out.
template <int MODE>   // 0: no filter; 1: fp16 row x fp32 query per candidate (gather); 2: scores[] of the whole batch
__global__ __launch_bounds__(Q_THREADS) void query_refine_kernel(QueryParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *qv = (float4 *)smem;
    __shared__ uint64_t s_red[Q_WAVES];
    __shared__ uint64_t s_top[Q_WAVES * KTH_MAX_K];
    __shared__ int s_nsurv;

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int64_t qi = blockIdx.x;
    const int nvec = P.dpad / 4;
    const float4 *src = P.items ? (const float4 *)(P.X + (int64_t)P.items[qi] * P.dpad)
                                : (const float4 *)(P.Q + qi * P.dpad);
    for (int i = tid; i < nvec; i += Q_THREADS) qv[i] = src[i];
    if (tid == 0) s_nsurv = 0;
    __syncthreads();

    int32_t *cand = P.cand + qi * P.cap;
    uint64_t *keys = P.keys + qi * P.cap;
    const int ncand = P.ncand[qi] < P.cap ? P.ncand[qi] : P.cap;
    const float pp = P.qpp[qi];
    int nsel = ncand;          // candidates that get the canonical dot; their ids in cand[0 .. nsel)

    if (MODE != 0 && ncand > P.k) {
        float *low = P.low + qi * P.cap;
        const float nan = __int_as_float(0x7fc00000);
        if (MODE == 1) {
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            const int nv8 = P.dpad / 8;
            for (int c = w; c < ncand; c += Q_WAVES) {
                const int32_t id = cand[c];
                const f16x8 *y = (const f16x8 *)(P.X16 + (int64_t)id * P.dpad);
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                auto mac8 = [&](const f16x8 hv, int i) {
                    const float4 qa = qv[2 * i], qb = qv[2 * i + 1];
                    s0 = fmaf((float)hv[0], qa.x, s0);
                    s1 = fmaf((float)hv[1], qa.y, s1);
                    s2 = fmaf((float)hv[2], qa.z, s2);
                    s3 = fmaf((float)hv[3], qa.w, s3);
                    s0 = fmaf((float)hv[4], qb.x, s0);
                    s1 = fmaf((float)hv[5], qb.y, s1);
                    s2 = fmaf((float)hv[6], qb.z, s2);
                    s3 = fmaf((float)hv[7], qb.w, s3);
                };
                int i = lane;
                for (; i + 5 * WAVE < nv8; i += 6 * WAVE) {   // six 1-KiB loads in flight before the first use
                    const f16x8 h0 = y[i], h1 = y[i + WAVE], h2 = y[i + 2 * WAVE], h3 = y[i + 3 * WAVE];
                    const f16x8 h4 = y[i + 4 * WAVE], h5 = y[i + 5 * WAVE];
                    mac8(h0, i);
                    mac8(h1, i + WAVE);
                    mac8(h2, i + 2 * WAVE);
                    mac8(h3, i + 3 * WAVE);
                    mac8(h4, i + 4 * WAVE);
                    mac8(h5, i + 5 * WAVE);
                }
                for (; i < nv8; i += WAVE) mac8(y[i], i);
                const float sc = P.xscale[id];   // 0: the row has no fp16 image (inf, NaN or below the scalable range)
                const float dotf = wave_sum_xor((s0 + s1) + (s2 + s3)) * sc;
                const float df = sc != 0.f ? ang_dist(pp, P.norm2[id], dotf) : nan;
                if (lane == 0) {
                    // a NaN sorts after every number: it takes none of the k places that set the threshold
                    keys[c] = ((uint64_t)f32_orderable(df + P.delta) << 32) | (uint32_t)c;
                    low[c] = df - P.delta;
                }
            }
        } else {
            const int32_t qsrc = P.items ? P.items[qi] : (int32_t)qi;
            const float qs = P.qscale[qsrc], qn = P.qn16[qsrc], qe = P.qe16[qsrc];
            const float rq = qe / (qn * 0.996f);          // |f| / |g|, |g| from its upper bound less the slack put on it
            const float *sc_row = P.scores + qi * P.n_items;
            for (int c = tid; c < ncand; c += Q_THREADS) {
                const int32_t id = cand[c];
                const float sc = P.xscale[id] * qs;
                const float rx = P.xe16[id] / (P.xn16[id] * 0.996f);
                float df = nan, dl = 0.f;
                if (sc != 0.f && rx < 0.125f && rq < 0.125f) {
                    // |sum y g - s t x.q| <= (EACC |y| + |d|) |g| + (|y| + |d|) |f|, and s |x| >= |y| - |d|, t |q| >= |g| - |f|:
                    // the cosine taken from the images is within this of the exact one
                    const float cerr = (((id < P.big_rows ? P.eacc_big : P.eacc) + rx) + (1.f + rx) * rq) / ((1.f - rx) * (1.f - rq));
                    df = ang_dist(pp, P.norm2[id], sc_row[id] * sc);
                    dl = 2.02f * cerr + 1e-5f;           // distance = 2 - 2 cos; the canonical norms and ang_dist's own rounding
                }
                keys[c] = ((uint64_t)f32_orderable(df + dl) << 32) | (uint32_t)c;
                low[c] = df - dl;
            }
        }
        __syncthreads();
        // the k-th smallest upper bound (keys are distinct: the candidate index is in them)
        const uint64_t kth = block_kth_smallest(keys, ncand, P.k, tid, s_red, s_top);
        const float thr = f32_from_orderable((uint32_t)(kth >> 32));
        // survivors, compacted to the front of keys[] as ids (order is irrelevant: the ranking below is by value)
        __syncthreads();
        for (int c0 = 0; c0 < ncand; c0 += Q_THREADS) {
            const int c = c0 + tid;
            const bool in = c < ncand && !(low[c] > thr);   // NaN stays in
            const int32_t id = c < ncand ? cand[c] : 0;
            __syncthreads();                                // every cand[c] of this round is read before a slot is written
            if (in) cand[atomicAdd(&s_nsurv, 1)] = id;      // slots < c0 + Q_THREADS: none beyond the round just read
            __syncthreads();
        }
        nsel = s_nsurv;
    }
    __syncthreads();
    for (int c = w; c < nsel; c += Q_WAVES) {
        const int32_t id = cand[c];
        const float pqv = wave_dot((const float4 *)(P.X + (int64_t)id * P.dpad), qv, nvec, lane);
        if (lane == 0) keys[c] = ((uint64_t)f32_orderable(ang_dist(pp, P.norm2[id], pqv)) << 32) | (uint32_t)id;
    }
    __syncthreads();

    // ---- k smallest (distance, id) pairs, in order
    const int kout = P.k < nsel ? P.k : nsel;
    uint64_t prev = 0;
    bool have_prev = false;
    if (nsel <= WAVE * KTH_HELD) {
        // few enough for one wave's registers (the usual case: ~2k survivors): k rounds of register compares and a wave
        // minimum, no barrier
        if (w == 0) {
            uint64_t held[KTH_HELD];
#pragma unroll
            for (int u = 0; u < KTH_HELD; u++) held[u] = lane + WAVE * u < nsel ? keys[lane + WAVE * u] : ~0ull;
            for (int r = 0; r < kout; r++) {
                uint64_t best = ~0ull;
#pragma unroll
                for (int u = 0; u < KTH_HELD; u++)
                    if ((r == 0 || held[u] > prev) && held[u] < best) best = held[u];
                prev = wave_min_u64(best);
                if (lane == 0) {
                    P.ids_out[qi * P.k + r] = (int32_t)(uint32_t)prev;
                    const float d = f32_from_orderable((uint32_t)(prev >> 32));
                    P.dist_out[qi * P.k + r] = sqrtf(d > 0.f ? d : 0.f);   // normalized_distance
                }
            }
        }
    } else
    for (int r = 0; r < kout; r++) {
        uint64_t best = ~0ull;
        for (int c = tid; c < nsel; c += Q_THREADS) {
            const uint64_t kk = keys[c];
            if ((!have_prev || kk > prev) && kk < best) best = kk;
        }
        best = block_min_u64(best, s_red, tid);
        if (tid == 0) {
            P.ids_out[qi * P.k + r] = (int32_t)(uint32_t)best;
            const float d = f32_from_orderable((uint32_t)(best >> 32));
            P.dist_out[qi * P.k + r] = sqrtf(d > 0.f ? d : 0.f);   // normalized_distance
        }
        prev = best;
        have_prev = true;
    }
    for (int r = kout + tid; r < P.k; r += Q_THREADS) {
        P.ids_out[qi * P.k + r] = -1;
        P.dist_out[qi * P.k + r] = INFINITY;
    }
    if (tid == 0) P.count_out[qi] = kout;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// [n][k] ids / distances of a batch -> the [n][2k] int32 message of the top-k all-gather: global ids, distance bits
__global__ void pack_topk_kernel(const int32_t *__restrict__ ids, const float *__restrict__ dist, int64_t total, int32_t k,
                                 int32_t id_offset, int32_t *__restrict__ packed)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t q = i / k;
    const int r = (int)(i - q * k);
    const int32_t id = ids[i];
    packed[q * 2 * k + r] = id >= 0 ? id + id_offset : -1;
    packed[q * 2 * k + k + r] = __float_as_int(dist[i]);
}

// q_host: query vectors in host OR device memory (unified addressing), q_stride floats apart (0 = dim)
int query_batch(morna_index *h, const float *q_host, int64_t q_stride, const int32_t *items_host, int64_t nq, int32_t k,
                int32_t search_k, int32_t *ids_out, float *dist_out, int32_t *count_out, int32_t *packed_dev, int64_t id_offset)
{
    if (q_stride <= 0) q_stride = h->dim;
    if (!h->built) {
        set_error("index has not been built: call build() before searching");
        return MORNA_E_STATE;
    }
    if (k <= 0 || nq < 0) {
        set_error("get_nns: n must be positive");
        return MORNA_E_INVALID;
    }
    if (nq == 0) return MORNA_OK;
    if (search_k == -1) search_k = (int32_t)std::min<int64_t>((int64_t)k * h->n_trees, INT32_MAX);
    if (items_host)
        for (int64_t i = 0; i < nq; i++)
            if (items_host[i] < 0 || items_host[i] >= h->n_items) {
                set_error("item id %d out of range [0, %lld)", items_host[i], (long long)h->n_items);
                return MORNA_E_RANGE;
            }
    const int64_t N = h->n_items;
    // |nns| < search_k before the last pop, which adds at most K ids
    const int64_t cap64 = std::min<int64_t>(N, (int64_t)std::max(search_k, 0) + h->K);
    const int32_t cap = (int32_t)std::max<int64_t>(cap64, 1);
    const int32_t bm_words = (int32_t)((N + 31) / 32);
    const bool bm_lds = (size_t)bm_words * 4 <= 64 * 1024;
    const size_t lds_q = (size_t)h->dpad * sizeof(float);
    const size_t lds = lds_q + (bm_lds ? (size_t)bm_words * 4 : 0);

    // candidate filter on the fp16 rows (MORNA_QUERY_FILTER=0: every candidate gets the fp32 dot); pays when a
    // query has many more candidates than results.  MORNA_QUERY_DENSE=0 keeps the per-candidate gather form.
    static const bool filter_on = !(getenv("MORNA_QUERY_FILTER") && atoi(getenv("MORNA_QUERY_FILTER")) == 0);
    static const bool dense_on = !(getenv("MORNA_QUERY_DENSE") && atoi(getenv("MORNA_QUERY_DENSE")) == 0);
    // fewer queries than would give every CU a workgroup: the spread form (MORNA_QUERY_SPREAD=0: one workgroup per query)
    const bool spread_on = !(getenv("MORNA_QUERY_SPREAD") && atoi(getenv("MORNA_QUERY_SPREAD")) == 0);
    const bool filter_pays = filter_on && cap > 4 * (int64_t)k;
    // the whole-batch contraction reads every fp16 row once; the gather form reads min(cap, ~K) rows per query
    // (C3, 48 / 63 queries: 207 us through the contraction, 231 / 282 us through the spread form; 32: the spread form's 174 us)
    auto dense_pays = [&](int64_t nb) { return nb >= 40 && nb * std::min<int64_t>(cap, h->K) >= 2 * N; };
    const bool spread = spread_on && nq < 64 && !(filter_pays && dense_on && dense_pays(nq));
    const bool use_filter = !spread && filter_pays;
    const bool may_dense = use_filter && dense_on && dense_pays(nq);

    const size_t per_q = (size_t)h->n_nodes * 8 + (size_t)cap * 16 + (q_host ? (size_t)h->dpad * 6 + 16 : 0) +
                         (bm_lds ? 0 : (size_t)bm_words * 4) + (size_t)k * 8 + 16 + (may_dense ? (size_t)N * 4 : 0);
    const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(nq, (int64_t)(((size_t)1 << 30) / per_q)));
    const size_t s_pq = align_up((size_t)batch * h->n_nodes * 8, 256), s_keys = align_up((size_t)batch * cap * 8, 256),
                 s_q = q_host ? align_up((size_t)batch * h->dpad * 4, 256) : 0,
                 s_q16 = q_host && may_dense ? align_up((size_t)batch * h->dpad * 2, 256) : 0,
                 s_qf = align_up((size_t)batch * 4, 256),
                 s_cand = align_up((size_t)batch * cap * 4, 256),
                 s_bm = bm_lds ? 0 : align_up((size_t)batch * bm_words * 4, 256),
                 s_ids = align_up((size_t)batch * k * 4, 256),
                 s_scores = may_dense ? align_up((size_t)batch * N * 4, 256) : 0;
    MORNA_TRY(h->ws.alloc(s_pq + s_keys + s_q + s_q16 + 9 * s_qf + 2 * s_cand + s_bm + 2 * s_ids + s_scores));
    MORNA_TRY(h->d_stat.alloc(4));
    if (use_filter) MORNA_TRY(split_mm_prepare_rows(h, h->stream));

    for (int64_t q0 = 0; q0 < nq; q0 += batch) {
        const int64_t nb = std::min(batch, nq - q0);
        const bool dense = may_dense && dense_pays(nb);
        uint8_t *p = h->ws.p;
        QueryParams P;
        P.X = h->X.p; P.norm2 = h->norm2.p; P.n_items = N; P.dpad = h->dpad; P.n_trees = h->n_trees; P.K = h->K;
        P.node_rec = h->node_rec.p; P.node_tree = h->node_tree.p; P.node_hp = h->node_hp.p; P.hp = h->hp.p;
        P.perm = h->perm.p; P.n_nodes = h->n_nodes;
        P.k = k; P.search_k = search_k; P.cap = cap; P.bm_words = bm_words;
        P.pq = (uint64_t *)p; p += s_pq;
        P.keys = (uint64_t *)p; p += s_keys;
        float *Qd = (float *)p; p += s_q;
        _Float16 *Q16 = (_Float16 *)p; p += s_q16;
        P.qpp = (float *)p; p += s_qf;
        float *q_n16 = (float *)p; p += s_qf;
        float *q_e16 = (float *)p; p += s_qf;
        float *q_scale = (float *)p; p += s_qf;
        P.ncand = (int32_t *)p; p += s_qf;
        int32_t *d_items = (int32_t *)p; p += s_qf;   // (part of the workspace: a hipMalloc / hipFree per call costs tens of microseconds)
        P.cand_off = (int64_t *)p; p += 2 * s_qf;
        P.zero_copy = spread ? 1 : 0;
        P.cand = (int32_t *)p; p += s_cand;
        P.low = (float *)p; p += s_cand;
        P.bm_global = (uint32_t *)p; p += s_bm;
        uint8_t *const out_block = p;   // ids, distances, counts: one block, one copy to the host
        P.ids_out = (int32_t *)p; p += s_ids;
        P.dist_out = (float *)p; p += s_ids;
        P.count_out = (int32_t *)p; p += s_qf;
        float *scores = (float *)p; p += s_scores;
        P.stat = h->d_stat.p;
        P.X16 = nullptr; P.xscale = nullptr; P.xn16 = P.xe16 = nullptr; P.delta = 0.f;
        P.scores = nullptr; P.qscale = P.qn16 = P.qe16 = nullptr; P.eacc = P.eacc_big = 0.f; P.big_rows = 0;
        if (use_filter) {
            const float *xn = (const float *)h->scratch[20].p;
            P.X16 = (const _Float16 *)h->scratch[19].p;
            P.xn16 = xn; P.xscale = xn + N; P.xe16 = xn + 2 * N;
            // gather form: |cos from the fp16 row - cos from the fp32 row| <= 2^-11 (one rounding to 11 bits, the query
            // is not rounded) + the fp32 accumulations of both dots (any order: < dpad * 2^-24 each); the distance is
            // 2 - 2 cos; 1e-5 covers the evaluation of ang_dist itself
            P.delta = 2.f * (1.02f * 0.00048828125f + 2.f * (float)h->dpad * 5.9604645e-8f) + 1e-5f;
            P.eacc = mm16_eacc(h->dpad);
        }
        P.Q = nullptr; P.items = nullptr;
        P.n_inline = 0;
        if (spread && !q_host && nb <= 16) {
            P.n_inline = (int32_t)nb;
            for (int64_t i = 0; i < nb; i++) P.inline_items[i] = items_host[q0 + i];
        } else
        if (q_host) {
            bool staged = false;
            if (spread && ids_out && !packed_dev) {   // (the batch ends with a host wait: the staging buffer is free again by then)
                // a few query vectors from host memory: padded on the host into page-locked memory and sent as ONE copy (a
                // memset + a 2-D copy out of pageable memory are two stream operations of ~10 us each)
                hipPointerAttribute_t at;
                const bool on_host = hipPointerGetAttributes(&at, q_host) != hipSuccess || at.type == hipMemoryTypeHost ||
                                     at.type == hipMemoryTypeUnregistered;
                (void)hipGetLastError();   // (an unregistered host pointer reports an error on some runtimes)
                if (on_host) {
                    const size_t need = (size_t)nb * h->dpad * 4;
                    if (need > h->host_q_cap) {
                        if (h->host_q) (void)hipHostFree(h->host_q);
                        h->host_q = nullptr;
                        h->host_q_cap = 0;
                        HIP_TRY(hipHostMalloc((void **)&h->host_q, need * 2, hipHostMallocDefault));
                        h->host_q_cap = need * 2;
                    }
                    float *hq = (float *)h->host_q;
                    for (int64_t i = 0; i < nb; i++) {
                        memcpy(hq + i * h->dpad, q_host + (q0 + i) * q_stride, (size_t)h->dim * 4);
                        memset(hq + i * h->dpad + h->dim, 0, (size_t)(h->dpad - h->dim) * 4);
                    }
                    HIP_TRY(hipMemcpyAsync(Qd, hq, need, hipMemcpyHostToDevice, h->stream));
                    staged = true;
                }
            }
            if (!staged) {
                HIP_TRY(hipMemsetAsync(Qd, 0, (size_t)nb * h->dpad * 4, h->stream));
                HIP_TRY(hipMemcpy2DAsync(Qd, (size_t)h->dpad * 4, q_host + q0 * q_stride, (size_t)q_stride * 4,
                                         (size_t)h->dim * 4, (size_t)nb, hipMemcpyDefault, h->stream));
            }
            P.Q = Qd;
        } else {
            HIP_TRY(hipMemcpyAsync(d_items, items_host + q0, (size_t)nb * 4, hipMemcpyHostToDevice, h->stream));
            P.items = d_items;
        }
        // the best-first descent, one wave per query (query_descend_kernel), on stream st
        auto launch_descend = [&](hipStream_t st) -> int {
            const size_t bl = bm_lds ? (size_t)bm_words * 4 : 0;
#define DESCEND(NVV)                                                                                                              \
    do {                                                                                                                          \
        if (bm_lds) {                                                                                                             \
            if (bl > 32 * 1024)                                                                                                   \
                HIP_TRY(hipFuncSetAttribute((const void *)query_descend_kernel<true, NVV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            (int)bl));                                                                            \
            hipLaunchKernelGGL((query_descend_kernel<true, NVV>), dim3((unsigned)nb), dim3(WAVE), bl, st, P);                     \
        } else {                                                                                                                  \
            hipLaunchKernelGGL((query_descend_kernel<false, NVV>), dim3((unsigned)nb), dim3(WAVE), 0, st, P);                     \
        }                                                                                                                         \
    } while (0)
            switch (h->dpad / 256) {   // float4 per lane of a row
            case 1: DESCEND(1); break;
            case 2: DESCEND(2); break;
            case 3: DESCEND(3); break;
            case 4: DESCEND(4); break;
            case 6: DESCEND(6); break;
            case 8: DESCEND(8); break;
            case 12: DESCEND(12); break;
            case 16: DESCEND(16); break;
            case 24: DESCEND(24); break;
            case 32: DESCEND(32); break;
            default: DESCEND(0); break;
            }
#undef DESCEND
            return MORNA_OK;
        };
        // spread form, answers wanted on the host: the last kernel writes them straight into page-locked memory of the handle
        // (a copy out of HBM is another ~8 us operation on the stream for 160 bytes per query)
        const bool direct = spread && ids_out && !packed_dev;
        if (direct) {
            const size_t need = 2 * s_ids + s_qf;
            if (need > h->host_small_cap) {
                if (h->host_small) (void)hipHostFree(h->host_small);
                h->host_small = nullptr;
                h->host_small_cap = 0;
                HIP_TRY(hipHostMalloc((void **)&h->host_small, need * 2, hipHostMallocMapped | hipHostMallocCoherent));
                h->host_small_cap = need * 2;
            }
            uint8_t *dp = nullptr;
            HIP_TRY(hipHostGetDevicePointer((void **)&dp, h->host_small, 0));
            P.ids_out = (int32_t *)dp;
            P.dist_out = (float *)(dp + s_ids);
            P.count_out = (int32_t *)(dp + 2 * s_ids);
        }
        if (spread) {
            // a batch too small to fill the chip with one workgroup per query: each query's work is dealt out (kernels above)
            ScopedTimer tm(h, MORNA_T_QUERY, 0);
            hipLaunchKernelGGL(query_roots_kernel, dim3((unsigned)((h->n_trees + Q_WAVES - 1) / Q_WAVES), (unsigned)nb), dim3(Q_THREADS), 0,
                               h->stream, P);
            MORNA_TRY(launch_descend(h->stream));
#ifdef MORNA_DESCEND_PROBE
            {
                float probe[40];
                HIP_TRY(hipStreamSynchronize(h->stream));
                HIP_TRY(hipMemcpy(probe, P.low + 2 * nb + 16, sizeof(probe), hipMemcpyDeviceToHost));
                fprintf(stderr, "[descend probe] cycles:");
                for (int i = 1; i < (int)probe[0] && i < 40; i++) fprintf(stderr, " %.0f", probe[i]);
                fprintf(stderr, "\n");
            }
#endif
            hipLaunchKernelGGL(query_cand_dots_kernel, dim3((unsigned)((cap + Q_WAVES * QS_CPW - 1) / (Q_WAVES * QS_CPW)), (unsigned)nb),
                               dim3(Q_THREADS), 0, h->stream, P);
            hipLaunchKernelGGL(query_topk_kernel, dim3((unsigned)nb), dim3(Q_THREADS), 0, h->stream, P);
        } else {
            // algorithmic bytes = 4*D*(hyperplane dots + unique candidates + 1) per
            // query; the kernel counts them into d_stat[0], resolve_timers() prices them
            ScopedTimer tm(h, MORNA_T_QUERY, 0);
            if (dense) {
                const _Float16 *q16 = P.X16;
                const int32_t *qrow = P.items;
                if (q_host) {
                    split_mm_convert_rows(h, Qd, nb, Q16, q_n16, q_e16, q_scale, h->stream);
                    q16 = Q16; qrow = nullptr;
                    P.qn16 = q_n16; P.qe16 = q_e16; P.qscale = q_scale;
                } else {
                    P.qn16 = P.xn16; P.qe16 = P.xe16; P.qscale = P.xscale;
                }
                HIP_TRY(hipEventRecord(h->ev_fork, h->stream));   // the query vectors (and their fp16 image) / item ids are in place
                HIP_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
                ScopedTimer tf(h, MORNA_T_QUERY_FILTER, 2 * (int64_t)nb * N * h->dpad);   // "bytes" = executed flops
                // 256 x 256 tiles for as many 256-row tiles as make whole rounds of the chip (one workgroup per CU), the
                // 128 x 128 form for the rows behind them (MORNA_QUERY_BIG=0: for all rows)
                static const bool big_on = !(getenv("MORNA_QUERY_BIG") && atoi(getenv("MORNA_QUERY_BIG")) == 0);
                int64_t big_tiles = 0;
                if (big_on && nb >= 512) {
                    const int64_t n_ct_big = (nb + 255) / 256;
                    int64_t a = h->n_cus, b = n_ct_big;   // tiles per round of the chip: n_cus / gcd(n_cus, n_ct_big)
                    while (b) { const int64_t t = a % b; a = b; b = t; }
                    const int64_t per_round = h->n_cus / a;
                    big_tiles = (N / 256) / per_round * per_round;
                }
                if (big_tiles > 0) {
                    const unsigned n_ct_big = (unsigned)((nb + 255) / 256);
                    HIP_TRY(hipFuncSetAttribute((const void *)query_scores_big_kernel<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 256 * 4));
                    hipLaunchKernelGGL((query_scores_big_kernel<64, 2>), dim3(8u * (unsigned)((big_tiles + 7) / 8) * n_ct_big), dim3(1024), 128 * 256 * 4,
                                       h->stream, P.X16, N, h->dpad, q16, qrow, (int32_t)nb, scores, (int32_t)big_tiles);
                    P.big_rows = big_tiles * 256;
                    P.eacc_big = (4.f * (float)h->dpad + 2.f) * 5.9604645e-8f + 4.1e-6f;   // EACC for ONE chain of dpad products
                }
                if (P.big_rows < N) {
                    const unsigned n_ct = (unsigned)((nb + MM16_TILE - 1) / MM16_TILE), n_rt = (unsigned)((N - P.big_rows + 127) / 128);
                    HIP_TRY(hipFuncSetAttribute((const void *)query_scores_kernel<128, 64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, MM16_LDS));
                    // (the rows behind the whole rounds of the 256 x 256 form: on the side stream, in front of the traversal,
                    // when the large form runs -- behind it on the main stream they were 0.05 ms of the batch's critical path)
                    hipLaunchKernelGGL((query_scores_kernel<128, 64, 2>), dim3(8u * ((n_rt + 7) / 8) * n_ct), dim3(MM16_THREADS), MM16_LDS,
                                       big_tiles > 0 ? h->stream2 : h->stream, P.X16, N, h->dpad, q16, qrow, (int32_t)nb, scores, P.big_rows);
                }
                P.scores = scores;
            }
            // beside the contraction (matrix cores, LDS) the traversal is a latency chain on one wave per query: it
            // runs on the side stream and the refine step waits for both
            hipStream_t ts = dense ? h->stream2 : h->stream;   // (forked above, before the contraction was enqueued)
            // Root margins by query groups (a hyperplane fetched once per QR_Q queries), then the one-wave descent of the
            // spread form -- while the rows have a length the register forms are built for and the roots do split;
            // otherwise (and with MORNA_QUERY_SPLIT_TRAVERSE=0) the fused kernel, one workgroup per query.
            const int nvq = h->dpad / 256;
            const bool nv_ok = h->dpad % 256 == 0 && (nvq == 1 || nvq == 2 || nvq == 3 || nvq == 4 || nvq == 6 || nvq == 8 || nvq == 12);
            const bool split_traverse = nv_ok && N > h->K && (size_t)QR_Q * h->dpad * 4 <= 128 * 1024 &&
                                        !(getenv("MORNA_QUERY_SPLIT_TRAVERSE") && atoi(getenv("MORNA_QUERY_SPLIT_TRAVERSE")) == 0);
            if (split_traverse) {
                const dim3 grid((unsigned)((nb + QR_Q - 1) / QR_Q), (unsigned)((h->n_trees + QR_TREES - 1) / QR_TREES));
                const size_t ql = (size_t)QR_Q * h->dpad * 4;
#define ROOTSB(NVV)                                                                                                      \
    do {                                                                                                                 \
        if (ql > 48 * 1024)                                                                                              \
            HIP_TRY(hipFuncSetAttribute((const void *)query_roots_batch_kernel<NVV>,                                      \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)ql));                           \
        hipLaunchKernelGGL((query_roots_batch_kernel<NVV>), grid, dim3(QR_THREADS), ql, ts, P, (int32_t)nb);             \
    } while (0)
                switch (nvq) {
                case 1: ROOTSB(1); break;
                case 2: ROOTSB(2); break;
                case 3: ROOTSB(3); break;
                case 4: ROOTSB(4); break;
                case 6: ROOTSB(6); break;
                case 8: ROOTSB(8); break;
                default: ROOTSB(12); break;
                }
#undef ROOTSB
                MORNA_TRY(launch_descend(ts));
            } else if (bm_lds) {
                if (lds > 48 * 1024)   // query image + sample bitmap can pass the default dynamic-LDS limit
                    HIP_TRY(hipFuncSetAttribute((const void *)query_traverse_kernel<true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(query_traverse_kernel<true>, dim3((unsigned)nb), dim3(Q_THREADS), lds, ts, P);
            } else {
                hipLaunchKernelGGL(query_traverse_kernel<false>, dim3((unsigned)nb), dim3(Q_THREADS), lds_q, ts, P);
            }
            if (dense) {
                HIP_TRY(hipEventRecord(h->ev_join, h->stream2));
                HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
            }
#define REFINE(MODE)                                                                                                         \
    do {                                                                                                                     \
        if (lds_q > 48 * 1024)                                                                                               \
            HIP_TRY(hipFuncSetAttribute((const void *)query_refine_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)lds_q));                                                                        \
        hipLaunchKernelGGL(query_refine_kernel<MODE>, dim3((unsigned)nb), dim3(Q_THREADS), lds_q, h->stream, P);             \
    } while (0)
            if (!use_filter) REFINE(0);
            else if (!dense) REFINE(1);
            else REFINE(2);
#undef REFINE
        }
        HIP_TRY(hipGetLastError());
        if (packed_dev) {
            // row-sharded search: the answers stay in HBM, packed as the message of the top-k all-gather
            hipLaunchKernelGGL(pack_topk_kernel, dim3((unsigned)((nb * k + 255) / 256)), dim3(256), 0, h->stream, P.ids_out,
                               P.dist_out, nb * k, k, (int32_t)id_offset, packed_dev + q0 * 2 * k);
            HIP_TRY(hipGetLastError());
        }
        if (direct) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            memcpy(ids_out + q0 * k, h->host_small, (size_t)nb * k * 4);
            if (dist_out) memcpy(dist_out + q0 * k, h->host_small + s_ids, (size_t)nb * k * 4);
            if (count_out) memcpy(count_out + q0, h->host_small + 2 * s_ids, (size_t)nb * 4);
        } else if (ids_out) {
            // one copy of the whole result block into page-locked memory of the handle (three copies into the caller's
            // pageable arrays cost ~25 us of idle device each), then plain memcpys
            const size_t out_bytes = 2 * s_ids + s_qf;
            if (out_bytes > h->host_out_cap) {
                if (h->host_out) (void)hipHostFree(h->host_out);
                h->host_out = nullptr;
                h->host_out_cap = 0;
                HIP_TRY(hipHostMalloc((void **)&h->host_out, out_bytes * 2, hipHostMallocDefault));
                h->host_out_cap = out_bytes * 2;
            }
            HIP_TRY(hipMemcpyAsync(h->host_out, out_block, out_bytes, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            memcpy(ids_out + q0 * k, h->host_out, (size_t)nb * k * 4);
            if (dist_out) memcpy(dist_out + q0 * k, h->host_out + s_ids, (size_t)nb * k * 4);
            if (count_out) memcpy(count_out + q0, h->host_out + 2 * s_ids, (size_t)nb * 4);
        }
        // (packed answers only: nothing to wait for -- the next batch, and whatever the caller orders on the handle's
        // stream, run behind this one)
    }
    return MORNA_OK;
}

// ---- row-sharded search: merge of the all-gathered per-shard answers on the device ----------------------------
// gathered[world][nq][2 kk] int32: kk global ids (-1 = empty slot, always last) then the kk fp32 distances' bits, each
// list ascending by (distance, id) as the refine kernel leaves it.  One thread per query walks the `world` list heads.
__global__ __launch_bounds__(256) void merge_topk_kernel(const int32_t *__restrict__ gathered, int32_t world, int64_t nq,
                                                         int32_t kk, int32_t k, int32_t *__restrict__ ids_out,
                                                         float *__restrict__ dist_out, int32_t *__restrict__ count_out)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    constexpr int MAXW = 64;
    uint8_t head[MAXW];   // kk <= 255 checked by the host
    for (int w = 0; w < world; w++) head[w] = 0;
    int n = 0;
    for (; n < k; n++) {
        int best = -1;
        uint32_t bk = 0;
        int32_t bi = 0;
        for (int w = 0; w < world; w++) {
            if (head[w] >= kk) continue;
            const int32_t *lst = gathered + ((int64_t)w * nq + q) * 2 * kk;
            const int32_t id = lst[head[w]];
            if (id < 0) continue;   // the rest of this shard's list is empty
            const uint32_t dk = f32_orderable(__int_as_float(lst[kk + head[w]]));
            if (best < 0 || dk < bk || (dk == bk && id < bi)) {
                best = w;
                bk = dk;
                bi = id;
            }
        }
        if (best < 0) break;
        const int32_t *lst = gathered + ((int64_t)best * nq + q) * 2 * kk;
        ids_out[q * k + n] = lst[head[best]];
        dist_out[q * k + n] = __int_as_float(lst[kk + head[best]]);
        head[best]++;
    }
    count_out[q] = n;
    for (; n < k; n++) {
        ids_out[q * k + n] = -1;
        dist_out[q * k + n] = INFINITY;
    }
}

int merge_topk_dev(morna_index *h, const int32_t *gathered_dev, int32_t world, int64_t nq, int32_t kk, int32_t k,
                   int32_t *ids_out, float *dist_out, int32_t *count_out)
{
    if (world <= 0 || world > 64 || nq < 0 || kk <= 0 || kk > 255 || k <= 0 || !gathered_dev || !ids_out) {
        set_error("merge_topk_dev: invalid argument (world <= 64, kk <= 255)");
        return MORNA_E_INVALID;
    }
    if (nq == 0) return MORNA_OK;
    const size_t s_ids = align_up((size_t)nq * k * 4, 256), s_cnt = align_up((size_t)nq * 4, 256);
    MORNA_TRY(h->ws.alloc(2 * s_ids + s_cnt));
    int32_t *d_ids = (int32_t *)h->ws.p;
    float *d_dist = (float *)(h->ws.p + s_ids);
    int32_t *d_cnt = (int32_t *)(h->ws.p + 2 * s_ids);
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, h->stream, gathered_dev, world, nq, kk,
                       k, d_ids, d_dist, d_cnt);
    HIP_TRY(hipGetLastError());
    // one copy of the result block into the handle's page-locked staging, then plain memcpys (as query_batch)
    const size_t out_bytes = 2 * s_ids + s_cnt;
    if (out_bytes > h->host_out_cap) {
        if (h->host_out) (void)hipHostFree(h->host_out);
        h->host_out = nullptr;
        h->host_out_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&h->host_out, out_bytes * 2, hipHostMallocDefault));
        h->host_out_cap = out_bytes * 2;
    }
    HIP_TRY(hipMemcpyAsync(h->host_out, d_ids, out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    memcpy(ids_out, h->host_out, (size_t)nq * k * 4);
    if (dist_out) memcpy(dist_out, h->host_out + s_ids, (size_t)nq * k * 4);
    if (count_out) memcpy(count_out, h->host_out + 2 * s_ids, (size_t)nq * 4);
    return MORNA_OK;
}

// =============================================================== exact search

#define E_ROWS 64       // rows per workgroup

// approx[q][row] = 2 - 2 cos in fp32 arithmetic (selection only, never returned)
template <int E_QT>   // queries sharing one pass over a block of rows
__global__ __launch_bounds__(256) void exact_scan_kernel(const float *__restrict__ X, const float *__restrict__ norm2,
                                                         int64_t n_items, int32_t dpad,
                                                         const float *__restrict__ Qf /* [nq][dpad] */,
                                                         const float *__restrict__ qn2 /* [nq] */, int64_t nq,
                                                         float *__restrict__ approx /* [nq][n_items] */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *qs = (float4 *)smem;   // [E_QT][nvec]
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int nvec = dpad / 4;
    const int64_t q0 = (int64_t)blockIdx.y * E_QT;
    const int nqt = (int)((nq - q0) < E_QT ? (nq - q0) : E_QT);
    for (int i = tid; i < E_QT * nvec; i += 256) {
        int qq = i / nvec, v = i - qq * nvec;
        qs[i] = qq < nqt ? ((const float4 *)(Qf + (q0 + qq) * dpad))[v] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * E_ROWS;
    for (int rr = w; rr < E_ROWS; rr += 4) {
        const int64_t r = r0 + rr;
        if (r >= n_items) break;
        const float4 *x = (const float4 *)(X + r * dpad);
        float acc[E_QT];
#pragma unroll
        for (int qq = 0; qq < E_QT; qq++) acc[qq] = 0.f;
        for (int i = lane; i < nvec; i += WAVE) {
            const float4 xv = x[i];
#pragma unroll
            for (int qq = 0; qq < E_QT; qq++) {
                const float4 qv = qs[qq * nvec + i];
                acc[qq] += xv.x * qv.x + xv.y * qv.y + xv.z * qv.z + xv.w * qv.w;
            }
        }
        const float rn = norm2[r];
#pragma unroll
        for (int qq = 0; qq < E_QT; qq++) {
            const float pq = wave_sum_xor(acc[qq]);
            if (lane == 0 && qq < nqt) {
                const double ppqq = (double)rn * (double)qn2[q0 + qq];
                approx[(q0 + qq) * n_items + r] = ppqq > 0.0 ? (float)(2.0 - 2.0 * (double)pq / sqrt(ppqq)) : 2.0f;
            }
        }
    }
}

// per query: threshold = k-th smallest approx value; candidates = everything within eps of it
__global__ __launch_bounds__(256) void exact_select_kernel(const float *__restrict__ approx, int64_t n_items,
                                                           int32_t k, float eps, int32_t cap,
                                                           int32_t *__restrict__ cand /* [nq][cap] */,
                                                           int32_t *__restrict__ ncand_out /* [nq] */)
{
    __shared__ uint64_t s_red[Q_WAVES];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const float *a = approx + qi * n_items;
    const int kk = (int)(k < n_items ? k : n_items);
    uint64_t prev = 0;
    bool have_prev = false;
    constexpr int HELD = 32;   // a shard of up to 8192 rows: the scan values of the query stay in registers for all k rounds
    if (n_items <= 256 * HELD) {
        uint32_t key32[HELD];
#pragma unroll
        for (int u = 0; u < HELD; u++) {
            const int64_t i = tid + 256 * u;
            key32[u] = i < n_items ? f32_orderable(a[i]) : 0xffffffffu;
        }
        for (int r = 0; r < kk; r++) {
            uint64_t best = ~0ull;
#pragma unroll
            for (int u = 0; u < HELD; u++) {
                const uint64_t key = ((uint64_t)key32[u] << 32) | (uint32_t)(tid + 256 * u);
                if (tid + 256 * u < n_items && (!have_prev || key > prev) && key < best) best = key;
            }
            best = block_min_u64(best, s_red, tid);
            prev = best;
            have_prev = true;
        }
    } else {
        for (int r = 0; r < kk; r++) {
            uint64_t best = ~0ull;
            for (int64_t i = tid; i < n_items; i += 256) {
                const uint64_t key = ((uint64_t)f32_orderable(a[i]) << 32) | (uint32_t)i;
                if ((!have_prev || key > prev) && key < best) best = key;
            }
            best = block_min_u64(best, s_red, tid);
            prev = best;
            have_prev = true;
        }
    }
    const float thr = f32_from_orderable((uint32_t)(prev >> 32)) + eps;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int64_t i = tid; i < n_items; i += 256) {
        if (a[i] <= thr) {
            int slot = atomicAdd(&s_n, 1);
            if (slot < cap) cand[qi * cap + slot] = (int32_t)i;
        }
    }
    __syncthreads();
    if (tid == 0) ncand_out[qi] = s_n;   // may exceed cap: the host then retries with more room
}

// cosine_distance in the reference's order (morna.py:101-114): one thread per candidate walks ITS row front to back in
// fp64 (pp += i*i; qq += j*j; pq += i*j), a 128-byte line of the row requested ahead of the one being consumed; the query
// is read through wave-uniform loads.  The pass is bound by the issue of its three dependent fp64 chains (8192 steps each
// at D = 8192), i.e. by the NUMBER of candidates: what keeps it short is the width of the scan's window (exact_scan_eps).
// Tried at configs[4]'s shard (6250 queries, ~100 candidates each when the window was 1e-3): rows staged through LDS by the
// whole workgroup 5.9 ms, a parallel-fp64 narrowing pass in front 4.4 ms, this form 3.4 ms.
#define RR_COLS 32
#define RR_THREADS 64   // one wave per query: the candidates are a few dozen, and a wave that waits for its chains leaves the SIMD to other queries
__global__ __launch_bounds__(RR_THREADS) void exact_rerank_kernel(const float *__restrict__ X, int32_t dim, int32_t dpad,
                                                           const double *__restrict__ Qd /* [nq][dim] */,
                                                           const int32_t *__restrict__ cand, const int32_t *__restrict__ ncand,
                                                           int32_t cap, int32_t k, double *__restrict__ cdist /* [nq][cap] */,
                                                           int32_t *__restrict__ ids_out, double *__restrict__ dist_out,
                                                           int32_t *__restrict__ count_out)
{
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const double *q = Qd + qi * dim;
    const int n = ncand[qi] < cap ? ncand[qi] : cap;
    const int32_t *c = cand + qi * cap;
    double *cd = cdist + qi * cap;
    for (int t = tid; t < n; t += RR_THREADS) {
        const float4 *row = (const float4 *)(X + (int64_t)c[t] * dpad);
        double pp = 0.0, qq = 0.0, pq = 0.0;
        float4 cur[RR_COLS / 4], nxt[RR_COLS / 4];
#pragma unroll
        for (int u = 0; u < RR_COLS / 4; u++) cur[u] = row[u];   // (dpad is a multiple of 256: every line read exists)
        for (int z0 = 0; z0 < dim; z0 += RR_COLS) {
            const bool more = z0 + RR_COLS < dim;
#pragma unroll
            for (int u = 0; u < RR_COLS / 4; u++) nxt[u] = row[(more ? z0 + RR_COLS : z0) / 4 + u];
            const int zn = dim - z0 < RR_COLS ? dim - z0 : RR_COLS;
#pragma unroll
            for (int u = 0; u < RR_COLS / 4; u++) {
                const float e4[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    if (4 * u + v < zn) {
                        const double i = (double)e4[v], j = q[z0 + 4 * u + v];
                        pp = __dadd_rn(pp, __dmul_rn(i, i));
                        qq = __dadd_rn(qq, __dmul_rn(j, j));
                        pq = __dadd_rn(pq, __dmul_rn(i, j));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RR_COLS / 4; u++) cur[u] = nxt[u];
        }
        const double ppqq = __dmul_rn(pp, qq);
        double distance = 2.0;
        if (ppqq > 0.0) distance = __dsub_rn(2.0, __ddiv_rn(__dmul_rn(2.0, pq), __dsqrt_rn(ppqq)));
        cd[t] = __dsqrt_rn(distance);   // NaN where Python's math.sqrt raises
    }
    // A NaN is a row for which Python's math.sqrt raises ValueError (negative radicand: the cosine rounded to more
    // than 1, i.e. a row all but parallel to the query).  The reference evaluates EVERY row, so one such row anywhere
    // fails the whole query; such a row has a scan distance of ~0 and is therefore always among the candidates.  The
    // query's count is reported as -1 and the caller raises (MornaSearch.exact_search_nn).
    __shared__ int s_nan;
    if (tid == 0) s_nan = 0;
    __syncthreads();
    {
        bool any_nan = false;
        for (int t = tid; t < n; t += RR_THREADS) any_nan |= cd[t] != cd[t];
        if (any_nan) s_nan = 1;
    }
    __syncthreads();
    // rank = number of candidates that bisect_left insertion leaves in front:
    // smaller distance, or equal distance and HIGHER id (inserted later, lands first).
    // NaN distances go last, by ascending id (the lists are still filled in, for diagnosis).
    for (int t = tid; t < n; t += RR_THREADS) {
        const double d = cd[t];
        const int32_t id = c[t];
        const bool dnan = d != d;
        int rank = 0;
        for (int u = 0; u < n; u++) {
            const double du = cd[u];
            const bool unan = du != du;
            bool before;
            if (dnan) before = !unan || c[u] < id;
            else before = (du < d) || (du == d && c[u] > id);
            rank += (before && u != t) ? 1 : 0;
        }
        if (rank < k) {
            ids_out[qi * k + rank] = id;
            dist_out[qi * k + rank] = d;
        }
    }
    const int kout = k < n ? k : n;
    for (int r = kout + tid; r < k; r += RR_THREADS) {
        ids_out[qi * k + r] = -1;
        dist_out[qi * k + r] = INFINITY;
    }
    if (tid == 0) count_out[qi] = s_nan ? -1 : kout;
}

// Selection for shards of more than 8192 rows: TWO passes over a query's scan values instead of k (the k rounds of
// exact_select_kernel read 4 * N * k bytes per query: 200 GB for configs[4]'s 50 000 queries x 50 000 rows, k = 20).
//   A  every thread takes the minimum of its strided share; the kk-th smallest of the 256 minima bounds the kk-th smallest
//      value from above (kk distinct elements lie at or below it);
//   B  everything within eps of that bound is collected as (key, id) pairs -- a superset of the candidates;
//   C  the kk-th smallest of the collected pairs, by counting, IS the kk-th smallest of all values: thr = it + eps;
//   D  the candidates are the collected pairs at or below thr -- the set exact_select_kernel produces.
// `pairs` is the memory of the re-rank's distance array (not yet in use).  More collected than `cap`: the count is reported
// and the host retries with room for all of them, as for the candidates themselves.  Needs k <= 256 (one minimum per thread).
__global__ __launch_bounds__(256) void exact_select2_kernel(const float *__restrict__ approx, int64_t n_items, int32_t k, float eps,
                                                            int32_t cap, uint2 *__restrict__ pairs /* [nq][cap] */,
                                                            int32_t *__restrict__ cand /* [nq][cap] */,
                                                            int32_t *__restrict__ ncand_out /* [nq] */)
{
    __shared__ uint64_t s_key[256];
    __shared__ int s_n;
    __shared__ uint32_t s_thr;
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const float *a = approx + qi * n_items;
    const int kk = (int)(k < n_items ? k : n_items);
    uint64_t mine = ~0ull;
    for (int64_t i = tid; i < n_items; i += 256) {
        const uint64_t key = ((uint64_t)f32_orderable(a[i]) << 32) | (uint32_t)i;
        mine = key < mine ? key : mine;
    }
    s_key[tid] = mine;
    if (tid == 0) s_n = 0;
    __syncthreads();
    {
        int below = 0;
        for (int j = 0; j < 256; j++) below += s_key[j] < mine ? 1 : 0;
        if (mine != ~0ull && below == kk - 1) s_thr = (uint32_t)(mine >> 32);
    }
    __syncthreads();
    const float bound = f32_from_orderable(s_thr) + eps;
    uint2 *pr = pairs + qi * cap;
    for (int64_t i = tid; i < n_items; i += 256) {
        const float v = a[i];
        if (v <= bound) {
            const int slot = atomicAdd(&s_n, 1);
            if (slot < cap) pr[slot] = make_uint2(f32_orderable(v), (uint32_t)i);
        }
    }
    __syncthreads();
    const int m = s_n;
    if (m > cap) {
        if (tid == 0) ncand_out[qi] = m;
        return;
    }
    for (int t = tid; t < m; t += 256) {
        const uint2 p = pr[t];
        const uint64_t kt = ((uint64_t)p.x << 32) | p.y;
        int below = 0;
        for (int u = 0; u < m; u++) {
            const uint2 o = pr[u];
            below += (((uint64_t)o.x << 32) | o.y) < kt ? 1 : 0;
        }
        if (below == kk - 1) s_thr = p.x;
    }
    __syncthreads();
    const float thr = f32_from_orderable(s_thr) + eps;
    __syncthreads();
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int t = tid; t < m; t += 256) {
        const uint2 p = pr[t];
        if (f32_from_orderable(p.x) <= thr) cand[qi * cap + atomicAdd(&s_n, 1)] = (int32_t)p.y;
    }
    __syncthreads();
    if (tid == 0) ncand_out[qi] = s_n;
}

// The query of exact_search_nn when it is a STORED row (or an fp32 row handed over in device memory): the fp32 values
// widened to fp64 -- what `zip(self.annoy_index.get_item_vector(i), query)` sees when the query came out of the index
// (morna.py:697-703).  One workgroup per query; src_stride floats between rows, items == null: row q.
__global__ void exact_widen_kernel(const float *__restrict__ src, int64_t src_stride, const int32_t *__restrict__ items,
                                   int32_t dim, double *__restrict__ Qd)
{
    const int64_t q = blockIdx.x;
    const float *row = src + (items ? (int64_t)items[q] : q) * src_stride;
    for (int z = threadIdx.x; z < dim; z += blockDim.x) Qd[q * dim + z] = (double)row[z];
}

// a batch's exact answers -> their place in the message of the row-sharded exact search (global ids)
__global__ void exact_pack_kernel(const int32_t *__restrict__ ids, const double *__restrict__ dist, const int32_t *__restrict__ cnt,
                                  int64_t nb, int32_t k, int32_t id_offset, int32_t *__restrict__ m_ids, int32_t *__restrict__ m_cnt,
                                  double *__restrict__ m_dist)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb * k) {
        const int32_t id = ids[i];
        m_ids[i] = id >= 0 ? id + id_offset : -1;
        m_dist[i] = dist[i];
    }
    if (i < nb) m_cnt[i] = cnt[i];
}

// fp64 query -> fp32 image + its squared norm (selection pass only)
__global__ void exact_prep_kernel(const double *__restrict__ Qd, int64_t nq, int32_t dim, int32_t dpad,
                                  float *__restrict__ Qf, float *__restrict__ qn2)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    if (q >= nq) return;
    float s = 0.f;
    for (int i = lane; i < dpad; i += WAVE) {
        float v = i < dim ? (float)Qd[q * dim + i] : 0.f;
        Qf[q * dpad + i] = v;
        s += v * v;
    }
    s = wave_sum_xor(s);
    if (lane == 0) qn2[q] = s;
}

// ---- batched scan on the matrix cores ---------------------------------------------
// Many queries against all rows is a dense contraction C[q][r] = sum_d Q[q][d] X[r][d]
// (2*Q*N*D flop over 4*D*(Q+N) bytes per tile pass): the one GEMM-shaped piece of the
// path, so it runs on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate).
// 128 queries x 128 rows per workgroup, K stepped by 32 through LDS, each of the 4 waves
// owns a 64 x 64 quadrant as 2 x 2 MFMA tiles.  The result only SELECTS candidates; the
// returned distances are recomputed in the reference's fp64 order by exact_rerank_kernel.
#define MM_TILE 128
#define MM_BK 32
#define MM_LD (MM_BK + 4)   // padded LDS row (floats); 144-byte rows keep float4 accesses aligned
#define MM_FLUSH 128        // columns per accumulation chain of the scan (a multiple of MM_BK, a power of two)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void exact_scan_mfma_kernel(const float *__restrict__ X, const float *__restrict__ norm2,
                                                              int64_t n_items, int32_t dpad,
                                                              const float *__restrict__ Qf, const float *__restrict__ qn2,
                                                              int64_t nq, float *__restrict__ approx)
{
    __shared__ __attribute__((aligned(16))) float As[MM_TILE * MM_LD];   // queries  [128][36]
    __shared__ __attribute__((aligned(16))) float Bs[MM_TILE * MM_LD];   // rows     [128][36]
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int wm = w >> 1, wn = w & 1;                 // the wave's 64 x 64 quadrant
    const int64_t r0 = (int64_t)blockIdx.x * MM_TILE, q0 = (int64_t)blockIdx.y * MM_TILE;

    // global -> register staging: 4 float4 per operand per thread (row = idx / 8, 16-byte column = idx % 8)
    float4 ga[4], gb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int idx = tid + it * 256, row = idx >> 3, c4 = idx & 7;
            const int64_t q = q0 + row, r = r0 + row;
            ga[it] = q < nq ? *(const float4 *)(Qf + q * dpad + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            gb[it] = r < n_items ? *(const float4 *)(X + r * dpad + k0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int idx = tid + it * 256, row = idx >> 3, c4 = idx & 7;
            *(float4 *)(As + row * MM_LD + c4 * 4) = ga[it];
            *(float4 *)(Bs + row * MM_LD + c4 * 4) = gb[it];
        }
    };
    // Blocked accumulation: the running accumulator of a tile is added into a second one every MM_FLUSH columns and
    // cleared, so that a chain of fmaf is MM_FLUSH products long and the second level adds dpad / MM_FLUSH partial sums --
    // an error bound of (MM_FLUSH + dpad / MM_FLUSH) u instead of dpad / 2 u (one chain per half of the columns, as this
    // kernel first had it): the selection window of exact_scan_eps() is 12x narrower at D = 8192, and the fp64 re-rank,
    // whose time is the number of candidates, 3-4x shorter.
    f32x16 acc[2][2], acc_b[2][2];   // running chain; sum of the flushed chains
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = acc_b[i][j][e] = 0.f;

    fetch(0);
    stash();
    __syncthreads();
    const int lr = lane & 31, lh = lane >> 5;
    for (int k0 = 0; k0 < dpad; k0 += MM_BK) {
        const bool more = k0 + MM_BK < dpad;
        if (more) fetch(k0 + MM_BK);                   // next K slab in flight under the MFMAs
#pragma unroll
        for (int blk = 0; blk < MM_BK / 8; blk++) {
            // lane half h supplies k = 8*blk + 4*h + j to MFMA j: any k order is a valid sum here
            float4 a4[2], b4[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                a4[t] = *(const float4 *)(As + (wm * 64 + t * 32 + lr) * MM_LD + blk * 8 + lh * 4);
                b4[t] = *(const float4 *)(Bs + (wn * 64 + t * 32 + lr) * MM_LD + blk * 8 + lh * 4);
            }
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++) {
                    f32x16 &c = acc[tm][tn];
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].x, b4[tn].x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].y, b4[tn].y, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].z, b4[tn].z, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].w, b4[tn].w, c, 0, 0, 0);
                }
        }
        if (((k0 + MM_BK) & (MM_FLUSH - 1)) == 0 || !more) {   // uniform: the chain ends here
#pragma unroll
            for (int tm = 0; tm < 2; tm++)
#pragma unroll
                for (int tn = 0; tn < 2; tn++) {
                    acc_b[tm][tn] = acc_b[tm][tn] + acc[tm][tn];
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[tm][tn][e] = 0.f;
                }
        }
        __syncthreads();
        if (more) stash();
        __syncthreads();
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) acc[tm][tn] = acc_b[tm][tn];
    // epilogue: C/D layout of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
    // A (queries) indexes the rows of the tile, B (matrix rows) its columns: 32 lanes write 32 consecutive r.
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
            const int64_t r = r0 + wn * 64 + tn * 32 + lr;
            const float rn = r < n_items ? norm2[r] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int64_t q = q0 + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (q < nq && r < n_items) {
                    const double ppqq = (double)rn * (double)qn2[q];
                    approx[q * n_items + r] = ppqq > 0.0 ? (float)(2.0 - 2.0 * (double)acc[tm][tn][e] / sqrt(ppqq)) : 2.0f;
                }
            }
        }
}

// How far the scan's 2 - 2 cos can be from the exact value, times two (the k-th smallest scan value that sets the
// threshold and the candidate itself can both be off): every row within this of the threshold is kept, so no row
// of the true top k is lost.  Unit roundoff u = 2^-24, errors relative to |x| |q| >= sum |x_i q_i|:
//   dot, vector-ALU scan : a lane adds dpad / 64 products in turn, then the 6-stage butterfly
//   dot, matrix-core scan: chains of MM_FLUSH products (v_mfma_f32_32x32x2_f32 is a chain of IEEE fmaf: measured bit for
//                          bit on MI355X by scripts/mfma_accum_probe.hip, profiles/r02_mfma_accum_probe.txt), their
//                          dpad / MM_FLUSH partial sums added one after the other
//   norms (norm2 of the row, qn2 of the query): sums of dpad / 64 squares per lane + butterfly; half of each enters cos
//   the fp32 image of the query and the rounding of the value itself: 2 u
static float exact_scan_eps(int32_t dpad, bool mfma)
{
    const double u = 5.9604644775390625e-8;
    const double e_dot = mfma ? (MM_FLUSH + dpad / MM_FLUSH + 8) * u : (dpad / 64 + 8) * u;
    const double e_norm = (dpad / 64 + 8) * u;
    return (float)(2.0 * 2.0 * (e_dot + e_norm + 2 * u));
}

template <int QT>
static void launch_exact_scan(morna_index *h, int64_t N, int64_t nb, const float *Qf, const float *qn2, float *approx)
{
    dim3 grid((unsigned)((N + E_ROWS - 1) / E_ROWS), (unsigned)((nb + QT - 1) / QT));
    hipLaunchKernelGGL(exact_scan_kernel<QT>, grid, dim3(256), (size_t)QT * h->dpad * 4, h->stream, h->X.p,
                       h->norm2.p, N, h->dpad, Qf, qn2, nb, approx);
}

// Message of the row-sharded exact search, nq queries: ids int32 [nq][k] (global, -1 = empty) | count int32 [nq] (-1: the
// reference raises for this query) | pad to 8 bytes | distances fp64 [nq][k]
size_t exact_msg_dist_offset(int64_t nq, int32_t k) { return align_up((size_t)nq * ((size_t)k + 1) * 4, 8); }
size_t exact_msg_bytes(int64_t nq, int32_t k) { return exact_msg_dist_offset(nq, k) + (size_t)nq * k * 8; }

// The queries: q_host fp64 [nq][dim] (host), or q_dev fp32 [nq][dim] (this device's memory), or stored rows items_host.
// The answers: host arrays, and / or msg_dev (device memory, exact_msg_bytes(nq, k)) with global ids = local + id_offset --
// then only enqueued on the handle's stream behind the last selection.
int exact_search_any(morna_index *h, const double *q_host, const float *q_dev, const int32_t *items_host, int64_t nq, int32_t k,
                     int32_t *ids_out, double *dist_out, int32_t *count_out, uint8_t *msg_dev, int64_t id_offset)
{
    MORNA_TRY(upload_host_rows(h));
    if (h->n_items <= 0) {
        set_error("exact search on an empty index");
        return MORNA_E_EMPTY;
    }
    if (k <= 0 || nq < 0) {
        set_error("exact search: k must be positive");
        return MORNA_E_INVALID;
    }
    if (nq == 0) return MORNA_OK;
    const int64_t N = h->n_items;
    const int32_t D = h->dim, dpad = h->dpad;
    if (items_host)
        for (int64_t i = 0; i < nq; i++)
            if (items_host[i] < 0 || items_host[i] >= N) {
                set_error("Item index %d out of range [0, %lld)", items_host[i], (long long)N);
                return MORNA_E_RANGE;
            }
    // queries resident per pass: their fp32 images share 64 KiB of LDS
    const int qt = (size_t)dpad * 4 * 8 <= 65536 ? 8 : (size_t)dpad * 4 * 4 <= 65536 ? 4
                   : (size_t)dpad * 4 * 2 <= 65536 ? 2 : 1;
    if ((size_t)dpad * 4 > 160 * 1024 || (size_t)D * 8 > 160 * 1024) {
        set_error("exact search: dimension %d does not fit LDS", D);
        return MORNA_E_INVALID;
    }
    // queries per batch: approx[nq][N] floats capped at 2 GiB.  The workspace stays with the handle (a hipMalloc / hipFree
    // pair per array and call cost more than a small search)
    const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(nq, ((int64_t)1 << 29) / std::max<int64_t>(N, 1)));
    const size_t s_qd = align_up((size_t)batch * D * 8, 256), s_qf = align_up((size_t)batch * dpad * 4, 256),
                 s_b4 = align_up((size_t)batch * 4, 256), s_ap = align_up((size_t)batch * N * 4, 256),
                 s_ids = align_up((size_t)batch * k * 4, 256), s_dist = align_up((size_t)batch * k * 8, 256);
    MORNA_TRY(h->ex_ws.alloc(s_qd + s_qf + 4 * s_b4 + s_ap + s_ids + s_dist));
    uint8_t *p = h->ex_ws.p;
    double *Qd = (double *)p; p += s_qd;
    float *Qf = (float *)p; p += s_qf;
    float *qn2 = (float *)p; p += s_b4;
    int32_t *ncand = (int32_t *)p; p += s_b4;
    int32_t *d_items = (int32_t *)p; p += s_b4;
    float *approx = (float *)p; p += s_ap;
    uint8_t *const out_block = p;   // ids, distances, counts: one block, one copy to the host
    int32_t *d_ids = (int32_t *)p; p += s_ids;
    double *d_dist = (double *)p; p += s_dist;
    int32_t *d_cnt = (int32_t *)p; p += s_b4;
    std::vector<int32_t> h_ncand((size_t)batch);
    int32_t cap = std::max<int32_t>(std::max(64, 4 * k), h->ex_cap), need_max = 0;
    for (int64_t q0 = 0; q0 < nq; q0 += batch) {
        const int64_t nb = std::min(batch, nq - q0);
        if (q_host) {
            HIP_TRY(hipMemcpyAsync(Qd, q_host + q0 * D, (size_t)nb * D * 8, hipMemcpyHostToDevice, h->stream));
        } else if (q_dev) {
            hipLaunchKernelGGL(exact_widen_kernel, dim3((unsigned)nb), dim3(256), 0, h->stream, q_dev + q0 * D, (int64_t)D, nullptr, D, Qd);
        } else {
            HIP_TRY(hipMemcpyAsync(d_items, items_host + q0, (size_t)nb * 4, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(exact_widen_kernel, dim3((unsigned)nb), dim3(256), 0, h->stream, h->X.p, (int64_t)dpad, d_items, D, Qd);
        }
        {
            // one pass over the matrix per qt queries: 4*D*N bytes each
            const int64_t q_per_pass = nb >= 32 ? MM_TILE : qt;
            ScopedTimer tm(h, MORNA_T_EXACT, 4 * (int64_t)D * N * ((nb + q_per_pass - 1) / q_per_pass));
            hipLaunchKernelGGL(exact_prep_kernel, dim3((unsigned)((nb * WAVE + 255) / 256)), dim3(256), 0, h->stream,
                               Qd, nb, D, dpad, Qf, qn2);
            ScopedTimer ts(h, MORNA_T_EXACT_SCAN, nb >= 32 ? 2 * (int64_t)nb * N * dpad : 0);   // "bytes" = flops on the matrix cores
            if (nb >= 32) {   // enough queries to fill MFMA tiles: dense contraction on the matrix cores
                dim3 grid((unsigned)((N + MM_TILE - 1) / MM_TILE), (unsigned)((nb + MM_TILE - 1) / MM_TILE));
                hipLaunchKernelGGL(exact_scan_mfma_kernel, grid, dim3(256), 0, h->stream, h->X.p, h->norm2.p, N, dpad, Qf,
                                   qn2, nb, approx);
            } else if (qt == 8) launch_exact_scan<8>(h, N, nb, Qf, qn2, approx);
            else if (qt == 4) launch_exact_scan<4>(h, N, nb, Qf, qn2, approx);
            else if (qt == 2) launch_exact_scan<2>(h, N, nb, Qf, qn2, approx);
            else launch_exact_scan<1>(h, N, nb, Qf, qn2, approx);
        }
        HIP_TRY(hipGetLastError());
        const float eps = exact_scan_eps(dpad, nb >= 32);   // which scan ran
        ScopedTimer tm_sel(h, MORNA_T_EXACT, 0);            // selection + fp64 re-rank: the same group as the scan
        const bool select2_on = !(getenv("MORNA_EXACT_SELECT2") && atoi(getenv("MORNA_EXACT_SELECT2")) == 0);   // (read per call: a test switches it)
        for (;;) {
            MORNA_TRY(h->ex_cand.alloc((size_t)batch * cap));
            MORNA_TRY(h->ex_cdist.alloc((size_t)batch * cap));
            if (N > 8192 && k <= 256 && select2_on)
                hipLaunchKernelGGL(exact_select2_kernel, dim3((unsigned)nb), dim3(256), 0, h->stream, approx, N, k, eps, cap,
                                   (uint2 *)h->ex_cdist.p, h->ex_cand.p, ncand);
            else
                hipLaunchKernelGGL(exact_select_kernel, dim3((unsigned)nb), dim3(256), 0, h->stream, approx, N, k, eps,
                                   cap, h->ex_cand.p, ncand);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h_ncand.data(), ncand, (size_t)nb * 4, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            int32_t need = 0;
            for (int64_t i = 0; i < nb; i++) need = std::max(need, h_ncand[(size_t)i]);
            need_max = std::max(need_max, need);
            if (need <= cap) break;
            cap = need;   // huge tie groups at the boundary: make room for all of them
        }
        hipLaunchKernelGGL(exact_rerank_kernel, dim3((unsigned)nb), dim3(RR_THREADS), 0, h->stream, h->X.p, D, dpad,
                           Qd, h->ex_cand.p, ncand, cap, k, h->ex_cdist.p, d_ids, d_dist, d_cnt);
        HIP_TRY(hipGetLastError());
        if (msg_dev) {
            hipLaunchKernelGGL(exact_pack_kernel, dim3((unsigned)((nb * k + 255) / 256)), dim3(256), 0, h->stream, d_ids, d_dist, d_cnt,
                               nb, k, (int32_t)id_offset, (int32_t *)msg_dev + q0 * k, (int32_t *)msg_dev + nq * k + q0,
                               (double *)(msg_dev + exact_msg_dist_offset(nq, k)) + q0 * k);
            HIP_TRY(hipGetLastError());
            h->unsettled = true;
        }
        if (ids_out) {
            const size_t out_bytes = s_ids + s_dist + s_b4;
            if (out_bytes > h->host_out_cap) {
                if (h->host_out) (void)hipHostFree(h->host_out);
                h->host_out = nullptr;
                h->host_out_cap = 0;
                HIP_TRY(hipHostMalloc((void **)&h->host_out, out_bytes * 2, hipHostMallocDefault));
                h->host_out_cap = out_bytes * 2;
            }
            HIP_TRY(hipMemcpyAsync(h->host_out, out_block, out_bytes, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            memcpy(ids_out + q0 * k, h->host_out, (size_t)nb * k * 4);
            if (dist_out) memcpy(dist_out + q0 * k, h->host_out + s_ids, (size_t)nb * k * 8);
            if (count_out) memcpy(count_out + q0, h->host_out + s_ids + s_dist, (size_t)nb * 4);
        }
    }
    h->ex_cap = need_max;   // the next call starts with the room this one needed (no more: one query with a tie of thousands
                            // does not size every later call)
    return MORNA_OK;
}

int exact_search(morna_index *h, const double *q, int64_t nq, int32_t k, int32_t *ids_out, double *dist_out,
                 int32_t *count_out)
{
    return exact_search_any(h, q, nullptr, nullptr, nq, k, ids_out, dist_out, count_out, nullptr, 0);
}

}  // namespace morna
