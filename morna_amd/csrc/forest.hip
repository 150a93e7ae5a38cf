// forest.hip -- level-synchronous random-projection forest build on gfx950.
//
// Replaces AnnoyIndex.build(n_trees) (reference call site morna.py:425; the
// algorithm is spotify/annoy's _make_tree / two_means / Angular::create_split,
// restated in oracle/annoy_oracle.c mode 1 and SURVEY.md section 2.1).
//
// All n_trees trees advance one level per round.  A tree is a permutation of the
// item ids (perm[tree][N]); a node is a contiguous segment of it.  During a build the
// permutations live in a work buffer of two images per tree, [tree][2][N]: a node of depth
// L has its items in image L & 1 (TASK_ITEMS_AT), a partition writes its children into the
// other image, and the finished leaves are gathered into perm at the end.  Per level:
//   two_means_wave_kernel  one WAVE per split node, centroids / current row / next
//                     row in registers: 200 sequential centroid updates, the row of
//                     step l+1 in flight during step l (the Kiss32 stream does not
//                     depend on data; it arrives by LDS-DMA).  two_means_strip_kernel:
//                     four waves per node, each owning 16 of the 64 canonical lanes of every
//                     dot product and that strip of the centroids, for levels whose nodes
//                     fit the chip at once.  two_means_kernel: the LDS form (one workgroup
//                     per node) for rows too long for the register file.
//   (splitmm.hip)     the sides of the whole level as one fp16 MFMA product that filters,
//                     exact fp32 dots for the pairs it cannot decide -- the form that runs
//                     while a tree has at most 32 split nodes, and, once the rows of the
//                     contraction are ordered and a row tile is multiplied with its own list
//                     of tasks, up to 256 per tree and for retries.  Otherwise:
//   split_kernel      every row of every split node is dotted (wavefront dot product,
//                     hyperplane resident in LDS) against its node's hyperplane, one
//                     workgroup per 64 positions of a node; launch order sorted by row
//                     id, one run per XCD, so the trees' re-reads hit L2.
//   split_rw_kernel   the same work at shallow levels (<= 4 split nodes per tree) as
//                     row windows x tree groups: a row is loaded once into registers
//                     and used for every tree of the group (runs with MORNA_SPLIT_MM=0).
//   post_counts_kernel  the level's right-side counts into a page-locked mailbox the host polls
//   (host)            annoy's 3-attempt / 0.95 imbalance rule on the counts
//   fallback_kernel   random sides for nodes still above 0.99
//   partition_kernel  stable partition of each segment by side, from one image of the tree into the other; keeps
//                     inv[tree][row], the position in the tree of the item a row of the contraction stands for,
//                     current (what the matrix-core split finds a row's node with)
//   pull_tasks_kernel / gather_leaves_kernel  task lists and node tables out of page-locked host memory; the leaves
//                     into perm when the build ends
// Node ids are handed out breadth-first, children of the i-th split node of a
// level get consecutive ids, exactly as oracle mode 1 does.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "devutil.hpp"

namespace morna {

#define TM_THREADS 256
#define TM_ITERS 200
#ifndef TM_STRIP_DEPTH
#define TM_STRIP_DEPTH 4   // rows in flight per node in two_means_strip_kernel
#endif

// ------------------------------------------------------------------ two_means

// normalise an LDS vector in place: v /= sqrt(dot(v, v)) when the norm is > 0
__device__ inline void lds_normalize(float *v, int dpad, int tid, int lane, int w, float *s_tmp)
{
    if (w == 0) {
        float n2 = wave_dot((const float4 *)v, (const float4 *)v, dpad / 4, lane);
        if (lane == 0) s_tmp[0] = sqrtf(n2);
    }
    __syncthreads();
    const float norm = s_tmp[0];
    if (norm > 0.f)
        for (int z = tid; z < dpad; z += TM_THREADS) v[z] = v[z] / norm;
    __syncthreads();
}

__global__ __launch_bounds__(TM_THREADS) void two_means_kernel(const float *__restrict__ X,
                                                               const float *__restrict__ norm2, int64_t n_items,
                                                               int32_t dpad, const int32_t *__restrict__ perm,
                                                               const SplitTask *__restrict__ tasks, uint32_t seed,
                                                               float *__restrict__ hp, int32_t *__restrict__ ones)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (threadIdx.x == 0) ones[blockIdx.x] = 0;   // the split kernels that follow count this task's right side here
    float *p = (float *)smem;        // centroid p   [dpad]
    float *q = p + dpad;             // centroid q   [dpad]
    float *xs = q + dpad;            // current row  [dpad]
    __shared__ float s_tmp[4];       // 0: scratch norm, 1: pp, 2: qq
    __shared__ float s_pq[2];

    const SplitTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int nvec = dpad / 4;
    const int32_t *items = perm + TASK_ITEMS_AT(t, n_items);
    Kiss32 rng(node_seed(seed, (uint32_t)t.tree, (uint32_t)t.level, (uint32_t)t.start, (uint32_t)t.attempt));

    // two distinct random items seed the centroids
    uint32_t i = rng.index((uint32_t)t.count);
    uint32_t j = rng.index((uint32_t)t.count - 1u);
    j += (j >= i);
    {
        const float4 *xi = (const float4 *)(X + (int64_t)items[i] * dpad);
        const float4 *xj = (const float4 *)(X + (int64_t)items[j] * dpad);
        for (int v = tid; v < nvec; v += TM_THREADS) {
            ((float4 *)p)[v] = xi[v];
            ((float4 *)q)[v] = xj[v];
        }
    }
    __syncthreads();
    lds_normalize(p, dpad, tid, lane, w, s_tmp);
    lds_normalize(q, dpad, tid, lane, w, s_tmp);
    if (w == 0) {
        float v = wave_dot((const float4 *)p, (const float4 *)p, nvec, lane);
        if (lane == 0) s_tmp[1] = v;
    } else if (w == 1) {
        float v = wave_dot((const float4 *)q, (const float4 *)q, nvec, lane);
        if (lane == 0) s_tmp[2] = v;
    }
    // first row of the loop
    uint32_t k = rng.index((uint32_t)t.count);
    int32_t it = items[k];
    {
        const float4 *xk = (const float4 *)(X + (int64_t)it * dpad);
        for (int v = tid; v < nvec; v += TM_THREADS) ((float4 *)xs)[v] = xk[v];
    }
    __syncthreads();

    int ic = 1, jc = 1;
    // registers for the prefetched next row: nvec / 256 float4 per thread (<= 8 for D <= 8192)
    constexpr int PF = 8;
    for (int l = 0; l < TM_ITERS; l++) {
        // prefetch row l+1 while row l is being used
        uint32_t k_next = 0;
        int32_t it_next = 0;
        float4 pf[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool more = l + 1 < TM_ITERS;
        if (more) {
            k_next = rng.index((uint32_t)t.count);
            it_next = items[k_next];
            const float4 *xn = (const float4 *)(X + (int64_t)it_next * dpad);
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int v = tid + u * TM_THREADS;
                if (v < nvec) pf[u] = xn[v];
            }
        }
        const float nk2 = norm2[it];
        if (w == 0) {
            float v = wave_dot((const float4 *)p, (const float4 *)xs, nvec, lane);
            if (lane == 0) s_pq[0] = v;
        } else if (w == 1) {
            float v = wave_dot((const float4 *)q, (const float4 *)xs, nvec, lane);
            if (lane == 0) s_pq[1] = v;
        }
        __syncthreads();
        const float di = (float)ic * ang_dist(s_tmp[1], nk2, s_pq[0]);
        const float dj = (float)jc * ang_dist(s_tmp[2], nk2, s_pq[1]);
        const float norm = sqrtf(nk2);
        if (norm > 0.f) {
            if (di < dj) {
                const float fic = (float)ic, fic1 = (float)(ic + 1);
                for (int z = tid; z < dpad; z += TM_THREADS) p[z] = (p[z] * fic + xs[z] / norm) / fic1;
                __syncthreads();
                if (w == 0) {
                    float v = wave_dot((const float4 *)p, (const float4 *)p, nvec, lane);
                    if (lane == 0) s_tmp[1] = v;
                }
                ic++;
            } else if (dj < di) {
                const float fjc = (float)jc, fjc1 = (float)(jc + 1);
                for (int z = tid; z < dpad; z += TM_THREADS) q[z] = (q[z] * fjc + xs[z] / norm) / fjc1;
                __syncthreads();
                if (w == 0) {
                    float v = wave_dot((const float4 *)q, (const float4 *)q, nvec, lane);
                    if (lane == 0) s_tmp[2] = v;
                }
                jc++;
            }
        }
        __syncthreads();   // everyone is done with xs (and sees the new pp / qq)
        if (more) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                int v = tid + u * TM_THREADS;
                if (v < nvec) ((float4 *)xs)[v] = pf[u];
            }
            it = it_next;
        }
        __syncthreads();
    }
    // create_split: n = normalize(p - q)
    for (int z = tid; z < dpad; z += TM_THREADS) p[z] = p[z] - q[z];
    __syncthreads();
    lds_normalize(p, dpad, tid, lane, w, s_tmp);
    float4 *out = (float4 *)(hp + (int64_t)t.slot * dpad);
    for (int v = tid; v < nvec; v += TM_THREADS) out[v] = ((float4 *)p)[v];
}

// ---- two_means, one WAVE per node, everything in registers ------------------------
// For dpad <= 3072 the two centroids, the current row and the prefetched next row fit the
// register file (4 x NV float4 per lane): no LDS, no workgroup barrier in the 200-step loop.
// Same arithmetic, same order as two_means_kernel (the lane owns the same elements).

#define EW4(dst, expr_x, expr_y, expr_z, expr_w) \
    do { (dst).x = (expr_x); (dst).y = (expr_y); (dst).z = (expr_z); (dst).w = (expr_w); } while (0)

template <int NV>
__device__ inline float reg_dot(const float4 (&a)[NV], const float4 (&b)[NV])
{
    Acc4 s = acc4_zero();
#pragma unroll
    for (int k = 0; k < NV; k++) fma4(s, a[k], b[k]);   // pad elements are 0 in both: fmaf(0, 0, s) == s
    return acc4_finish(s);
}

// rows are padded to NV * 256 floats exactly (dpad is a multiple of 256): no lane is ever idle
template <int NV>
__device__ inline void reg_load_row(const float *row, int nvec, int lane, float4 (&x)[NV])
{
    const float4 *xp = (const float4 *)row;
#pragma unroll
    for (int k = 0; k < NV; k++) x[k] = xp[lane + k * WAVE];
}

template <int NV>
__device__ inline void reg_normalize(float4 (&v)[NV])
{
    const float norm = sqrtf(reg_dot<NV>(v, v));
    if (norm > 0.f) {
#pragma unroll
        for (int k = 0; k < NV; k++) EW4(v[k], v[k].x / norm, v[k].y / norm, v[k].z / norm, v[k].w / norm);
    }
}

// RN32(1 / n) for the counts a centroid can reach in TM_ITERS steps (1 + 200, then + 1): folded by the compiler,
// read through the scalar cache when a count changes.
struct RecipTable {
    float v[TM_ITERS + 8];
    constexpr RecipTable() : v()
    {
        for (int n = 1; n < TM_ITERS + 8; n++) v[n] = 1.0f / (float)n;
    }
};
__constant__ RecipTable k_recip = RecipTable();
__device__ inline float tm_recip(int n) { return k_recip.v[__builtin_amdgcn_readfirstlane(n)]; }   // n is wave-uniform

// One wave per node: both centroids and the current row in registers (3 x NV float4 per lane).  The row of the
// NEXT step is fetched by LDS-DMA (global_load_lds, no destination registers) at the top of a step and read into
// the row registers when the step's update is done: a register double buffer instead (NV = 12: 48 more VGPRs
// and 24 v_mov per step) does not fit two waves per SIMD without spilling, and a spill inside this loop waits
// for the whole prefetch (vmcnt is counted in order).
template <int NV>
__global__ __launch_bounds__(256, 3) void two_means_wave_kernel(const float *__restrict__ X, const RowInfo *__restrict__ rowinfo,
                                                             int64_t n_items, int32_t dpad,
                                                             const int32_t *__restrict__ perm,
                                                             const SplitTask *__restrict__ tasks, int32_t n_tasks,
                                                             uint32_t seed, float *__restrict__ hp, int32_t *__restrict__ ones)
{
    __shared__ float4 xnext[256 / WAVE][NV * WAVE];   // per wave: the row of the coming step
    const int lane = threadIdx.x & (WAVE - 1);
    // the wave number as a scalar: the task, the node's Kiss32 stream and the row base addresses are then
    // wave-uniform to the compiler (scalar registers)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int ti = blockIdx.x * (256 / WAVE) + wv;
    if (ti >= n_tasks) return;   // no barrier below: waves are independent
    if (lane == 0) ones[ti] = 0;   // the split kernels that follow count this task's right side here
    const SplitTask t = tasks[ti];
    const int nvec = dpad / 4;
    const int32_t *items = perm + TASK_ITEMS_AT(t, n_items);
    Kiss32 rng(node_seed(seed, (uint32_t)t.tree, (uint32_t)t.level, (uint32_t)t.start, (uint32_t)t.attempt));

    uint32_t i = rng.index((uint32_t)t.count);
    uint32_t j = rng.index((uint32_t)t.count - 1u);
    j += (j >= i);
    float4 p[NV], q[NV], x[NV];
    reg_load_row<NV>(X + (int64_t)items[i] * dpad, nvec, lane, p);
    reg_load_row<NV>(X + (int64_t)items[j] * dpad, nvec, lane, q);
    reg_normalize<NV>(p);
    reg_normalize<NV>(q);
    float pp = reg_dot<NV>(p, p), qq = reg_dot<NV>(q, q);
    uint32_t k = rng.index((uint32_t)t.count);
    const int32_t it = items[k];
    reg_load_row<NV>(X + (int64_t)it * dpad, nvec, lane, x);
    // The Kiss32 stream does not depend on data: the INDEX of row l+2 and the RowInfo of row l+1 are requested a step
    // before row l+1 itself, so that row's load never waits for its index (two dependent HBM round trips per step
    // otherwise bound the deep levels).  Draws past step 199 are never used: the stream is the node's own.
    // These look-ups go through the vector memory path (vgather) and come back to scalar registers when used.
    int32_t it_n1 = vgather(items, rng.index((uint32_t)t.count));
    RowInfo riv = vgather(rowinfo, (uint32_t)it);   // in flight: RowInfo of the row of the coming step
    float4 *xl = xnext[wv];

    int ic = 1, jc = 1;
    float r2p = tm_recip(2), r2q = tm_recip(2);   // RN32 of 1 / (ic + 1), 1 / (jc + 1), fetched when the count changes
    for (int l = 0; l < TM_ITERS; l++) {
        // this step's row: what was requested a step ago, now wave-uniform values in scalar registers
        const float nk2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(riv.norm2)));
        const int norm_bits = __builtin_amdgcn_readfirstlane(__float_as_int(riv.norm));
        const long long r1_bits = ((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(riv.rnorm) >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(riv.rnorm));
        const int32_t it_next = __builtin_amdgcn_readfirstlane(it_n1);
        {
            // row l+1 -> LDS while row l is used; the reads of the previous row out of this buffer have returned.
            // (After the last step this fetches a row nobody uses: the stream has more draws and every index is valid.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const float4 *src = (const float4 *)(X + (int64_t)it_next * dpad) + lane;
            // the instruction's immediate offset (< 4 KiB) moves both addresses: one base per four 1-KiB pieces
#define TM_GLDS(KK)                                                                                                       \
    if constexpr ((KK) < NV)                                                                                              \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ((KK) & ~3) * WAVE),      \
                                         (__attribute__((address_space(3))) void *)(xl + ((KK) & ~3) * WAVE), 16,         \
                                         ((KK) & 3) * 1024, 0)
            TM_GLDS(0); TM_GLDS(1); TM_GLDS(2); TM_GLDS(3); TM_GLDS(4); TM_GLDS(5);
            TM_GLDS(6); TM_GLDS(7); TM_GLDS(8); TM_GLDS(9); TM_GLDS(10); TM_GLDS(11);
#undef TM_GLDS
            static_assert(NV <= 12, "TM_GLDS list");
        }
        it_n1 = vgather(items, rng.index((uint32_t)t.count));   // index of row l+2
        riv = vgather(rowinfo, (uint32_t)it_next);
        const float di = (float)ic * ang_dist(pp, nk2, reg_dot<NV>(p, x));
        const float dj = (float)jc * ang_dist(qq, nk2, reg_dot<NV>(q, x));
        const float norm = __int_as_float(norm_bits & 0x7fffffff);   // sqrtf(nk2)
        const unsigned long long force = norm_bits < 0 ? ~0ull : 0ull;   // x / norm may be subnormal somewhere in this row
        const double r1 = __longlong_as_double(r1_bits);
        if (norm > 0.f) {
            if (di < dj) {
                const float f0 = (float)ic, f1 = (float)(ic + 1);
#pragma unroll
                for (int kk = 0; kk < NV; kk++) p[kk] = centroid_step4(p[kk], x[kk], f0, f1, norm, r1, r2p, force);
                pp = reg_dot<NV>(p, p);
                ic++;
                r2p = tm_recip(ic + 1);
            }
            // not `else if`: as two independent branches each centroid is updated in place; chained, the compiler
            // writes the new p to a second set of registers and copies p there on every step that leaves it alone
            if (dj < di) {
                const float f0 = (float)jc, f1 = (float)(jc + 1);
#pragma unroll
                for (int kk = 0; kk < NV; kk++) q[kk] = centroid_step4(q[kk], x[kk], f0, f1, norm, r1, r2q, force);
                qq = reg_dot<NV>(q, q);
                jc++;
                r2q = tm_recip(jc + 1);
            }
        }
        // the row registers are free: row l+1 has had the whole step to land in LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < NV; kk++) x[kk] = xl[kk * WAVE + lane];
    }
    // create_split: n = normalize(p - q)
#pragma unroll
    for (int kk = 0; kk < NV; kk++) EW4(p[kk], p[kk].x - q[kk].x, p[kk].y - q[kk].y, p[kk].z - q[kk].z, p[kk].w - q[kk].w);
    reg_normalize<NV>(p);
    float4 *out = (float4 *)(hp + (int64_t)t.slot * dpad);
#pragma unroll
    for (int kk = 0; kk < NV; kk++) out[lane + kk * WAVE] = p[kk];
}

// ---- two_means, four waves per node, by STRIPS of the canonical lanes ----------------------
// The shallow levels have fewer nodes than the chip has SIMDs and every node is one dependent chain of 200
// steps: what counts is the length of a step.  Here wave w of a node's workgroup owns the canonical lanes
// 16 w .. 16 w + 15 of every dot product: its lane l holds, for each 256-float k-step, element 64 w + l of the
// two centroids and of the row, i.e. chain (l & 3) of canonical lane 16 w + (l >> 2).  A dot is then
//   * the lane's ONE fmaf chain over the k-steps (the same chain wave_dot runs, four to a lane),
//   * the fold (a0 + a1) + (a2 + a3) across four neighbouring lanes (two DPP quad permutes),
//   * 64 folded values through LDS (one barrier), after which every wave runs the canonical butterfly on all of
//     them and holds the same result: same operations on the same values as wave_dot, bit for bit.
// Nothing else is exchanged and nothing is computed twice except the butterfly and the scalar distance
// arithmetic: a wave updates only its strip of the chosen centroid (12 elements per lane at D = 3000) and never
// sees the rest.  The dots of step l+1 (and the self-dot of the centroid just updated) are folded right after
// the update of step l, so a step has ONE barrier.  Rows are fetched DEPTH steps ahead into a register ring
// (a strip of a row is 12 registers), their index and RowInfo likewise: a step is several times shorter than an
// HBM round trip.  The loop is unrolled DEPTH times so that ring slots are compile-time registers.
template <int NV, int DEPTH>
__global__ __launch_bounds__(256) void two_means_strip_kernel(const float *__restrict__ X, const RowInfo *__restrict__ rowinfo,
                                                            int64_t n_items, int32_t dpad,
                                                            const int32_t *__restrict__ perm,
                                                            const SplitTask *__restrict__ tasks, uint32_t seed,
                                                            float *__restrict__ hp, int32_t *__restrict__ ones)
{
    static_assert(NV % 4 == 0, "the update works on groups of four k-steps");
    if (threadIdx.x == 0) ones[blockIdx.x] = 0;   // the split kernels that follow count this task's right side here
    static_assert(TM_ITERS % DEPTH == 0, "the step loop is unrolled DEPTH times");
    constexpr int NS = NV / 4;
    __shared__ float ex[2][3][WAVE];   // folded values of up to three dots per canonical lane, by exchange parity

    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int L = 16 * w + (lane >> 2);   // the canonical lane this lane's chain belongs to
    const SplitTask t = tasks[blockIdx.x];
    const int32_t *items = perm + TASK_ITEMS_AT(t, n_items);
    // every wave runs the node's Kiss32 stream itself: no index is exchanged
    Kiss32 rng(node_seed(seed, (uint32_t)t.tree, (uint32_t)t.level, (uint32_t)t.start, (uint32_t)t.attempt));
    const float *Xs = X + 64 * w + lane;   // this lane's element of k-step 0 of row 0

    // this lane's strip of a row: element 64 w + l of each k-step, four k-steps to a float4
    auto load_strip = [&](int32_t it, float4 (&dst)[NS]) {
        const float *r = Xs + (int64_t)it * dpad;
#pragma unroll
        for (int s = 0; s < NS; s++) EW4(dst[s], r[(4 * s) * 256], r[(4 * s + 1) * 256], r[(4 * s + 2) * 256], r[(4 * s + 3) * 256]);
    };
    // the lane's chain of a canonical dot: fmaf over the k-steps in order, from 0
    auto chain = [&](const float4 (&a)[NS], const float4 (&b)[NS]) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            acc = fmaf(a[s].x, b[s].x, acc);
            acc = fmaf(a[s].y, b[s].y, acc);
            acc = fmaf(a[s].z, b[s].z, acc);
            acc = fmaf(a[s].w, b[s].w, acc);
        }
        return acc;
    };
    int xpar = 0;
    // up to three dots at once: fold, exchange, butterfly.  Results in lanes 0 / 16 / 32 of the return value.
    auto exchange3 = [&](float a, float b, float c) {
        // (a0 + a1) + (a2 + a3) over the four chains of a canonical lane (quad_perm [1,0,3,2], then [2,3,0,1])
        a = a + dpp_move<0xB1>(a);
        b = b + dpp_move<0xB1>(b);
        c = c + dpp_move<0xB1>(c);
        a = a + dpp_move<0x4E>(a);
        b = b + dpp_move<0x4E>(b);
        c = c + dpp_move<0x4E>(c);
        if ((lane & 3) == 0) {
            ex[xpar][0][L] = a;
            ex[xpar][1][L] = b;
            ex[xpar][2][L] = c;
        }
        __syncthreads();
        const float f[3] = {ex[xpar][0][lane], ex[xpar][1][lane], ex[xpar][2][lane]};
        xpar ^= 1;   // the next exchange writes the other buffer: a wave may still be reading this one
        return wave_sum_multi<3>(f, lane);
    };
    auto lane_value = [](float u, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(u), l)); };

    uint32_t i = rng.index((uint32_t)t.count);
    uint32_t j = rng.index((uint32_t)t.count - 1u);
    j += (j >= i);
    float4 p[NS], q[NS];
    load_strip(items[i], p);
    load_strip(items[j], q);
    {   // normalise both (reg_normalize), then their self-dots
        const float u = exchange3(chain(p, p), chain(q, q), 0.f);
        const float np = sqrtf(lane_value(u, 0)), nq = sqrtf(lane_value(u, 16));
        if (np > 0.f) {
#pragma unroll
            for (int s = 0; s < NS; s++) EW4(p[s], p[s].x / np, p[s].y / np, p[s].z / np, p[s].w / np);
        }
        if (nq > 0.f) {
#pragma unroll
            for (int s = 0; s < NS; s++) EW4(q[s], q[s].x / nq, q[s].y / nq, q[s].z / nq, q[s].w / nq);
        }
    }
    float pp, qq;
    {
        const float u = exchange3(chain(p, p), chain(q, q), 0.f);
        pp = lane_value(u, 0);
        qq = lane_value(u, 16);
    }

    // ring slot r holds the row of step l with l % DEPTH == r, its item id and its RowInfo (still vector registers
    // when they arrive; made scalar when used); it_ahead is the item of the step DEPTH ahead of the one running
    float4 x[DEPTH][NS];
    RowInfo ri[DEPTH];
#pragma unroll
    for (int r = 0; r < DEPTH; r++) {
        const int32_t it = items[rng.index((uint32_t)t.count)];
        load_strip(it, x[r]);
        ri[r] = rowinfo[it];
    }
    int32_t it_ahead = vgather(items, rng.index((uint32_t)t.count));

    int ic = 1, jc = 1;
    float r2p = tm_recip(2), r2q = tm_recip(2);   // RN32 of 1 / (ic + 1), 1 / (jc + 1)
    float u = exchange3(chain(p, x[0]), chain(q, x[0]), 0.f);   // the dots of step 0
    int upd_prev = 0;
#ifdef MORNA_TM_PROBE
    long long c_decide = 0, c_update = 0, c_fetch = 0, c_dots = 0, c_t;
#define TMP(acc) do { const long long n_ = clock64(); acc += n_ - c_t; c_t = n_; } while (0)
    c_t = clock64();
#else
#define TMP(acc) do { } while (0)
#endif
    for (int l0 = 0; l0 < TM_ITERS; l0 += DEPTH) {
#pragma unroll
        for (int r = 0; r < DEPTH; r++) {
            // ---- decide (every wave: same values, same decision)
            if (upd_prev == 1) pp = lane_value(u, 32);
            if (upd_prev == 2) qq = lane_value(u, 32);
            const float nk2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ri[r].norm2)));
            const int norm_bits = __builtin_amdgcn_readfirstlane(__float_as_int(ri[r].norm));
            const long long r1_bits = ((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(ri[r].rnorm) >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(ri[r].rnorm));
            const float di = (float)ic * ang_dist(pp, nk2, lane_value(u, 0));
            const float dj = (float)jc * ang_dist(qq, nk2, lane_value(u, 16));
            const float norm = __int_as_float(norm_bits & 0x7fffffff);   // sqrtf(nk2)
            const unsigned long long force = norm_bits < 0 ? ~0ull : 0ull;
            const double r1 = __longlong_as_double(r1_bits);
            int upd = 0;
            if (norm > 0.f) upd = di < dj ? 1 : (dj < di ? 2 : 0);
            TMP(c_decide);
            // ---- update this wave's strip of the chosen centroid, and its chain of the new self-dot
            float cc = 0.f;
            if (upd == 1) {
                const float f0 = (float)ic, f1 = (float)(ic + 1);
#pragma unroll
                for (int s = 0; s < NS; s++) p[s] = centroid_step4(p[s], x[r][s], f0, f1, norm, r1, r2p, force);
                cc = chain(p, p);
                ic++;
                r2p = tm_recip(ic + 1);
            }
            if (upd == 2) {
                const float f0 = (float)jc, f1 = (float)(jc + 1);
#pragma unroll
                for (int s = 0; s < NS; s++) q[s] = centroid_step4(q[s], x[r][s], f0, f1, norm, r1, r2q, force);
                cc = chain(q, q);
                jc++;
                r2q = tm_recip(jc + 1);
            }
            upd_prev = upd;
            TMP(c_update);
            // ---- slot r is free: the row DEPTH steps ahead goes there.  (Past step 199 these fetch rows nobody
            // uses: the stream has more draws and every index is valid.)
            const int32_t it_new = it_ahead;
            load_strip(it_new, x[r]);
            ri[r] = vgather(rowinfo, (uint32_t)it_new);
            it_ahead = vgather(items, rng.index((uint32_t)t.count));
            TMP(c_fetch);
            // ---- the dots of the next step, with the row in the next slot
            constexpr int DUMMY = 0;
            (void)DUMMY;
            const int rn = (r + 1) % DEPTH;
            u = exchange3(chain(p, x[rn]), chain(q, x[rn]), cc);
            TMP(c_dots);
        }
    }
#ifdef MORNA_TM_PROBE
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("[tm_strip probe] cycles per step: decide %lld update %lld fetch %lld dots+exchange %lld (nodes %d)\n",
               c_decide / TM_ITERS, c_update / TM_ITERS, c_fetch / TM_ITERS, c_dots / TM_ITERS, (int)gridDim.x);
#endif
#undef TMP
    // create_split: n = normalize(p - q), each wave its strip
#pragma unroll
    for (int s = 0; s < NS; s++) EW4(p[s], p[s].x - q[s].x, p[s].y - q[s].y, p[s].z - q[s].z, p[s].w - q[s].w);
    {
        const float un = exchange3(chain(p, p), 0.f, 0.f);
        const float nn = sqrtf(lane_value(un, 0));
        if (nn > 0.f) {
#pragma unroll
            for (int s = 0; s < NS; s++) EW4(p[s], p[s].x / nn, p[s].y / nn, p[s].z / nn, p[s].w / nn);
        }
    }
    float *out = hp + (int64_t)t.slot * dpad + 64 * w + lane;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        out[(4 * s) * 256] = p[s].x;
        out[(4 * s + 1) * 256] = p[s].y;
        out[(4 * s + 2) * 256] = p[s].z;
        out[(4 * s + 3) * 256] = p[s].w;
    }
}

// ---------------------------------------------------------------- split kernel

#define SP_THREADS 256
#define SP_WAVES (SP_THREADS / WAVE)
#define SP_ROWS 64      // rows per workgroup

// ---- XCD-aware launch order of the split kernel ---------------------------------
// Every row is read once per TREE at every level, and a node's item list is sorted
// by id (stable partitions of the identity permutation).  Chunks are therefore
// ordered by the id of their first row, and the sorted list is cut into 8
// contiguous runs, one per XCD (workgroup b runs on XCD b % 8): the ~n_trees
// chunks that need the same rows then run back to back on ONE XCD and find them in
// its L2 instead of HBM.  The order only affects speed, never results.

#define SCHED_MAX_BUCKETS 65536

__global__ void sched_bucket_kernel(const SplitTask *__restrict__ tasks, int32_t n_tasks, int32_t n_chunks,
                                    const int32_t *__restrict__ perm, int64_t n_items, int32_t n_buckets,
                                    int32_t *__restrict__ hist, int2 *__restrict__ info /* [n_chunks] task, bucket */)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    int lo = 0, hi = n_tasks - 1;   // the task owning this chunk (tasks sorted by chunk0)
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (tasks[mid].chunk0 <= c) lo = mid; else hi = mid - 1;
    }
    const SplitTask t = tasks[lo];
    const int pos0 = (c - t.chunk0) * 64;
    const int64_t first = perm[TASK_ITEMS_AT(t, n_items) + pos0];
    const int b = (int)(first * n_buckets / n_items);
    atomicAdd(&hist[b], 1);
    info[c] = make_int2(lo, b);
}

__global__ __launch_bounds__(1024) void sched_scan_kernel(const int32_t *__restrict__ hist, int32_t n_buckets,
                                                          int32_t *__restrict__ cursor /* [n_buckets] start offsets */)
{
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    const int per = (n_buckets + 1023) / 1024;
    const int lo = tid * per, hi = lo + per < n_buckets ? lo + per : n_buckets;
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += hist[i];
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
        cursor[i] = run;
        run += hist[i];
    }
}

__global__ void sched_scatter_kernel(const SplitTask *__restrict__ tasks, int32_t n_chunks,
                                     const int2 *__restrict__ info, int32_t *__restrict__ cursor,
                                     int2 *__restrict__ sched /* [n_chunks] task, chunk in task */)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int2 e = info[c];
    const int pos = atomicAdd(&cursor[e.y], 1);
    sched[pos] = make_int2(e.x, c - tasks[e.x].chunk0);
}

// rows [chunk*SP_ROWS, ...) of one task get their side
__global__ __launch_bounds__(SP_THREADS) void split_kernel(const float *__restrict__ X, int64_t n_items, int32_t dpad,
                                                           const int32_t *__restrict__ perm,
                                                           const SplitTask *__restrict__ tasks,
                                                           const int2 *__restrict__ sched, int32_t n_chunks,
                                                           uint32_t seed, const float *__restrict__ hp,
                                                           uint8_t *__restrict__ side, int32_t *__restrict__ ones)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *hs = (float4 *)smem;   // hyperplane [dpad]
    __shared__ int s_ones;

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int nvec = dpad / 4;
    // workgroup b runs on XCD b % 8: XCD x walks the x-th eighth of the sorted chunk list
    const int per_xcd = (n_chunks + 7) / 8;
    const int si = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || si >= n_chunks) return;
    const int2 se = sched[si];
    const int lo = se.x;
    const SplitTask t = tasks[lo];
    const int pos0 = se.y * SP_ROWS;
    const int nrows = (t.count - pos0) < SP_ROWS ? (t.count - pos0) : SP_ROWS;

    const float4 *hsrc = (const float4 *)(hp + (int64_t)t.slot * dpad);
    for (int v = tid; v < nvec; v += SP_THREADS) hs[v] = hsrc[v];
    if (tid == 0) s_ones = 0;
    __syncthreads();

    const int32_t *items = perm + TASK_ITEMS_AT(t, n_items) + pos0;
    uint8_t *sd = side + (int64_t)t.tree * n_items + t.start + pos0;
    const uint32_t nseed = node_seed(seed, (uint32_t)t.tree, (uint32_t)t.level, (uint32_t)t.start, (uint32_t)t.attempt);
    int my_ones = 0;
    // A wave takes FOUR rows at a time: one LDS read of the hyperplane serves four rows, four row
    // streams are in flight, and the four lane-sums are reduced together (wave_sum_multi: row j's dot
    // arrives in lane 16 j, which writes its side).  Same chains per row as wave_dot.
    // Default cache policy on purpose: with non-temporal loads the rows are not kept in L2 for the
    // other trees' chunks that follow on this XCD (measured: 8.7 ms instead of 4.9 ms per level).
    for (int r = 4 * w; r < nrows; r += 4 * SP_WAVES) {
        const float4 *x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int rj = r + j < nrows ? r + j : r;   // a missing row repeats row r; its result is dropped
            x[j] = (const float4 *)(X + (int64_t)items[rj] * dpad);
        }
        Acc4 c[4] = {acc4_zero(), acc4_zero(), acc4_zero(), acc4_zero()};
#pragma unroll 2
        for (int i = lane; i < nvec; i += WAVE) {   // nvec is a multiple of 64: no lane idles
            const float4 x0 = x[0][i], x1 = x[1][i], x2 = x[2][i], x3 = x[3][i];
            const float4 hv = hs[i];
            fma4(c[0], x0, hv);
            fma4(c[1], x1, hv);
            fma4(c[2], x2, hv);
            fma4(c[3], x3, hv);
        }
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; j++) f[j] = (c[j].lo.x + c[j].lo.y) + (c[j].hi.x + c[j].hi.y);
        const float d = wave_sum_multi<4>(f, lane);
        const int rr = r + (lane >> 4);
        int s = 0;
        if ((lane & 15) == 0 && rr < nrows) {
            // Angular::side: dot != 0 ? dot > 0 : coin flip
            s = d != 0.f ? (d > 0.f) : pos_flip(nseed, (uint32_t)(pos0 + rr));
            sd[rr] = (uint8_t)s;
        }
        my_ones += __builtin_popcountll(__ballot(s));
    }
    if (lane == 0 && my_ones) atomicAdd(&s_ones, my_ones);
    __syncthreads();
    if (tid == 0 && s_ones) atomicAdd(&ones[lo], s_ones);
}

// ---- shallow levels: row-window form of the split kernel ---------------------------
// While a tree has at most RW_SLOTS / G split nodes, a workgroup takes a WINDOW of RW_ROWS
// consecutive row ids and a GROUP of G trees: the <= RW_SLOTS hyperplanes those trees
// can need sit in LDS, each wave loads a row ONCE into registers and dots it against
// the hyperplane of its node in every tree of the group.  L2 -> CU traffic per
// (row, tree) drops from one row to 1/G row; same wave_dot order, same results.

#define RW_THREADS 1024   // 16 waves: the kernel must stay within 128 VGPRs (one row buffer per wave)
#define RW_WAVES (RW_THREADS / WAVE)
#define RW_ROWS 256       // rows per window: the hyperplane staging is paid once per 16 rows of every wave
#define RW_SLOTS 12       // hyperplanes resident in LDS (144 KB at D = 3000): one workgroup per CU

// inverse of the permutation restricted to the tasks: row -> (task index, position in segment)
__global__ void invert_kernel(const SplitTask *__restrict__ tasks, int32_t n_tasks, int32_t n_chunks,
                              const int32_t *__restrict__ perm, int64_t n_items, int32_t *__restrict__ row_task,
                              int32_t *__restrict__ row_pos)
{
    const int c = blockIdx.x;
    if (c >= n_chunks) return;
    int lo = 0, hi = n_tasks - 1;   // the task owning this chunk (tasks are sorted by chunk0)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tasks[mid].chunk0 <= c) lo = mid; else hi = mid - 1;
    }
    const int a = lo;
    const SplitTask t = tasks[a];
    const int p = (c - t.chunk0) * 64 + threadIdx.x;
    if (p < t.count) {
        const int64_t row = perm[TASK_ITEMS_AT(t, n_items) + p];
        row_task[(int64_t)t.tree * n_items + row] = a;
        row_pos[(int64_t)t.tree * n_items + row] = p;
    }
}

template <int NV, int GT>   // NV float4 per lane hold one row (dpad / 256); GT trees per group
__global__ __launch_bounds__(RW_THREADS) void split_rw_kernel(
    const float *__restrict__ X, int64_t n_items, int32_t dpad, const SplitTask *__restrict__ tasks,
    const int32_t *__restrict__ tree_first /* [n_trees + 1] */, int32_t n_trees, int32_t G,
    const int32_t *__restrict__ row_task, const int32_t *__restrict__ row_pos, uint32_t seed,
    const float *__restrict__ hp, uint8_t *__restrict__ side, int32_t *__restrict__ ones, int32_t n_windows,
    int32_t n_groups)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *hs = (float4 *)smem;   // [RW_SLOTS][nvec]
    __shared__ int s_ones[RW_SLOTS], s_start[RW_SLOTS], s_hp[RW_SLOTS];
    __shared__ uint32_t s_seed[RW_SLOTS];

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int nvec = dpad / 4;
    // workgroup b runs on XCD b % 8: all tree groups of one window stay on one XCD, back to back
    const int x = blockIdx.x & 7, r = blockIdx.x >> 3;
    const int win = (r / n_groups) * 8 + x, g = r % n_groups;
    if (win >= n_windows) return;
    const int t0 = g * G, t1 = (t0 + G) < n_trees ? (t0 + G) : n_trees;
    const int a0 = tree_first[t0], nslots = tree_first[t1] - a0;   // <= RW_SLOTS by construction
    if (tid < nslots) {
        const SplitTask t = tasks[a0 + tid];
        s_ones[tid] = 0;
        s_start[tid] = t.start;
        s_hp[tid] = t.slot;
        s_seed[tid] = node_seed(seed, (uint32_t)t.tree, (uint32_t)t.level, (uint32_t)t.start, (uint32_t)t.attempt);
    }
    __syncthreads();
    // stage the group's hyperplanes as ONE flat copy: only one workgroup fits a CU, nothing else
    // would hide a slot-by-slot chain of dependent loads
    for (int idx = tid; idx < nslots * nvec; idx += RW_THREADS) {
        const int s = idx / nvec, v = idx - s * nvec;
        hs[idx] = ((const float4 *)(hp + (int64_t)s_hp[s] * dpad))[v];
    }
    __syncthreads();
    if (nslots > 0) {
        constexpr int LSTEP = WAVE / (GT == 3 ? 4 : GT);
        const int ng = t1 - t0;
        const int64_t row_base = (int64_t)win * RW_ROWS;
        // a row and its (task, position) per tree of the group are fetched one row ahead
        auto load_row = [&](int rr, float4(&xr)[NV], int &ma, int &mp) {
            const int64_t row = row_base + rr;
            const float4 *xp = (const float4 *)(X + row * dpad);
#pragma unroll
            for (int k = 0; k < NV; k++) xr[k] = xp[lane + k * WAVE];   // dpad == NV * 256: no idle lane
            ma = -1;
            mp = 0;
            if (lane % LSTEP == 0 && lane / LSTEP < ng) {   // lane gi * LSTEP looks after tree gi of the group
                ma = row_task[(int64_t)(t0 + lane / LSTEP) * n_items + row];
                mp = row_pos[(int64_t)(t0 + lane / LSTEP) * n_items + row];
            }
        };
        // the row against its node's hyperplane in every tree of the group, branch-free and with k
        // outermost: GT independent FMA chains are in flight, a tree that does not own the row reads
        // slot 0 and its result is dropped.  The GT lane-sums are reduced together
        // (wave_sum_multi): tree gi's dot arrives in lane gi * LSTEP, the lane that holds the
        // row's (task, position) for that tree and writes its side byte.
        auto process = [&](const float4(&xr)[NV], int ma, int mp) {
            const float4 *hv[GT];
            Acc4 c[GT];
#pragma unroll
            for (int gi = 0; gi < GT; gi++) {
                const int a = __builtin_amdgcn_readlane(ma, gi * LSTEP);   // lanes of absent trees hold -1
                hv[gi] = hs + (a < 0 ? 0 : a - a0) * nvec + lane;
                c[gi] = acc4_zero();
            }
#pragma unroll
            for (int k = 0; k < NV; k++) {
#pragma unroll
                for (int gi = 0; gi < GT; gi++) fma4(c[gi], xr[k], hv[gi][k * WAVE]);
            }
            float f[GT];
#pragma unroll
            for (int gi = 0; gi < GT; gi++) f[gi] = (c[gi].lo.x + c[gi].lo.y) + (c[gi].hi.x + c[gi].hi.y);
            const float d = wave_sum_multi<GT>(f, lane);
            if (ma >= 0) {   // only lanes gi * LSTEP of present trees
                const int s = ma - a0;
                const int sd = d != 0.f ? (d > 0.f) : pos_flip(s_seed[s], (uint32_t)mp);
                side[(int64_t)(t0 + lane / LSTEP) * n_items + s_start[s] + mp] = (uint8_t)sd;
                if (sd) atomicAdd(&s_ones[s], 1);
            }
        };
        auto valid = [&](int rr) { return rr < RW_ROWS && row_base + rr < n_items; };
        // ONE row buffer per wave (<= 128 VGPRs), 16 waves per CU: the other three waves of the SIMD cover
        // a wave's load latency.  Measured against two waves per SIMD with three row buffers in rotation
        // (prefetch distance 2): 2.06 / 2.11 / 3.08 ms instead of 2.29 / 2.70 / 3.70 ms per level.
        float4 xr[NV];
        int ma = -1, mp = 0;
        for (int rr = w; valid(rr); rr += RW_WAVES) {
            load_row(rr, xr, ma, mp);
            process(xr, ma, mp);
        }
    }
    __syncthreads();
    if (tid < nslots && s_ones[tid]) atomicAdd(&ones[a0 + tid], s_ones[tid]);
}

// random sides for nodes whose best split is still > 0.99 imbalanced
__global__ __launch_bounds__(256) void fallback_kernel(const SplitTask *__restrict__ tasks, int64_t n_items,
                                                       int32_t dpad, uint32_t seed, uint8_t *__restrict__ side,
                                                       int32_t *__restrict__ ones, float *__restrict__ hp)
{
    __shared__ int s_cnt;
    const SplitTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x;
    uint8_t *sd = side + (int64_t)t.tree * n_items + t.start;
    for (int z = tid; z < dpad; z += 256) hp[(int64_t)t.slot * dpad + z] = 0.f;   // m->v = 0
    for (int round = 0;; round++) {
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        int mine = 0;
        if (round >= 32) {   // give up on chance: halve by position
            for (int p = tid; p < t.count; p += 256) {
                int s = p >= t.count / 2;
                sd[p] = (uint8_t)s;
                mine += s;
            }
        } else {
            const uint32_t ns = node_seed(seed, (uint32_t)t.tree, (uint32_t)t.level, (uint32_t)t.start, (uint32_t)(3 + round));
            for (int p = tid; p < t.count; p += 256) {
                int s = pos_flip(ns, (uint32_t)p);
                sd[p] = (uint8_t)s;
                mine += s;
            }
        }
        if (mine) atomicAdd(&s_cnt, mine);
        __syncthreads();
        const int n1 = s_cnt;
        __syncthreads();
        if (round >= 32 || !(split_imbalance(t.count - n1, n1) > 0.99)) {
            if (tid == 0) ones[blockIdx.x] = n1;
            break;
        }
    }
}

// annoy's rule for the sides of one attempt (_make_tree): attempts 0 and 1 stand when the imbalance is below 0.95,
// the last attempt (2) unless it is above 0.99 (then the node gets random sides: tasks of "attempt" 3, which always
// stand).  The host driver applies the same expressions to the same counts.
__host__ __device__ inline bool sides_stand(int attempt, int64_t n0, int64_t n1)
{
    if (attempt >= 3) return true;
    const double imb = split_imbalance(n0, n1);
    return attempt == 2 ? !(imb > 0.99) : imb < 0.95;
}

// stable partition of one segment by side; one workgroup per node (PT threads: 1024 while nodes are large).
// Launched right behind the split of every attempt, before the host has seen the counts: a node whose sides do
// not stand is left alone (its next attempt, or the fallback, partitions it).
// inv (may be null): inv[tree][item] = position of the item in the tree's permutation, kept current here -- what the
// matrix-core split looks a row's node up with (one look-up per (row, tree), no separate inversion pass per level).
// The right-side counts of a level's tasks, posted to page-locked host memory tagged with the call's epoch: a launch of its
// own in front of the partition, so that ALL counts are with the host a few microseconds after the split and its
// bookkeeping for the next attempt or level runs while the partition does (posted by the partition workgroups themselves,
// the last counts arrived when the last workgroups started, near the partition's end: 20-65 us of idle device per level).
__global__ void post_counts_kernel(const int32_t *__restrict__ ones, int32_t n, unsigned long long *__restrict__ host_counts,
                                   uint32_t epoch)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // relaxed: the word carries everything the host reads (a release here would write back this CU's whole L2 share)
    if (i < n)
        __hip_atomic_store(&host_counts[i], ((unsigned long long)epoch << 32) | (unsigned int)ones[i], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// the level's task list, pulled out of page-locked host memory by the device itself: a hipMemcpyAsync of more than a few KB
// goes to the SDMA engine, ~20 us of start-up during which the stream has nothing else to run
__global__ void pull_tasks_kernel(const int4 *__restrict__ src /* host memory, device-visible */, int4 *__restrict__ dst, int32_t n16)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

template <int PT>
__global__ __launch_bounds__(PT) void partition_kernel(const SplitTask *__restrict__ tasks, int64_t n_items,
                                                       const uint8_t *__restrict__ side,
                                                       const int32_t *__restrict__ ones, int32_t *__restrict__ perm /* work buffer [tree][2][n_items] */,
                                                       int32_t *__restrict__ inv,
                                                       const int32_t *__restrict__ rank /* inv is indexed by rank[item] (or null: by item) */)
{
    __shared__ int s_w1[PT / WAVE];
    const SplitTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), w = tid / WAVE;
    const int n1 = ones[blockIdx.x], n0 = t.count - n1;
    if (!sides_stand(t.attempt, n0, n1)) return;   // uniform over the workgroup
    const int64_t base = (int64_t)t.tree * n_items + t.start;   // of the node's side bytes (and of its positions in `inv`)
    // the node's items are read from the image of its depth and its children are written into the other one: no copy back
    const int32_t *src = perm + TASK_ITEMS_AT(t, n_items);
    int32_t *dst_img = perm + ((int64_t)t.tree * 2 + ((t.level + 1) & 1)) * n_items + t.start;
    // PT_PER consecutive positions per thread and round: a round costs two barriers whatever it moves, and with one
    // position per thread the root level took 49 of them (0.08 ms).  Ranks: right-side and valid counts of a thread
    // packed into one integer (16 bits each: a round has at most 4096 positions), scanned over the wave, wave totals
    // through LDS.
    constexpr int PT_PER = 4;
    int run0 = 0, run1 = 0;   // items already placed on each side
    // a round's sides and items are asked for one round ahead (they depend on nothing the round before computes): the
    // kernel is a chain of memory round trips, three per round when each waits for the one before
    int sd_n[PT_PER], it_n[PT_PER];
    auto fetch = [&](int p0) {
        const int pb = p0 + tid * PT_PER;
#pragma unroll
        for (int u = 0; u < PT_PER; u++) {
            const bool valid = pb + u < t.count;
            sd_n[u] = valid ? (int)side[base + pb + u] : -1;
            it_n[u] = valid ? src[pb + u] : 0;
        }
    };
    fetch(0);
    for (int p0 = 0; p0 < t.count; p0 += PT * PT_PER) {
        int sd[PT_PER], it[PT_PER], pk = 0;
#pragma unroll
        for (int u = 0; u < PT_PER; u++) {
            sd[u] = sd_n[u];
            it[u] = it_n[u];
            pk += sd[u] >= 0 ? (sd[u] ? 0x10001 : 0x10000) : 0;
        }
        if (p0 + PT * PT_PER < t.count) fetch(p0 + PT * PT_PER);   // uniform
        int incl = pk;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const int v = __shfl_up(incl, o, WAVE);
            if (lane >= o) incl += v;
        }
        if (lane == WAVE - 1) s_w1[w] = incl;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < PT / WAVE; i++) {
            if (i < w) before += s_w1[i];
            total += s_w1[i];
        }
        const int excl = incl - pk + before;
        int r1 = excl & 0xffff, rv = excl >> 16;   // right-side / valid positions before this thread's first
        int rk[PT_PER];
#pragma unroll
        for (int u = 0; u < PT_PER; u++) rk[u] = (rank && sd[u] >= 0) ? rank[it[u]] : it[u];   // the four look-ups together
#pragma unroll
        for (int u = 0; u < PT_PER; u++)
            if (sd[u] >= 0) {
                const int dst = sd[u] ? (n0 + run1 + r1) : (run0 + (rv - r1));
                dst_img[dst] = it[u];
                if (inv) inv[(int64_t)t.tree * n_items + rk[u]] = t.start + dst;
                r1 += sd[u];
                rv += 1;
            }
        run1 += total & 0xffff;
        run0 += (total >> 16) - (total & 0xffff);
        __syncthreads();
    }
}

// roots: image 0 of every tree = the identity
__global__ void iota_perm_kernel(int32_t *perm /* work buffer [tree][2][n_items] */, int32_t *inv, int64_t n_items, int64_t total)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const int64_t t = i / n_items, r = i - t * n_items;
        perm[t * 2 * n_items + r] = (int32_t)r;
        if (inv) inv[i] = (int32_t)r;
    }
}

// the finished forest's leaves, out of the image of their depth into the permutation the searches read: one workgroup per leaf
struct LeafSeg {
    int32_t tree, level, start, count;
};
__global__ __launch_bounds__(256) void gather_leaves_kernel(const LeafSeg *__restrict__ leaves, const int32_t *__restrict__ work,
                                                            int64_t n_items, int32_t *__restrict__ perm)
{
    const LeafSeg l = leaves[blockIdx.x];
    const int32_t *src = work + ((int64_t)l.tree * 2 + (l.level & 1)) * n_items + l.start;
    int32_t *dst = perm + (int64_t)l.tree * n_items + l.start;
    for (int i = threadIdx.x; i < l.count; i += 256) dst[i] = src[i];
}

// -------------------------------------------------------------- host driver

int build_forest(morna_index *h, int32_t n_trees, uint32_t seed)
{
    if (h->ev_tables_pending) {   // a previous build may still be copying the host node tables this one is about to clear
        HIP_TRY(hipEventSynchronize(h->ev_tables));   // (only those copies: what was enqueued since is not waited for)
        h->ev_tables_pending = false;
    }
    MORNA_TRY(upload_host_rows(h));
    if (h->n_items <= 0) {
        set_error("no items were added to the index before build()");
        return MORNA_E_EMPTY;
    }
    if (n_trees <= 0) {
        set_error("build: n_trees must be positive (annoy's q = -1 auto mode is not used by morna)");
        return MORNA_E_INVALID;
    }
    if (!h->norms_valid) MORNA_TRY(compute_norms(h));
    if (seed == 0) seed = 123456789u;   // Kiss32Random's default seed
    const int64_t N = h->n_items;
    const int32_t dpad = h->dpad, K = h->K, D = h->dim;
    if (dpad > 8192) {   // two_means prefetches a row in 8 float4 per thread; 3 rows of LDS
        set_error("build: dimension %d is above the supported 8192", D);
        return MORNA_E_INVALID;
    }
    h->built = false;
    h->n_trees = n_trees;
    h->seed = seed;
    h->stats = morna_forest_stats();
    h->stats.n_items = N;
    h->stats.dim = D;
    h->stats.leaf_capacity = K;
    h->stats.n_trees = n_trees;

    MORNA_TRY(h->perm.alloc((size_t)n_trees * N));
    // scratch lives in the handle: a rebuild (or the next level) reuses it without hipMalloc
    ScratchRef<int32_t> work(h->scratch[8]), d_ones(h->scratch[9]);   // work: two images of every tree's permutation (TASK_ITEMS_AT)
    ScratchRef<uint8_t> side(h->scratch[10]);
    ScratchRef<SplitTask> d_tasks(h->scratch[11]);
    MORNA_TRY(work.alloc((size_t)n_trees * N * 2));
    MORNA_TRY(side.alloc((size_t)n_trees * N));
    // launch-order scratch of the split kernel: buckets of 32 row ids
    ScratchRef<int32_t> d_hist(h->scratch[12]), d_cursor(h->scratch[13]);
    ScratchRef<int2> d_info(h->scratch[14]), d_sched(h->scratch[15]);
    const int32_t n_buckets = (int32_t)std::min<int64_t>(SCHED_MAX_BUCKETS, std::max<int64_t>(1, (N + 31) / 32));
    MORNA_TRY(d_hist.alloc((size_t)n_buckets));
    MORNA_TRY(d_cursor.alloc((size_t)n_buckets));
    // row-window form of the shallow levels: inverse permutation per tree
    ScratchRef<int32_t> row_task(h->scratch[16]), row_pos(h->scratch[17]), d_tree_first(h->scratch[18]);
    // matrix-core split: position of every item in every tree's permutation, kept current by partition_kernel
    static const bool mm_on = !(getenv("MORNA_SPLIT_MM") && atoi(getenv("MORNA_SPLIT_MM")) == 0);
    static const bool order_on = !(getenv("MORNA_SPLIT_ORDER") && atoi(getenv("MORNA_SPLIT_ORDER")) == 0);
    ScratchRef<int32_t> inv(h->scratch[26]);
    if (mm_on) MORNA_TRY(inv.alloc((size_t)n_trees * N));
    int32_t *inv_p = mm_on ? inv.p : nullptr;   // by item; by row once the rows have been ordered (rank_p)
    const int32_t *rank_p = nullptr;
    h->ord_valid = false;
    std::vector<int32_t> tree_first;
    {
        const int64_t total = (int64_t)n_trees * N;
        hipLaunchKernelGGL(iota_perm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, work.p, inv_p, N, total);
        HIP_TRY(hipGetLastError());
    }

    // host-side node table, breadth-first ids; roots are 0..T-1
    std::vector<int32_t> &rec = h->h_node_rec, &ntree = h->h_node_tree, &nhp = h->h_node_hp;
    rec.clear(); ntree.clear(); nhp.clear();
    std::vector<Seg> cur((size_t)n_trees), nxt;
    for (int t = 0; t < n_trees; t++) {
        cur[(size_t)t] = Seg{t, 0, 0, (int32_t)N, t};
        rec.insert(rec.end(), {-1, -1, 0, (int32_t)N});
        ntree.push_back(t);
        nhp.push_back(-1);
    }
    // hyperplanes are produced level by level straight into h->hp, which grows geometrically
    // (and is kept across rebuilds); slot = n_split_total + index within the level
    int64_t n_split_total = 0;
    int rc = MORNA_OK;
    std::vector<SplitTask> tasks;
    std::vector<int32_t> h_ones;
    int32_t level = 0;

    auto cleanup = [&]() { (void)hipStreamSynchronize(h->stream); };   // the last partition may still be running
#define F_TRY(e)                                                    \
    do {                                                            \
        hipError_t _e = (e);                                        \
        if (_e != hipSuccess) {                                     \
            set_error("%s failed: %s", #e, hipGetErrorString(_e));  \
            cleanup();                                              \
            return MORNA_E_HIP;                                     \
        }                                                           \
    } while (0)

    // the node-table entries of the level just finished, written once the next level is under way (or at the end)
    struct LateTables {
        std::vector<Seg> parents;
        std::vector<int32_t> ones;
        int32_t first_id = 0;
        int64_t first_hp = 0;
        bool pending = false;
    } late;
    int32_t next_node_id = (int32_t)ntree.size();
    std::vector<LeafSeg> leaves;   // where each finished leaf's items are (gather_leaves_kernel)
    auto write_late_tables = [&]() {
        if (!late.pending) return;
        late.pending = false;
        const size_t S2 = late.parents.size();
        rec.reserve(rec.size() + S2 * 8);
        ntree.reserve(ntree.size() + S2 * 2);
        nhp.reserve(nhp.size() + S2 * 2);
        for (size_t i = 0; i < S2; i++) {
            const Seg &s = late.parents[i];
            const int32_t n1 = late.ones[i], n0 = s.count - n1;
            const int32_t id0 = late.first_id + 2 * (int32_t)i, id1 = id0 + 1;
            rec[(size_t)s.node * 4 + 0] = id0;
            rec[(size_t)s.node * 4 + 1] = id1;
            nhp[(size_t)s.node] = (int32_t)(late.first_hp + (int64_t)i);
            rec.insert(rec.end(), {-1, -1, s.start, n0});
            rec.insert(rec.end(), {-1, -1, s.start + n0, n1});
            ntree.push_back(s.tree);
            ntree.push_back(s.tree);
            nhp.push_back(-1);
            nhp.push_back(-1);
        }
    };
    while (!cur.empty()) {
        // split nodes of this level, in cur order
        std::vector<int32_t> split_idx;
        for (size_t i = 0; i < cur.size(); i++) {
            if (cur[i].count > K) split_idx.push_back((int32_t)i);
            else leaves.push_back(LeafSeg{cur[i].tree, cur[i].level, cur[i].start, cur[i].count});
        }
        h->stats.max_depth = std::max<int64_t>(h->stats.max_depth, level);
        if (split_idx.empty()) break;
        const int32_t S = (int32_t)split_idx.size();
        {
            const size_t need = (size_t)(n_split_total + S) * dpad;
            if (need > h->hp.n) {
                F_TRY(hipStreamSynchronize(h->stream));   // the previous level's partition may still be running
                DevBuf<float> bigger;
                if ((rc = bigger.alloc(std::max(need, h->hp.n * 2)))) return rc;
                if (n_split_total > 0)
                    F_TRY(hipMemcpy(bigger.p, h->hp.p, (size_t)n_split_total * dpad * 4, hipMemcpyDeviceToDevice));
                h->hp.release();
                h->hp.p = bigger.p;
                h->hp.n = bigger.n;
                bigger.p = nullptr;
                bigger.n = 0;
            }
        }
        float *hp_level = h->hp.p + (size_t)n_split_total * dpad;

        // final outcome per split node of this level
        std::vector<int32_t> final_ones((size_t)S, 0);
        std::vector<int32_t> pending((size_t)S);
        for (int32_t i = 0; i < S; i++) pending[(size_t)i] = i;
        if ((rc = d_tasks.alloc((size_t)S)) || (rc = d_ones.alloc((size_t)S))) { cleanup(); return rc; }

        auto make_tasks = [&](const std::vector<int32_t> &which, int attempt, std::vector<SplitTask> &out) {
            out.resize(which.size());
            int32_t chunk = 0;
            for (size_t a = 0; a < which.size(); a++) {
                const Seg &s = cur[(size_t)split_idx[(size_t)which[a]]];
                out[a] = SplitTask{s.tree, s.level, s.start, s.count, which[a], attempt, chunk, 0};
                chunk += (s.count + SP_ROWS - 1) / SP_ROWS;
            }
            return chunk;
        };

        // Partition of the nodes whose sides stand, enqueued right behind the kernels that wrote the sides and their
        // counts (d_tasks / d_ones as they are on the device).  The counts travel to the host on the side stream,
        // from the moment the split is done: the host's bookkeeping for the next attempt or level runs while the
        // partition does.
        // Partition of the nodes whose sides stand, enqueued right behind the kernels that wrote the sides and their
        // counts (d_tasks / d_ones as they are on the device).  The counts reach the host through page-locked memory
        // that every partition workgroup writes before anything else, tagged with this call's epoch: the host polls the
        // tags (no event hand-over, no copy, no stream wait) and prepares the next attempt or level while the partition runs.
        auto partition_and_fetch_counts = [&](int32_t A, int64_t level_rows) -> int {
            if ((size_t)A > h->host_counts_cap) {
                if (hipStreamSynchronize(h->stream) != hipSuccess) return MORNA_E_HIP;   // nobody writes the old buffer any more
                if (h->host_counts) (void)hipHostFree(h->host_counts);
                h->host_counts = nullptr;
                h->host_counts_cap = std::max<size_t>(4096, (size_t)A * 2);
                if (hipHostMalloc((void **)&h->host_counts, h->host_counts_cap * 8, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
                    h->host_counts_cap = 0;
                    set_error("forest build: hipHostMalloc of the count mailbox failed");
                    return MORNA_E_HIP;
                }
                memset(h->host_counts, 0, h->host_counts_cap * 8);
            }
            const uint32_t epoch = ++h->count_epoch ? h->count_epoch : ++h->count_epoch;   // never 0: the mailbox starts zeroed
            hipLaunchKernelGGL(post_counts_kernel, dim3((unsigned)((A + 255) / 256)), dim3(256), 0, h->stream, d_ones.p, A, h->host_counts,
                               epoch);
            {
                ScopedTimer tm(h, MORNA_T_PARTITION, 0);
                if (level_rows >= (int64_t)A * 2048)
                    hipLaunchKernelGGL(partition_kernel<1024>, dim3((unsigned)A), dim3(1024), 0, h->stream, d_tasks.p, N, side.p,
                                       d_ones.p, work.p, inv_p, rank_p);
                else
                    hipLaunchKernelGGL(partition_kernel<256>, dim3((unsigned)A), dim3(256), 0, h->stream, d_tasks.p, N, side.p,
                                       d_ones.p, work.p, inv_p, rank_p);
            }
            if (hipGetLastError() != hipSuccess) {
                set_error("forest build: partition launch failed");
                return MORNA_E_HIP;
            }
            write_late_tables();   // the previous level's node-table entries, while this level's kernels run
            h_ones.resize((size_t)A);
            volatile unsigned long long *box = h->host_counts;
            int64_t spins = 0;
            for (int32_t a = 0; a < A; a++) {
                unsigned long long v;
                while ((uint32_t)((v = box[a]) >> 32) != epoch) {
                    if (++spins > (int64_t)1 << 22) {
                        // not there after a long while: let the stream say what happened (a failed launch, a fault), or
                        // simply wait for a very large level the slow way
                        if (hipStreamSynchronize(h->stream) != hipSuccess) {
                            set_error("forest build: the partition kernel failed");
                            return MORNA_E_HIP;
                        }
                        spins = 0;
                    }
                    __builtin_ia32_pause();
                }
                h_ones[(size_t)a] = (int32_t)(uint32_t)v;
            }
            return MORNA_OK;
        };

        // the level's task list goes up through page-locked staging (the copy of an earlier attempt has long run: its
        // counts have been read back since)
        auto upload_tasks = [&](const std::vector<SplitTask> &tk) -> int {
            const size_t bytes = tk.size() * sizeof(SplitTask);
            if (bytes > h->host_tables_cap) {
                if (hipStreamSynchronize(h->stream) != hipSuccess) return MORNA_E_HIP;
                if (h->host_tables) (void)hipHostFree(h->host_tables);
                h->host_tables = nullptr;
                h->host_tables_cap = 0;
                if (hipHostMalloc((void **)&h->host_tables, std::max<size_t>(bytes * 2, 1 << 16), hipHostMallocMapped) != hipSuccess) {
                    set_error("forest build: hipHostMalloc of the task staging failed");
                    return MORNA_E_HIP;
                }
                h->host_tables_cap = std::max<size_t>(bytes * 2, 1 << 16);
            }
            memcpy(h->host_tables, tk.data(), bytes);
            static_assert(sizeof(SplitTask) % 16 == 0, "tasks are moved in 16-byte words");
            void *dev_view = nullptr;
            if (hipHostGetDevicePointer(&dev_view, h->host_tables, 0) != hipSuccess) {
                set_error("forest build: the task staging is not visible to the device");
                return MORNA_E_HIP;
            }
            const int32_t n16 = (int32_t)(bytes / 16);
            if (n16 > 0) hipLaunchKernelGGL(pull_tasks_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, h->stream, (const int4 *)dev_view, (int4 *)d_tasks.p, n16);
            if (hipGetLastError() != hipSuccess) {
                set_error("forest build: task upload failed");
                return MORNA_E_HIP;
            }
            return MORNA_OK;
        };

        for (int attempt = 0; attempt < 3 && !pending.empty(); attempt++) {
            const int32_t n_chunks = make_tasks(pending, attempt, tasks);
            const int32_t A = (int32_t)tasks.size();
            int64_t rows = 0;
            for (const SplitTask &t : tasks) rows += t.count;
            // shallow level, whole level pending: row-window form (rows reused across a tree group)
            int max_per_tree = 0;
            tree_first.assign((size_t)n_trees + 1, A);
            for (int32_t a = A - 1; a >= 0; a--) tree_first[(size_t)tasks[(size_t)a].tree] = a;
            for (int t = n_trees - 1; t >= 0; t--)
                if (tree_first[(size_t)t] > tree_first[(size_t)t + 1]) tree_first[(size_t)t] = tree_first[(size_t)t + 1];
            for (int t = 0; t < n_trees; t++)
                max_per_tree = std::max(max_per_tree, tree_first[(size_t)t + 1] - tree_first[(size_t)t]);
            const int nv = (dpad / 4 + WAVE - 1) / WAVE;
            // measured on MI355X (C3, 1e7 rows per level): 2.4 / 2.8 / 3.8 ms with 1 / 2 / 4 nodes per tree
            // against 4.7 ms for the chunk form, so it is used while a tree has at most 4 split nodes
            // (MORNA_SPLIT_RW=0 turns it off, =2 restricts it to 2 nodes per tree)
            static const int rw_max = getenv("MORNA_SPLIT_RW") ? atoi(getenv("MORNA_SPLIT_RW")) : 4;
            const bool nv_ok = nv == 1 || nv == 2 || nv == 3 || nv == 4 || nv == 6 || nv == 8 || nv == 12;
            const bool use_rw = attempt == 0 && max_per_tree >= 1 && max_per_tree <= std::min(rw_max, RW_SLOTS / 2) &&
                                nv_ok && rows * 2 >= (int64_t)n_trees * N;
            // While a tree has few split nodes and most rows still sit in split nodes, the whole level is one
            // contraction on the matrix cores (splitmm.hip): its cost grows with the nodes per tree (every row
            // meets every hyperplane), the chunk form's does not -- they meet near 64 nodes per tree.
            // MORNA_SPLIT_MM=0 turns it off (row-window / chunk forms as before).
            // Once the rows of the contraction are ordered and the level has enough tasks for the per-tile lists
            // (splitmm.hip), a row tile meets only the tasks that hold one of its rows, and the cost follows the (tile,
            // task) pairs that exist: then deeper levels (up to 256 nodes per tree) and the retries of a level go there too,
            // as does any set of tasks whose all-pairs product in 128 x 128 tiles is cheaper than streaming the rows once
            // per tree (measured at D = 3000: 0.13 us per tile pair, 0.47 ns per row of the chunk form, both ~ dpad).
            // On a shard of 200k x 3000 the chunk form took 28 of the split's 34 ms, retries and levels 7+.
            static const bool lists_env = !(getenv("MORNA_SPLIT_LISTS") && atoi(getenv("MORNA_SPLIT_LISTS")) == 0);
            const bool mm_lists = lists_env && h->ord_valid && A >= 512 && max_per_tree <= 256;
            const double mm_dense_us = (double)((N + 127) / 128) * (double)((A + 127) / 128) * 0.13, chunk_us = (double)rows * 0.47e-3;
            const bool use_mm = mm_on && max_per_tree >= 1 &&
                                (mm_lists || (max_per_tree <= 32 && (attempt == 0 ? rows * 2 >= (int64_t)n_trees * N : mm_dense_us < chunk_us)));
            if ((rc = upload_tasks(tasks))) { cleanup(); return rc; }
            // (d_ones is zeroed by the two_means kernel of the attempt, task by task: a hipMemsetAsync costs ~15 us of
            // idle device around its few microseconds)
            bool side_work = false;
            if (use_mm && !h->half_valid) {
                // once per set of rows: their fp16 image, on the side stream while two_means (a latency chain on few CUs
                // at the root level) has the main one.  Nothing else of the matrix-core split needs a side stream: a
                // row's node is looked up through `inv`, which the partition of the previous level left current (round 1
                // inverted the permutations in a kernel of its own per level, behind two event hand-overs of ~17 us each).
                F_TRY(hipEventRecord(h->ev_fork, h->stream));
                F_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
                if ((rc = split_mm_prepare_rows(h, h->stream2))) { cleanup(); return rc; }
                F_TRY(hipEventRecord(h->ev_join, h->stream2));
                side_work = true;
            } else if (use_mm && order_on && level == 1 && !h->ord_valid && N >= 8192 && N <= ((int64_t)1 << 22) && n_trees >= 2) {
                // second level: the rows of the contraction are put in an order in which neighbours are alike (the sides of
                // the root splits say which are), on the side stream under this level's two_means; from here on `inv` is
                // kept by row (splitmm.hip, split_mm_order_rows; its counting sort's table is 16 KB per 256 rows: up to 4 M rows)
                F_TRY(hipEventRecord(h->ev_fork, h->stream));
                F_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
                if ((rc = split_mm_order_rows(h, side.p, inv_p, n_trees, h->stream2, &rank_p, &inv_p))) { cleanup(); return rc; }
                F_TRY(hipEventRecord(h->ev_join, h->stream2));
                side_work = true;
            } else if (use_rw && !use_mm) {
                // row-window form (MORNA_SPLIT_MM=0): row -> (task, position) per tree on the side stream
                if ((rc = row_task.alloc((size_t)n_trees * N)) || (rc = row_pos.alloc((size_t)n_trees * N)) ||
                    (rc = d_tree_first.alloc((size_t)n_trees + 1))) { cleanup(); return rc; }
                F_TRY(hipEventRecord(h->ev_fork, h->stream));
                F_TRY(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
                F_TRY(hipMemcpyAsync(d_tree_first.p, tree_first.data(), ((size_t)n_trees + 1) * 4, hipMemcpyHostToDevice, h->stream2));
                if (rows != (int64_t)n_trees * N)   // rows outside every split node must read "no task"
                    F_TRY(hipMemsetAsync(row_task.p, 0xFF, (size_t)n_trees * N * 4, h->stream2));
                hipLaunchKernelGGL(invert_kernel, dim3((unsigned)n_chunks), dim3(64), 0, h->stream2, d_tasks.p, A, n_chunks,
                                   work.p, N, row_task.p, row_pos.p);
                F_TRY(hipEventRecord(h->ev_join, h->stream2));
                side_work = true;
            }
            {
                ScopedTimer tm(h, MORNA_T_TWO_MEANS, 4 * (int64_t)D * (TM_ITERS + 2) * A);
                const int nvq = (dpad / 4 + WAVE - 1) / WAVE;   // float4 per lane per row
                const unsigned wg = (unsigned)((A + 3) / 4);
#define TMW_LAUNCH(NVV)                                                                                              \
    hipLaunchKernelGGL(two_means_wave_kernel<NVV>, dim3(wg), dim3(256), 0, h->stream, h->X.p, h->rowinfo.p, N, dpad, \
                       work.p, d_tasks.p, A, seed, hp_level, d_ones.p)
                // Four waves per node (strips) while the level's nodes fit the chip at once (2 workgroups per CU): the
                // node's 200-step chain is then 2-3x shorter (C3: 0.32 / 0.28 ms instead of 0.7 ms at the two
                // shallowest levels).  Deeper levels are bound by the HBM gather of the rows (C3, 1600 nodes: 5.9 TB/s)
                // and four waves per node only add work there.  MORNA_TM_STRIP=0: one wave per node everywhere.
                static const bool tm_strip_on = !(getenv("MORNA_TM_STRIP") && atoi(getenv("MORNA_TM_STRIP")) == 0);
                const bool tm_strip = tm_strip_on && A <= 2 * h->n_cus;
                const bool strip_runs = tm_strip && (nvq == 12 || nvq == 8 || nvq == 4 || (nvq >= 16 && nvq <= 32 && nvq % 4 == 0));
                ScopedTimer tk(h, strip_runs ? MORNA_T_TM_STRIP : MORNA_T_TM_WAVE, 4 * (int64_t)D * (TM_ITERS + 2) * A);   // per kernel, beside the group
#define TMS_LAUNCH(NVV)                                                                                                 \
    hipLaunchKernelGGL((two_means_strip_kernel<NVV, TM_STRIP_DEPTH>), dim3((unsigned)A), dim3(256), 0, h->stream, h->X.p, \
                       h->rowinfo.p, N, dpad, work.p, d_tasks.p, seed, hp_level, d_ones.p)
                if (tm_strip && nvq == 12) TMS_LAUNCH(12);
                else if (tm_strip && nvq == 8) TMS_LAUNCH(8);
                else if (tm_strip && nvq == 4) TMS_LAUNCH(4);
                else if (tm_strip && nvq == 16) TMS_LAUNCH(16);   // rows too long for the one-wave register form: strips
                else if (tm_strip && nvq == 20) TMS_LAUNCH(20);   // (8 float4 per lane and array at D = 8192) instead of the
                else if (tm_strip && nvq == 24) TMS_LAUNCH(24);   // LDS form
                else if (tm_strip && nvq == 28) TMS_LAUNCH(28);
                else if (tm_strip && nvq == 32) TMS_LAUNCH(32);
                else if (nvq == 1) TMW_LAUNCH(1);
                else if (nvq == 2) TMW_LAUNCH(2);
                else if (nvq == 3) TMW_LAUNCH(3);
                else if (nvq == 4) TMW_LAUNCH(4);
                else if (nvq == 6) TMW_LAUNCH(6);
                else if (nvq == 8) TMW_LAUNCH(8);
                else if (nvq == 12) TMW_LAUNCH(12);
                else   // other row lengths (or too long for the register file): centroids in LDS, one workgroup per node
                    hipLaunchKernelGGL(two_means_kernel, dim3((unsigned)A), dim3(TM_THREADS), (size_t)dpad * 4 * 3, h->stream,
                                       h->X.p, h->norm2.p, N, dpad, work.p, d_tasks.p, seed, hp_level, d_ones.p);
#undef TMW_LAUNCH
#undef TMS_LAUNCH
            }
            if (!use_mm && !use_rw) {
                // chunk form: launch order = chunks sorted by first row id, one contiguous run per XCD
                if ((rc = d_info.alloc((size_t)n_chunks)) || (rc = d_sched.alloc((size_t)n_chunks))) { cleanup(); return rc; }
                F_TRY(hipMemsetAsync(d_hist.p, 0, (size_t)n_buckets * 4, h->stream));
                const unsigned cb = (unsigned)((n_chunks + 255) / 256);
                hipLaunchKernelGGL(sched_bucket_kernel, dim3(cb), dim3(256), 0, h->stream, d_tasks.p, A, n_chunks, work.p, N,
                                   n_buckets, d_hist.p, d_info.p);
                hipLaunchKernelGGL(sched_scan_kernel, dim3(1), dim3(1024), 0, h->stream, d_hist.p, n_buckets, d_cursor.p);
                hipLaunchKernelGGL(sched_scatter_kernel, dim3(cb), dim3(256), 0, h->stream, d_tasks.p, n_chunks, d_info.p,
                                   d_cursor.p, d_sched.p);
            }
            if (side_work) F_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
            if (use_mm) {
                ScopedTimer tm(h, MORNA_T_SPLIT, 4 * (int64_t)D * (rows + A));
                if ((rc = split_mm_level(h, d_tasks.p, A, S, hp_level, work.p, inv_p, seed, side.p, d_ones.p))) {
                    cleanup();
                    return rc;
                }
            } else if (use_rw) {
                // trees per group: 8 / 4 / 3 / 2 with 1 / 2 / 3-4 / 5-6 split nodes per tree (<= RW_SLOTS hyperplanes)
                const int G = max_per_tree == 1 ? 8 : max_per_tree == 2 ? 4 : max_per_tree <= 4 ? 3 : 2;
                const int n_windows = (int)((N + RW_ROWS - 1) / RW_ROWS), n_groups = (n_trees + G - 1) / G;
                ScopedTimer tm(h, MORNA_T_SPLIT, 4 * (int64_t)D * (rows + A));
                const unsigned grid = 8u * (unsigned)((n_windows + 7) / 8) * (unsigned)n_groups;
                const size_t lds = (size_t)G * max_per_tree * dpad * 4;
#define RW_LAUNCH_G(NVV, GTT)                                                                                            \
    do {                                                                                                                 \
        F_TRY(hipFuncSetAttribute((const void *)split_rw_kernel<NVV, GTT>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                  (int)lds));                                                                            \
        hipLaunchKernelGGL((split_rw_kernel<NVV, GTT>), dim3(grid), dim3(RW_THREADS), lds, h->stream, h->X.p, N, dpad,     \
                           d_tasks.p, d_tree_first.p, n_trees, G, row_task.p, row_pos.p, seed, hp_level, side.p, d_ones.p, \
                           n_windows, n_groups);                                                                         \
    } while (0)
#define RW_LAUNCH(NVV)                          \
    do {                                        \
        if (G == 8) RW_LAUNCH_G(NVV, 8);        \
        else if (G == 4) RW_LAUNCH_G(NVV, 4);   \
        else if (G == 3) RW_LAUNCH_G(NVV, 3);   \
        else RW_LAUNCH_G(NVV, 2);               \
    } while (0)
                if (nv == 1) RW_LAUNCH(1);
                else if (nv == 2) RW_LAUNCH(2);
                else if (nv == 3) RW_LAUNCH(3);
                else if (nv == 4) RW_LAUNCH(4);
                else if (nv == 6) RW_LAUNCH(6);
                else if (nv == 8) RW_LAUNCH(8);
                else RW_LAUNCH(12);
#undef RW_LAUNCH
#undef RW_LAUNCH_G
            } else {
                // algorithmic bytes: 4*D*sum|node| + 4*D*#split nodes
                ScopedTimer tm(h, MORNA_T_SPLIT, 4 * (int64_t)D * (rows + A));
                const unsigned grid = 8u * (unsigned)((n_chunks + 7) / 8);
                hipLaunchKernelGGL(split_kernel, dim3(grid), dim3(SP_THREADS), (size_t)dpad * 4, h->stream, h->X.p, N, dpad,
                                   work.p, d_tasks.p, d_sched.p, n_chunks, seed, hp_level, side.p, d_ones.p);
            }
            F_TRY(hipGetLastError());
            if ((rc = partition_and_fetch_counts(A, rows))) { cleanup(); return rc; }
            h->stats.split_attempts += A;
            h->stats.split_rows += rows;
            std::vector<int32_t> still;
            for (int32_t a = 0; a < A; a++) {
                const int32_t i = pending[(size_t)a];
                final_ones[(size_t)i] = h_ones[(size_t)a];
                const int64_t n1 = h_ones[(size_t)a], n0 = tasks[(size_t)a].count - n1;
                if (!(split_imbalance(n0, n1) < 0.95)) still.push_back(i);   // attempts 0, 1: sides_stand(); 2: sorted out below
            }
            pending.swap(still);
        }
        // "If we didn't find a hyperplane, just randomize sides as a last option"
        std::vector<int32_t> fb;
        for (int32_t i : pending) {
            const int64_t n1 = final_ones[(size_t)i], n0 = cur[(size_t)split_idx[(size_t)i]].count - n1;
            if (split_imbalance(n0, n1) > 0.99) fb.push_back(i);
        }
        if (!fb.empty()) {
            make_tasks(fb, 3, tasks);
            const int32_t A = (int32_t)tasks.size();
            if ((rc = upload_tasks(tasks))) { cleanup(); return rc; }
            hipLaunchKernelGGL(fallback_kernel, dim3((unsigned)A), dim3(256), 0, h->stream, d_tasks.p, N, dpad, seed,
                               side.p, d_ones.p, hp_level);
            F_TRY(hipGetLastError());
            int64_t fb_rows = 0;
            for (const SplitTask &t : tasks) fb_rows += t.count;
            if ((rc = partition_and_fetch_counts(A, fb_rows))) { cleanup(); return rc; }
            for (int32_t a = 0; a < A; a++) final_ones[(size_t)fb[(size_t)a]] = h_ones[(size_t)a];
            h->stats.fallback_nodes += A;
        }
        // children: ids base + 2*i + side for the i-th split node of the level.  Only the segments the NEXT level is cut
        // from are made now; the node table's entries (a few thousand vector appends) wait until that level's kernels
        // have been enqueued -- the device is idle while the host stands between a level's counts and the next launch
        nxt.clear();
        nxt.reserve((size_t)S * 2);
        late.parents.resize((size_t)S);
        late.ones.assign(final_ones.begin(), final_ones.end());
        late.first_id = next_node_id;
        late.first_hp = n_split_total;
        for (int32_t i = 0; i < S; i++) {
            const Seg &s = cur[(size_t)split_idx[(size_t)i]];
            const int32_t n1 = final_ones[(size_t)i], n0 = s.count - n1;
            const int32_t id0 = next_node_id + 2 * i, id1 = id0 + 1;
            late.parents[(size_t)i] = s;
            nxt.push_back(Seg{s.tree, s.level + 1, s.start, n0, id0});
            nxt.push_back(Seg{s.tree, s.level + 1, s.start + n0, n1, id1});
        }
        next_node_id += 2 * S;
        late.pending = true;
        n_split_total += S;
        cur.swap(nxt);
        level++;
    }

    write_late_tables();
    // consolidate hyperplanes and node tables in HBM
    h->n_split = n_split_total;
    h->n_nodes = (int64_t)ntree.size();
    if ((rc = h->hp.alloc((size_t)dpad))) return rc;   // never a null hyperplane table (N <= K: no split at all)
    if ((rc = h->node_rec.alloc(rec.size())) || (rc = h->node_tree.alloc(ntree.size())) || (rc = h->node_hp.alloc(nhp.size()))) {
        cleanup();
        return rc;
    }
    {
        // through page-locked staging: a copy from the pageable vectors is staged by the runtime, ~20 us of host time each,
        // and the device has nothing else to do just then
        const size_t b_rec = rec.size() * 4, b_tree = ntree.size() * 4, b_hp = nhp.size() * 4, b_leaf = leaves.size() * sizeof(LeafSeg);
        const size_t o_tree = b_rec /* 16 bytes per node */, o_hp = (o_tree + b_tree + 15) / 16 * 16, o_leaf = (o_hp + b_hp + 15) / 16 * 16,
                     need = o_leaf + b_leaf;
        static_assert(sizeof(LeafSeg) == 16, "leaves are moved in 16-byte words");
        if (need > h->host_tables_cap) {
            if (h->host_tables) {
                if (h->ev_tables_pending) F_TRY(hipEventSynchronize(h->ev_tables));   // a copy out of the old buffer may be in flight
                F_TRY(hipStreamSynchronize(h->stream));                               // (or a task list being pulled)
                (void)hipHostFree(h->host_tables);
            }
            h->host_tables = nullptr;
            h->host_tables_cap = 0;
            F_TRY(hipHostMalloc((void **)&h->host_tables, need * 2, hipHostMallocMapped));
            h->host_tables_cap = need * 2;
        }
        memcpy(h->host_tables, rec.data(), b_rec);
        memcpy(h->host_tables + o_tree, ntree.data(), b_tree);
        memcpy(h->host_tables + o_hp, nhp.data(), b_hp);
        memcpy(h->host_tables + o_leaf, leaves.data(), b_leaf);
        // (pulled by the device, as the task lists are; the tables are whole 16-byte words apart from their tails)
        void *dev_view = nullptr;
        F_TRY(hipHostGetDevicePointer(&dev_view, h->host_tables, 0));
        const uint8_t *dv = (const uint8_t *)dev_view;
        auto pull = [&](void *dst, const uint8_t *src, size_t bytes) -> hipError_t {
            const int32_t n16 = (int32_t)(bytes / 16);
            if (n16 > 0)
                hipLaunchKernelGGL(pull_tasks_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, h->stream, (const int4 *)src,
                                   (int4 *)dst, n16);
            if (bytes % 16)   // the tail, if any
                return hipMemcpyAsync((uint8_t *)dst + (size_t)n16 * 16, h->host_tables + (src - dv) + (size_t)n16 * 16, bytes % 16,
                                      hipMemcpyHostToDevice, h->stream);
            return hipGetLastError();
        };
        // the leaves first: their items go from the work images into the permutation the searches read
        ScratchRef<uint8_t> d_leaves(h->scratch[34]);
        if ((rc = d_leaves.alloc(std::max<size_t>(b_leaf, 16)))) { cleanup(); return rc; }
        F_TRY(pull(d_leaves.p, dv + o_leaf, b_leaf));
        if (!leaves.empty())
            hipLaunchKernelGGL(gather_leaves_kernel, dim3((unsigned)leaves.size()), dim3(256), 0, h->stream, (const LeafSeg *)d_leaves.p,
                               work.p, N, h->perm.p);
        F_TRY(hipGetLastError());
        F_TRY(pull(h->node_rec.p, dv, b_rec));
        F_TRY(pull(h->node_tree.p, dv + o_tree, b_tree));
        F_TRY(pull(h->node_hp.p, dv + o_hp, b_hp));
    }
    // No wait here: the last partition and these copies are ordered on the handle's stream in front of whatever the caller
    // does next with the handle (a search starts without the device draining first); the staging they read belongs to the
    // handle and is next written by another build, which waits for ev_tables first.  Blocking copies settle() as well.
    if (!h->ev_tables) F_TRY(hipEventCreateWithFlags(&h->ev_tables, hipEventDisableTiming));
    F_TRY(hipEventRecord(h->ev_tables, h->stream));
    h->ev_tables_pending = true;
    h->unsettled = true;
#undef F_TRY
    h->stats.n_nodes = h->n_nodes;
    h->stats.n_split = h->n_split;
    h->stats.n_leaves = h->n_nodes - h->n_split;
    h->built = true;
    return MORNA_OK;
}

}  // namespace morna
