// comm.hip -- the row-sharded search inside the library: an RCCL communicator owned by the handle.
//
// The reference has no distributed path; SURVEY.md 8(e) defines this one: rank g of G holds the rows with global ids
// [off[g], off[g+1]) and its own forest, a query is answered by every shard, and the ONE exchange on the data path is an
// all-gather of the per-shard top-k -- Q * k * 8 bytes per rank for the approximate search (Q * (12 k + 4) for the exact
// one, whose distances are fp64), latency-bound over xGMI -- after which every rank merges to the same answer.  By-item
// queries first all-gather the query rows (Q * D * 4 bytes, HBM -> xGMI -> HBM).  Everything is enqueued on the handle's
// stream: per-shard search, ncclAllGather, merge kernel; the host waits once, for the merged result.
//
// RCCL is looked up at run time (dlopen): a process that never shards never loads it, and one that has PyTorch loaded
// shares PyTorch's copy instead of bringing a second one.  No RCCL -> morna_comm_init fails (MORNA_E_STATE); nothing
// falls back to the host.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "devutil.hpp"

namespace morna {

namespace {

struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl R;
    static bool tried = false;
    if (tried) return R;
    tried = true;
    // a copy the process already has (PyTorch's is loaded under this name) before one of our own
    R.so = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!R.so) R.so = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!R.so) R.so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!R.so) R.so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!R.so) return R;
    R.GetUniqueId = (decltype(R.GetUniqueId))dlsym(R.so, "ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.so, "ncclCommInitRank");
    R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.so, "ncclCommDestroy");
    R.AllGather = (decltype(R.AllGather))dlsym(R.so, "ncclAllGather");
    R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.so, "ncclGetErrorString");
    R.ok = R.GetUniqueId && R.CommInitRank && R.CommDestroy && R.AllGather && R.GetErrorString;
    return R;
}

#define NCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t _r = (expr);                                                                   \
        if (_r != ncclSuccess) {                                                                    \
            set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__); \
            return MORNA_E_HIP;                                                                     \
        }                                                                                           \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int need_comm(morna_index *h)
{
    if (!h->comm) {
        set_error("the handle has no communicator: call morna_comm_init first");
        return MORNA_E_STATE;
    }
    return MORNA_OK;
}

// every rank's row count -> the global id offsets.  One 8-byte all-gather and one host wait: at the first sharded call
// behind morna_comm_init and again whenever THIS handle's row count has changed since (a rebuild of the same number of rows
// -- bench.py's steps -- exchanges nothing and waits for nothing).  The exchange is a collective, so the ranks must change
// their row counts together; a caller that resizes some shards only calls morna_comm_info(.., offsets) on every rank,
// which always exchanges.
int sync_sizes(morna_index *h)
{
    MORNA_TRY(upload_host_rows(h));
    if (h->comm_sizes_valid) return MORNA_OK;
    const int world = h->comm_world;
    MORNA_TRY(h->cm_small.alloc((size_t)(world + 1) * 8));
    int64_t *d = (int64_t *)h->cm_small.p;
    const int64_t mine = h->n_items;
    HIP_TRY(hipMemcpyAsync(d + world, &mine, 8, hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(rccl().AllGather(d + world, d, 1, ncclInt64, (ncclComm_t)h->comm, h->stream));
    std::vector<int64_t> sizes((size_t)world);
    HIP_TRY(hipMemcpyAsync(sizes.data(), d, (size_t)world * 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->comm_offsets.assign((size_t)world + 1, 0);
    for (int g = 0; g < world; g++) h->comm_offsets[(size_t)g + 1] = h->comm_offsets[(size_t)g] + sizes[(size_t)g];
    if (h->comm_offsets.back() >= (int64_t)INT32_MAX) {
        set_error("sharded search: %lld items in total, global ids must stay below 2^31", (long long)h->comm_offsets.back());
        return MORNA_E_RANGE;
    }
    h->comm_sizes_valid = true;
    return MORNA_OK;
}

// result block in HBM -> the caller's arrays, through the handle's page-locked staging; waits for the stream
int fetch_results(morna_index *h, const uint8_t *d_block, size_t s_ids, size_t s_dist, size_t dist_elt, int64_t nq, int32_t k,
                  int32_t *ids_out, void *dist_out, int32_t *count_out)
{
    const size_t s_cnt = align_up((size_t)nq * 4, 256), out_bytes = s_ids + s_dist + s_cnt;
    if (out_bytes > h->host_out_cap) {
        if (h->host_out) (void)hipHostFree(h->host_out);
        h->host_out = nullptr;
        h->host_out_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&h->host_out, out_bytes * 2, hipHostMallocDefault));
        h->host_out_cap = out_bytes * 2;
    }
    HIP_TRY(hipMemcpyAsync(h->host_out, d_block, out_bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->unsettled = false;
    memcpy(ids_out, h->host_out, (size_t)nq * k * 4);
    if (dist_out) memcpy(dist_out, h->host_out + s_ids, (size_t)nq * k * dist_elt);
    if (count_out) memcpy(count_out, h->host_out + s_ids + s_dist, (size_t)nq * 4);
    return MORNA_OK;
}

}  // namespace

// ---- merge of the all-gathered exact answers ---------------------------------------------------------------------
// gathered[world] messages (exact_msg_bytes(nq, kk) each, knn.hip): what exact_search_nn's bisect_left scan over ALL rows
// would keep (morna.py:705-712) -- ascending distance, among equal distances the HIGHER global id first.  A shard whose
// count is -1 ("the reference raises ValueError for this query": a row all but parallel to it) fails the query for the
// whole matrix.  NaN distances (only such queries have them) go behind the numbers, higher id first; empty slots last.
// One wave per query; a candidate's place is the number of candidates in front of it.
__global__ __launch_bounds__(64) void merge_exact_kernel(const uint8_t *__restrict__ gathered, size_t msg_bytes, size_t dist_off,
                                                         int32_t world, int64_t nq, int32_t kk, int32_t k,
                                                         int32_t *__restrict__ ids_out, double *__restrict__ dist_out,
                                                         int32_t *__restrict__ count_out)
{
    const int64_t q = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = world * kk;
    __shared__ int s_failed, s_real;
    if (tid == 0) s_failed = s_real = 0;
    __syncthreads();
    for (int w = tid; w < world; w += 64)
        if (((const int32_t *)(gathered + (size_t)w * msg_bytes))[nq * kk + q] < 0) s_failed = 1;
    auto id_of = [&](int c) { return ((const int32_t *)(gathered + (size_t)(c / kk) * msg_bytes))[q * kk + c % kk]; };
    auto d_of = [&](int c) { return ((const double *)(gathered + (size_t)(c / kk) * msg_bytes + dist_off))[q * kk + c % kk]; };
    for (int c = tid; c < n; c += 64) {
        const int32_t id = id_of(c);
        if (id < 0) continue;
        atomicAdd(&s_real, 1);
        const double d = d_of(c);
        const bool dnan = d != d;
        int rank = 0;
        for (int u = 0; u < n; u++) {
            const int32_t iu = id_of(u);
            if (iu < 0 || u == c) continue;
            const double du = d_of(u);
            const bool unan = du != du;
            bool before;
            if (dnan) before = !unan || iu > id;
            else before = !unan && (du < d || (du == d && iu > id));
            rank += before ? 1 : 0;
        }
        if (rank < k) {
            ids_out[q * k + rank] = id;
            dist_out[q * k + rank] = d;
        }
    }
    __syncthreads();
    const int kout = s_real < k ? s_real : k;
    for (int r = kout + tid; r < k; r += 64) {
        ids_out[q * k + r] = -1;
        dist_out[q * k + r] = INFINITY;
    }
    if (tid == 0) count_out[q] = s_failed ? -1 : kout;
}

// rows [n_each[g]] of rank g out of the padded all-gather image [world][n_max][dim] -> [sum n_each][dim]
__global__ void compact_rows_kernel(const float *__restrict__ allq, int64_t n_max, int32_t dim, const int64_t *__restrict__ row_src,
                                    float *__restrict__ out)
{
    const float *src = allq + row_src[blockIdx.x] * dim;
    float *dst = out + (int64_t)blockIdx.x * dim;
    for (int z = threadIdx.x; z < dim; z += blockDim.x) dst[z] = src[z];
}

static int merge_exact_dev(morna_index *h, const uint8_t *gathered_dev, int32_t world, int64_t nq, int32_t kk, int32_t k,
                           int32_t *ids_out, double *dist_out, int32_t *count_out)
{
    if (world <= 0 || nq < 0 || kk <= 0 || k <= 0 || !gathered_dev || !ids_out) {
        set_error("merge_exact: invalid argument");
        return MORNA_E_INVALID;
    }
    if (nq == 0) return MORNA_OK;
    const size_t s_ids = align_up((size_t)nq * k * 4, 256), s_dist = align_up((size_t)nq * k * 8, 256), s_cnt = align_up((size_t)nq * 4, 256);
    MORNA_TRY(h->cm_out.alloc(s_ids + s_dist + s_cnt));
    uint8_t *p = h->cm_out.p;
    hipLaunchKernelGGL(merge_exact_kernel, dim3((unsigned)nq), dim3(64), 0, h->stream, gathered_dev, exact_msg_bytes(nq, kk),
                       exact_msg_dist_offset(nq, kk), world, nq, kk, k, (int32_t *)p, (double *)(p + s_ids), (int32_t *)(p + s_ids + s_dist));
    HIP_TRY(hipGetLastError());
    return fetch_results(h, p, s_ids, s_dist, 8, nq, k, ids_out, dist_out, count_out);
}

// this rank's stored rows `local_items` as its share of the queries -> every rank's, all-gathered: [sum n_each][dim] fp32
// in cm_q (device).  n_each[world] or null (then the counts are exchanged first: one more latency-bound collective).
static int gather_query_rows(morna_index *h, const int32_t *local_items, int64_t n_local, const int64_t *n_each, int64_t *nq_out,
                             const float **q_dev_out)
{
    const int world = h->comm_world, rank = h->comm_rank;
    Rccl &R = rccl();
    if (n_local < 0 || (n_local > 0 && !local_items)) {
        set_error("sharded by-item search: null item list");
        return MORNA_E_INVALID;
    }
    for (int64_t i = 0; i < n_local; i++)
        if (local_items[i] < 0 || local_items[i] >= h->n_items) {
            set_error("Item index %d out of range [0, %lld) of this shard", local_items[i], (long long)h->n_items);
            return MORNA_E_RANGE;
        }
    std::vector<int64_t> each((size_t)world);
    if (n_each) {
        each.assign(n_each, n_each + world);
        if (each[(size_t)rank] != n_local) {
            set_error("sharded by-item search: n_each[%d] = %lld but this rank hands over %lld items", rank, (long long)each[(size_t)rank],
                      (long long)n_local);
            return MORNA_E_INVALID;
        }
    } else {
        MORNA_TRY(h->cm_small.alloc((size_t)(world + 1) * 8));
        int64_t *d = (int64_t *)h->cm_small.p;
        HIP_TRY(hipMemcpyAsync(d + world, &n_local, 8, hipMemcpyHostToDevice, h->stream));
        NCCL_TRY(R.AllGather(d + world, d, 1, ncclInt64, (ncclComm_t)h->comm, h->stream));
        HIP_TRY(hipMemcpyAsync(each.data(), d, (size_t)world * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    int64_t n_max = 0, total = 0;
    for (int64_t v : each) {
        if (v < 0) {
            set_error("sharded by-item search: negative query count");
            return MORNA_E_INVALID;
        }
        n_max = std::max(n_max, v);
        total += v;
    }
    *nq_out = total;
    *q_dev_out = nullptr;
    if (total == 0) return MORNA_OK;
    const int32_t dim = h->dim;
    // every rank's rows padded to the largest count (ncclAllGather moves equal counts), then compacted -- ALWAYS, also when
    // the counts are equal: one code path, the one the 1-rank tests run, for 12 MB of extra copy per 1000 queries
    const size_t s_mine = align_up((size_t)n_max * dim * 4, 256), s_all = align_up((size_t)world * n_max * dim * 4, 256),
                 s_cmp = align_up((size_t)total * dim * 4, 256), s_src = align_up((size_t)total * 8, 256);
    MORNA_TRY(h->cm_q.alloc(s_mine + s_all + s_cmp + s_src));
    float *mine = (float *)h->cm_q.p, *allq = (float *)(h->cm_q.p + s_mine);
    float *cmp = (float *)(h->cm_q.p + s_mine + s_all);
    int64_t *d_src = (int64_t *)(h->cm_q.p + s_mine + s_all + s_cmp);
    if (n_local < n_max) HIP_TRY(hipMemsetAsync(mine, 0, s_mine, h->stream));   // (the padding is never read back; kept defined)
    if (n_local > 0) MORNA_TRY(morna_get_item_vectors_dev(h, local_items, n_local, mine));   // enqueued, no host wait
    // row of the gathered image that holds query j (rank g's i-th query sits at g * n_max + i), through page-locked memory
    const size_t src_bytes = (size_t)total * 8;
    if (src_bytes > h->host_q_cap) {
        if (h->host_q) (void)hipHostFree(h->host_q);
        h->host_q = nullptr;
        h->host_q_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&h->host_q, src_bytes * 2, hipHostMallocDefault));
        h->host_q_cap = src_bytes * 2;
    }
    // (every user of the staging block ends with a host wait -- fetch_results here, the small-batch path of query_batch --:
    // no earlier copy can still be reading it)
    int64_t *src = (int64_t *)h->host_q;
    int64_t at = 0;
    for (int g = 0; g < world; g++)
        for (int64_t i = 0; i < each[(size_t)g]; i++) src[at++] = (int64_t)g * n_max + i;
    HIP_TRY(hipMemcpyAsync(d_src, src, src_bytes, hipMemcpyHostToDevice, h->stream));
    NCCL_TRY(R.AllGather(mine, allq, (size_t)n_max * dim, ncclFloat, (ncclComm_t)h->comm, h->stream));
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)total), dim3(256), 0, h->stream, allq, n_max, dim, d_src, cmp);
    HIP_TRY(hipGetLastError());
    h->unsettled = true;
    *q_dev_out = cmp;
    return MORNA_OK;
}

static int approx_sharded(morna_index *h, const float *q, int64_t nq, int32_t k, int32_t search_k, int32_t *ids_out, float *dist_out,
                          int32_t *count_out)
{
    if (k <= 0 || k > 255 || nq < 0 || !ids_out) {
        set_error("sharded search: need 0 < k <= 255 and an id buffer");
        return MORNA_E_INVALID;
    }
    if (nq == 0) return MORNA_OK;
    const int world = h->comm_world;
    const size_t s_msg = align_up((size_t)nq * 2 * k * 4, 256);
    MORNA_TRY(h->cm_msg.alloc(s_msg * (size_t)(world + 1)));
    int32_t *packed = (int32_t *)h->cm_msg.p, *gathered = (int32_t *)(h->cm_msg.p + s_msg);
    MORNA_TRY(morna_get_nns_by_vector_packed(h, q, nq, k, search_k, h->comm_offsets[(size_t)h->comm_rank], packed));
    NCCL_TRY(rccl().AllGather(packed, gathered, (size_t)nq * 2 * k, ncclInt32, (ncclComm_t)h->comm, h->stream));
    return morna_merge_topk_packed(h, gathered, world, nq, k, k, ids_out, dist_out, count_out);
}

static int exact_sharded(morna_index *h, const double *q_host, const float *q_dev, int64_t nq, int32_t k, int32_t *ids_out,
                         double *dist_out, int32_t *count_out)
{
    if (k <= 0 || nq < 0 || !ids_out) {
        set_error("sharded exact search: k must be positive");
        return MORNA_E_INVALID;
    }
    if (nq == 0) return MORNA_OK;
    const int world = h->comm_world;
    const size_t msg = exact_msg_bytes(nq, k), s_msg = align_up(msg, 256);
    MORNA_TRY(h->cm_msg.alloc(s_msg + msg * (size_t)world));
    uint8_t *mine = h->cm_msg.p, *gathered = h->cm_msg.p + s_msg;
    MORNA_TRY(exact_search_any(h, q_host, q_dev, nullptr, nq, k, nullptr, nullptr, nullptr, mine, h->comm_offsets[(size_t)h->comm_rank]));
    NCCL_TRY(rccl().AllGather(mine, gathered, msg, ncclChar, (ncclComm_t)h->comm, h->stream));
    return merge_exact_dev(h, gathered, world, nq, k, k, ids_out, dist_out, count_out);
}

}  // namespace morna

using namespace morna;

#define CHECK_H(h)                      \
    if (!(h)) {                         \
        set_error("null index handle"); \
        return MORNA_E_INVALID;         \
    }

extern "C" {

int morna_comm_unique_id(uint8_t *id_out)
{
    if (!id_out) {
        set_error("comm_unique_id: null buffer");
        return MORNA_E_INVALID;
    }
    Rccl &R = rccl();
    if (!R.ok) {
        set_error("RCCL (librccl.so) could not be loaded: the row-sharded search has no other transport");
        return MORNA_E_STATE;
    }
    static_assert(sizeof(ncclUniqueId) == MORNA_COMM_ID_BYTES, "MORNA_COMM_ID_BYTES is sizeof(ncclUniqueId)");
    ncclUniqueId id;
    NCCL_TRY(R.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return MORNA_OK;
}

int morna_comm_init(morna_index *h, const uint8_t *id, int32_t rank, int32_t world)
{
    CHECK_H(h);
    if (!id || world <= 0 || world > 64 || rank < 0 || rank >= world) {
        set_error("comm_init: need an id and 0 <= rank < world <= 64");
        return MORNA_E_INVALID;
    }
    if (h->comm) {
        set_error("comm_init: the handle already owns a communicator");
        return MORNA_E_STATE;
    }
    Rccl &R = rccl();
    if (!R.ok) {
        set_error("RCCL (librccl.so) could not be loaded: the row-sharded search has no other transport");
        return MORNA_E_STATE;
    }
    HIP_TRY(hipSetDevice(h->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    NCCL_TRY(R.CommInitRank(&comm, world, uid, rank));
    h->comm = comm;
    h->comm_rank = rank;
    h->comm_world = world;
    h->comm_sizes_valid = false;
    return MORNA_OK;
}

int morna_comm_destroy(morna_index *h)
{
    CHECK_H(h);
    if (!h->comm) return MORNA_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    ncclResult_t r = rccl().CommDestroy((ncclComm_t)h->comm);
    h->comm = nullptr;
    h->comm_world = 1;
    h->comm_rank = 0;
    h->comm_sizes_valid = false;
    if (r != ncclSuccess) {
        set_error("ncclCommDestroy failed: %s", rccl().GetErrorString(r));
        return MORNA_E_HIP;
    }
    return MORNA_OK;
}

int morna_comm_info(morna_index *h, int32_t *rank, int32_t *world, int64_t *offsets)
{
    CHECK_H(h);
    MORNA_TRY(need_comm(h));
    HIP_TRY(hipSetDevice(h->device));
    if (rank) *rank = h->comm_rank;
    if (world) *world = h->comm_world;
    if (offsets) {
        h->comm_sizes_valid = false;   // asked for: exchanged (every rank calls this together)
        MORNA_TRY(sync_sizes(h));
        memcpy(offsets, h->comm_offsets.data(), (size_t)(h->comm_world + 1) * 8);
    }
    return MORNA_OK;
}

int morna_get_nns_by_vector_sharded(morna_index *h, const float *q, int64_t nq, int32_t k, int32_t search_k, int32_t *ids_out,
                                    float *dist_out, int32_t *count_out)
{
    CHECK_H(h);
    MORNA_TRY(need_comm(h));
    HIP_TRY(hipSetDevice(h->device));
    if (nq > 0 && !q) {
        set_error("sharded search: null query buffer");
        return MORNA_E_INVALID;
    }
    MORNA_TRY(sync_sizes(h));
    return approx_sharded(h, q, nq, k, search_k, ids_out, dist_out, count_out);
}

int morna_get_nns_by_item_sharded(morna_index *h, const int32_t *local_items, int64_t n_local, const int64_t *n_each, int32_t k,
                                  int32_t search_k, int32_t *ids_out, float *dist_out, int32_t *count_out)
{
    CHECK_H(h);
    MORNA_TRY(need_comm(h));
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(sync_sizes(h));
    int64_t nq = 0;
    const float *q_dev = nullptr;
    MORNA_TRY(gather_query_rows(h, local_items, n_local, n_each, &nq, &q_dev));
    return approx_sharded(h, q_dev, nq, k, search_k, ids_out, dist_out, count_out);
}

int morna_exact_search_sharded(morna_index *h, const double *q, int64_t nq, int32_t k, int32_t *ids_out, double *dist_out,
                               int32_t *count_out)
{
    CHECK_H(h);
    MORNA_TRY(need_comm(h));
    HIP_TRY(hipSetDevice(h->device));
    if (nq > 0 && !q) {
        set_error("sharded exact search: null query buffer");
        return MORNA_E_INVALID;
    }
    MORNA_TRY(sync_sizes(h));
    return exact_sharded(h, q, nullptr, nq, k, ids_out, dist_out, count_out);
}

int morna_exact_search_by_item_sharded(morna_index *h, const int32_t *local_items, int64_t n_local, const int64_t *n_each, int32_t k,
                                       int32_t *ids_out, double *dist_out, int32_t *count_out)
{
    CHECK_H(h);
    MORNA_TRY(need_comm(h));
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(sync_sizes(h));
    int64_t nq = 0;
    const float *q_dev = nullptr;
    MORNA_TRY(gather_query_rows(h, local_items, n_local, n_each, &nq, &q_dev));
    return exact_sharded(h, nullptr, q_dev, nq, k, ids_out, dist_out, count_out);
}

int morna_exact_search_by_item(morna_index *h, const int32_t *items, int64_t nq, int32_t k, int32_t *ids_out, double *dist_out,
                               int32_t *count_out)
{
    CHECK_H(h);
    if (!items || !ids_out) {
        set_error("exact_search_by_item: null buffer");
        return MORNA_E_INVALID;
    }
    HIP_TRY(hipSetDevice(h->device));
    return exact_search_any(h, nullptr, nullptr, items, nq, k, ids_out, dist_out, count_out, nullptr, 0);
}

int64_t morna_exact_packed_bytes(int64_t nq, int32_t k) { return nq < 0 || k <= 0 ? 0 : (int64_t)exact_msg_bytes(nq, k); }

int morna_exact_search_packed(morna_index *h, const double *q, const float *q_dev, const int32_t *items, int64_t nq, int32_t k,
                              int64_t id_offset, uint8_t *packed_dev)
{
    CHECK_H(h);
    if (((q != nullptr) + (q_dev != nullptr) + (items != nullptr)) != 1 || !packed_dev) {
        set_error("exact_search_packed: exactly one of q / q_dev / items, and a message buffer");
        return MORNA_E_INVALID;
    }
    if (id_offset < 0 || id_offset + h->n_items > (int64_t)INT32_MAX) {
        set_error("exact_search_packed: global ids past 2^31 do not fit the message");
        return MORNA_E_RANGE;
    }
    HIP_TRY(hipSetDevice(h->device));
    return exact_search_any(h, q, q_dev, items, nq, k, nullptr, nullptr, nullptr, packed_dev, id_offset);
}

int morna_merge_exact_packed(morna_index *h, const uint8_t *gathered_dev, int32_t world, int64_t nq, int32_t kk, int32_t k,
                             int32_t *ids_out, double *dist_out, int32_t *count_out)
{
    CHECK_H(h);
    HIP_TRY(hipSetDevice(h->device));
    return merge_exact_dev(h, gathered_dev, world, nq, kk, k, ids_out, dist_out, count_out);
}

}  // extern "C"
