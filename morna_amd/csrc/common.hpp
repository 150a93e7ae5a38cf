// common.hpp -- shared declarations of libmorna_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/morna_hip.h"

#define WAVE 64

namespace morna {

void set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            morna::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                             __LINE__);                                                       \
            return MORNA_E_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define MORNA_TRY(expr)            \
    do {                           \
        int _r = (expr);           \
        if (_r != MORNA_OK) return _r; \
    } while (0)

// ---- device buffer with explicit lifetime -----------------------------------
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;  // elements
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t count)
    {
        if (count <= n && p) return MORNA_OK;
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return MORNA_E_HIP;
        }
        n = count;
        return MORNA_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// typed view of one persistent scratch slot (same .p / .alloc surface as DevBuf)
template <typename T>
struct ScratchRef {
    DevBuf<uint8_t> &b;
    T *p = nullptr;
    explicit ScratchRef(DevBuf<uint8_t> &buf) : b(buf), p((T *)buf.p) {}
    int alloc(size_t count)
    {
        int rc = b.alloc(count * sizeof(T));
        p = (T *)b.p;
        return rc;
    }
};

// one split attempt of one node, as the forest kernels see it
struct SplitTask {
    int32_t tree, level, start, count;
    int32_t slot;      // hyperplane slot
    int32_t attempt;
    int32_t chunk0;    // first chunk index of this task in the split kernel grid
    int32_t pad;
};

// What two_means needs to know of a row besides its elements, made once per set of rows (row_norms_kernel) and
// fetched with one 16-byte load per step: the operations are the ones the step would otherwise perform itself.
struct alignas(16) RowInfo {
    float norm2;    // canonical dot(x, x)
    float norm;     // sqrtf(norm2); SIGN BIT SET when some x_i / norm may be a non-zero subnormal (centroid_step4)
    double rnorm;   // RN64(1 / (double)norm)
};
static_assert(sizeof(RowInfo) == 16, "RowInfo is one dwordx4");

// A node of the forest as the host driver tracks it (perm segment of one tree).
struct Seg {
    int32_t tree, level, start, count, node;
};

// Where a node's items are DURING a build: the work buffer holds two images of every tree's permutation, [tree][2][n_items],
// and a node of depth `level` has its items in image (level & 1) -- a partition reads one image and writes the node's two
// children into the other, so nothing is copied back (forest.hip; the leaves are gathered into h->perm at the end).
#define TASK_ITEMS_AT(t, n_items) (((int64_t)(t).tree * 2 + ((t).level & 1)) * (int64_t)(n_items) + (t).start)

struct Timer {
    double ms = 0;
    int64_t launches = 0, bytes = 0;
};

// one timed launch group whose events have been recorded but not yet read
struct PendingEv {
    hipEvent_t a, b;
    int which;
    int64_t bytes;
};

}  // namespace morna

struct morna_index {
    int32_t dim = 0;     // D (f in annoy)
    int32_t dpad = 0;    // row stride in floats: D rounded up to 256 (one 1-KiB load per wave and k-step)
    int32_t device = 0;
    int32_t n_cus = 256;               // compute units of the device (MI355X: 256)
    int32_t K = 0;       // leaf capacity D + 2
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;                 // side stream of the forest build (work that does not depend on two_means)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_tables = nullptr;    // behind the node-table copies of the last forest build
    bool ev_tables_pending = false;
    // right-side counts of a level, written by partition_kernel straight into page-locked host memory ((epoch << 32) |
    // count per task): the host polls them instead of an event hand-over + copy + stream wait per level
    unsigned long long *host_counts = nullptr;
    size_t host_counts_cap = 0;
    uint32_t count_epoch = 0;
    uint8_t *host_out = nullptr;       // page-locked staging of query results
    size_t host_out_cap = 0;
    uint8_t *host_small = nullptr;     // page-locked, device-visible: the answers of a small query batch, written by the kernel itself
    size_t host_small_cap = 0;
    uint8_t *host_q = nullptr;         // page-locked staging of a small batch's query vectors, padded to the row stride
    size_t host_q_cap = 0;
    uint8_t *host_tables = nullptr;    // page-locked staging of the node tables on their way to HBM (forest.hip)
    size_t host_tables_cap = 0;

    // host staging of add_item() rows until build()
    std::vector<float> host_rows;  // [host_n][dim]
    int64_t host_n = 0;
    bool host_dirty = false;

    // items in HBM
    int64_t n_items = 0;
    morna::DevBuf<float> X;      // [n_items][dpad], pad columns zero
    morna::DevBuf<float> norm2;  // [n_items] canonical dot(x, x)
    morna::DevBuf<morna::RowInfo> rowinfo;   // [n_items] norm2 again with what two_means derives from it
    bool norms_valid = false;
    bool unsettled = false;      // build_features() returned without waiting for its kernels: blocking copies must settle() first
    bool half_valid = false;     // scratch[19] / [20] hold the fp16 image of X, its norms and scales (splitmm.hip)
    bool ord_valid = false;      // the build under way has ordered its rows for the split contraction (splitmm.hip, scratch[29..33])

    // staged junction lines (CSR by line, file order)
    int64_t J = 0, nnz = 0, key_bytes_n = 0;
    morna::DevBuf<uint8_t> s_keys;
    morna::DevBuf<int64_t> s_key_off, s_row_ptr;
    morna::DevBuf<int32_t> s_ids, s_cov;
    morna::DevBuf<double> s_idf;
    bool staged = false;
    // optional order of the items in which the lines' sample lists ascend (morna_stage_item_order): rank of every item
    // and the item at every rank
    morna::DevBuf<int32_t> item_rank, item_at;
    int64_t order_n = 0;

    // forest
    int32_t n_trees = 0;
    uint32_t seed = 0;
    bool built = false;
    int64_t n_nodes = 0, n_split = 0;
    morna::DevBuf<int32_t> perm;        // [n_trees][n_items]
    morna::DevBuf<int32_t> node_rec;    // [n_nodes][4] {child0 | -1, child1 | -1, start, count}
    morna::DevBuf<int32_t> node_tree;   // [n_nodes]
    morna::DevBuf<int32_t> node_hp;     // [n_nodes] hyperplane slot or -1
    morna::DevBuf<float> hp;            // [n_split][dpad]
    morna_forest_stats stats = {};
    std::vector<int32_t> h_node_rec, h_node_tree, h_node_hp;  // host mirrors

    // query workspace (grown on demand)
    morna::DevBuf<uint8_t> ws;
    // exact search: query images, scan values and results of a batch; the candidates and their fp64 distances (knn.hip)
    morna::DevBuf<uint8_t> ex_ws;
    morna::DevBuf<int32_t> ex_cand;
    morna::DevBuf<double> ex_cdist;
    int32_t ex_cap = 0;                // candidates per query the last exact search needed room for
    // build scratch kept between calls (feature and forest builds reuse it instead of
    // hipMalloc / hipFree on every call); slots are named in features.hip / forest.hip
    morna::DevBuf<uint8_t> scratch[35];
    // [0] rows read by query kernels (hyperplane dots + candidates + 1 per query)
    morna::DevBuf<unsigned long long> d_stat;

    // row-sharded search (comm.hip): the RCCL communicator of this shard, every shard's first global id, message buffers
    void *comm = nullptr;              // ncclComm_t
    int32_t comm_rank = 0, comm_world = 1;
    bool comm_sizes_valid = false;     // comm_offsets describe the rows the ranks hold NOW (cleared whenever this handle's rows change)
    std::vector<int64_t> comm_offsets; // [world + 1]
    morna::DevBuf<uint8_t> cm_small, cm_msg, cm_q, cm_out;

    // timing
    bool timing = false;
    uint32_t timing_mask = 0;          // bit `which` set: that kernel group is bracketed by events
    morna::Timer timers[MORNA_T_COUNT];
    std::vector<morna::PendingEv> pending_ev;  // resolved by resolve_timers()
    std::vector<hipEvent_t> free_ev;
};

namespace morna {

// scope timer: records a HIP event pair on the handle's stream around a launch
// group; no host synchronisation here, the pairs are read by resolve_timers().
hipEvent_t take_event(morna_index *h);
void resolve_timers(morna_index *h);
struct ScopedTimer {
    morna_index *h;
    PendingEv pe;
    ScopedTimer(morna_index *h_, int which, int64_t bytes) : h(h_)
    {
        pe.which = which;
        pe.bytes = bytes;
        pe.a = pe.b = nullptr;
        if (!h->timing || !(h->timing_mask & (1u << which))) return;
        pe.a = take_event(h);
        pe.b = take_event(h);
        (void)hipEventRecord(pe.a, h->stream);
    }
    ~ScopedTimer()
    {
        if (!pe.a) return;
        (void)hipEventRecord(pe.b, h->stream);
        h->pending_ev.push_back(pe);
    }
};

// Before a blocking copy (which is not ordered against the handle's non-blocking stream) reads what the feature
// build wrote: wait for the stream once.
inline int settle(morna_index *h)
{
    if (h->unsettled) {
        hipError_t e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e));
            return MORNA_E_HIP;
        }
        h->unsettled = false;
    }
    return MORNA_OK;
}

// implemented in features.hip / forest.hip / knn.hip
int upload_host_rows(morna_index *h);
int compute_norms(morna_index *h);
int build_features(morna_index *h, int64_t n_items);
int hash_keys_device(morna_index *h, const uint8_t *key_bytes, const int64_t *key_off, int64_t J,
                     int32_t *hash_out, int32_t *col_out, int32_t *sign_out);
int build_forest(morna_index *h, int32_t n_trees, uint32_t seed);
// splitmm.hip: the split of a whole level on the matrix cores (fp16 filter, exact fp32 for what it cannot decide)
int split_mm_prepare_rows(morna_index *h, hipStream_t stream);   // stream: the handle's main or side stream
int split_mm_convert_rows(morna_index *h, const float *src, int64_t rows, _Float16 *dst, float *norm, float *err,
                          float *inv_scale, hipStream_t stream);
int split_mm_order_rows(morna_index *h, const uint8_t *side, const int32_t *inv_by_item, int32_t n_trees, hipStream_t stream,
                        const int32_t **rank_out, int32_t **inv_out);
int split_mm_level(morna_index *h, const SplitTask *d_tasks, int32_t n_tasks, int32_t n_slots, const float *hp_level,
                   const int32_t *perm, const int32_t *inv, uint32_t seed, uint8_t *side, int32_t *ones);
// packed_dev (device memory, or null): [nq][2k] int32 message of the row-sharded search -- ids + id_offset, distance bits
int query_batch(morna_index *h, const float *q_host, int64_t q_stride, const int32_t *items_host, int64_t nq, int32_t k,
                int32_t search_k, int32_t *ids_out, float *dist_out, int32_t *count_out, int32_t *packed_dev = nullptr,
                int64_t id_offset = 0);
int merge_topk_dev(morna_index *h, const int32_t *gathered_dev, int32_t world, int64_t nq, int32_t kk, int32_t k,
                   int32_t *ids_out, float *dist_out, int32_t *count_out);
int exact_search(morna_index *h, const double *q, int64_t nq, int32_t k, int32_t *ids_out,
                 double *dist_out, int32_t *count_out);
// the same for fp32 queries in device memory or for stored rows, answers to the host and / or packed for the sharded merge
int exact_search_any(morna_index *h, const double *q_host, const float *q_dev, const int32_t *items_host, int64_t nq, int32_t k,
                     int32_t *ids_out, double *dist_out, int32_t *count_out, uint8_t *msg_dev, int64_t id_offset);
size_t exact_msg_dist_offset(int64_t nq, int32_t k);
size_t exact_msg_bytes(int64_t nq, int32_t k);

}  // namespace morna
