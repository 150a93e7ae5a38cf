// api.hip -- the extern "C" surface declared in include/morna_hip.h.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "common.hpp"
#include "devutil.hpp"

namespace morna {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

hipEvent_t take_event(morna_index *h)
{
    if (!h->free_ev.empty()) {
        hipEvent_t e = h->free_ev.back();
        h->free_ev.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void resolve_timers(morna_index *h)
{
    if (h->pending_ev.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    unsigned long long rows = 0;
    bool have_q = false, have_mm = false;
    for (const PendingEv &pe : h->pending_ev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pe.a, pe.b) == hipSuccess) {
            h->timers[pe.which].ms += ms;
            h->timers[pe.which].launches += 1;
            h->timers[pe.which].bytes += pe.bytes;
        }
        if (pe.which == MORNA_T_QUERY) have_q = true;
        if (pe.which == MORNA_T_SPLIT_MM) have_mm = true;
        h->free_ev.push_back(pe.a);
        h->free_ev.push_back(pe.b);
    }
    h->pending_ev.clear();
    if (have_q && h->d_stat.p) {
        // rows read by the query kernels since the last reset -> 4*D bytes each
        if (hipMemcpy(&rows, h->d_stat.p, sizeof(rows), hipMemcpyDeviceToHost) == hipSuccess) {
            h->timers[MORNA_T_QUERY].bytes += (int64_t)rows * 4 * h->dim;
            (void)hipMemset(h->d_stat.p, 0, sizeof(rows));
        }
    }
    if (have_mm && h->d_stat.p) {
        // 256 x 256 x dpad products the split contraction launched through its per-tile task lists (splitmm.hip)
        unsigned long long tiles = 0;
        if (hipMemcpy(&tiles, h->d_stat.p + 1, sizeof(tiles), hipMemcpyDeviceToHost) == hipSuccess) {
            h->timers[MORNA_T_SPLIT_MM].bytes += (int64_t)tiles * 2 * 256 * 256 * h->dpad;
            (void)hipMemset(h->d_stat.p + 1, 0, sizeof(tiles));
        }
    }
}

}  // namespace morna

using namespace morna;

// rows ids[b] of the padded matrix -> packed [n][dim]
__global__ void gather_rows_kernel(const float *__restrict__ X, const int32_t *__restrict__ ids, int32_t dim,
                                   int32_t dpad, float *__restrict__ out)
{
    const float *src = X + (int64_t)ids[blockIdx.x] * dpad;
    float *dst = out + (int64_t)blockIdx.x * dim;
    for (int z = threadIdx.x; z < dim; z += blockDim.x) dst[z] = src[z];
}

#define CHECK_H(h)                            \
    if (!(h)) {                                 \
        set_error("null index handle");         \
        return MORNA_E_INVALID;                 \
    }

extern "C" {

const char *morna_last_error(void) { return g_err; }

int32_t morna_hash32(const uint8_t *key, int64_t len) { return (int32_t)murmur3_32(key, len, 0u); }

int morna_index_create(int32_t dim, int32_t device, morna_index **out)
{
    if (!out || dim <= 0) {
        set_error("AnnoyIndex: dimension must be positive (got %d)", dim);
        return MORNA_E_INVALID;
    }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (%s): libmorna_hip has no CPU path", hipGetErrorString(e));
        return MORNA_E_HIP;
    }
    if (device < 0 || device >= ndev) {
        set_error("device %d out of range (%d visible)", device, ndev);
        return MORNA_E_INVALID;
    }
    HIP_TRY(hipSetDevice(device));
    morna_index *h = new (std::nothrow) morna_index();
    if (!h) {
        set_error("out of host memory");
        return MORNA_E_INVALID;
    }
    h->dim = dim;
    h->dpad = (dim + 255) / 256 * 256;   // whole 1-KiB wave loads: every lane active in every k-step
    h->K = dim + 2;
    h->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) h->n_cus = cus;
    }
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate: %s", hipGetErrorString(e));
        delete h;
        return MORNA_E_HIP;
    }
    int rc = h->d_stat.alloc(4);
    if (rc == MORNA_OK && hipMemset(h->d_stat.p, 0, 4 * sizeof(unsigned long long)) != hipSuccess) rc = MORNA_E_HIP;
    if (rc == MORNA_OK && (hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
                           hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess)) {
        set_error("hipStreamCreate / hipEventCreate failed");
        rc = MORNA_E_HIP;
    }
    if (rc != MORNA_OK) {
        if (h->ev_tables) (void)hipEventDestroy(h->ev_tables);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        if (h->ev_join) (void)hipEventDestroy(h->ev_join);
        if (h->stream2) (void)hipStreamDestroy(h->stream2);
        (void)hipStreamDestroy(h->stream);
        delete h;
        return rc;
    }
    *out = h;
    return MORNA_OK;
}

int morna_device_count(int32_t *count_out)
{
    if (!count_out) return MORNA_E_INVALID;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    *count_out = e == hipSuccess ? n : 0;
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s): libmorna_hip has no CPU path", hipGetErrorString(e));
        return MORNA_E_HIP;
    }
    return MORNA_OK;
}

int morna_index_destroy(morna_index *h)
{
    if (!h) return MORNA_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)morna_comm_destroy(h);
    for (const PendingEv &pe : h->pending_ev) {
        (void)hipEventDestroy(pe.a);
        (void)hipEventDestroy(pe.b);
    }
    for (hipEvent_t e : h->free_ev) (void)hipEventDestroy(e);
    if (h->stream2) (void)hipStreamSynchronize(h->stream2);
    if (h->ev_tables) (void)hipEventDestroy(h->ev_tables);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->host_counts) (void)hipHostFree(h->host_counts);
    if (h->host_out) (void)hipHostFree(h->host_out);
    if (h->host_small) (void)hipHostFree(h->host_small);
    if (h->host_q) (void)hipHostFree(h->host_q);
    if (h->host_tables) (void)hipHostFree(h->host_tables);
    hipStream_t s = h->stream, s2 = h->stream2;
    delete h;   // DevBuf destructors free HBM
    if (s2) (void)hipStreamDestroy(s2);
    if (s) (void)hipStreamDestroy(s);
    return MORNA_OK;
}

// ---- items -------------------------------------------------------------------

static int ensure_host_rows(morna_index *h, int64_t n)
{
    if (h->host_n == 0 && h->n_items > 0 && !h->host_dirty) {
        // items live only in HBM (feature build / load): bring them back before mixing
        h->host_rows.assign((size_t)h->n_items * h->dim, 0.f);
        MORNA_TRY(settle(h));
        HIP_TRY(hipMemcpy2D(h->host_rows.data(), (size_t)h->dim * 4, h->X.p, (size_t)h->dpad * 4, (size_t)h->dim * 4,
                            (size_t)h->n_items, hipMemcpyDeviceToHost));
        h->host_n = h->n_items;
    }
    if (n > h->host_n) {
        h->host_rows.resize((size_t)n * h->dim, 0.f);   // annoy zero-fills skipped ids
        h->host_n = n;
    }
    return MORNA_OK;
}

int morna_add_item(morna_index *h, int32_t id, const double *v)
{
    CHECK_H(h);
    MORNA_TRY(settle(h));
    if (id < 0 || !v) {
        set_error("add_item: item id must be non-negative");
        return MORNA_E_RANGE;
    }
    if (h->built) {
        set_error("You can't add an item to a built index");
        return MORNA_E_STATE;
    }
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(ensure_host_rows(h, (int64_t)id + 1));
    float *dst = h->host_rows.data() + (size_t)id * h->dim;
    for (int32_t z = 0; z < h->dim; z++) dst[z] = (float)v[z];   // Python float -> C float
    h->host_dirty = true;
    return MORNA_OK;
}

int morna_add_items_f32(morna_index *h, int32_t first_id, const float *rows, int64_t n)
{
    CHECK_H(h);
    MORNA_TRY(settle(h));
    if (first_id < 0 || n < 0 || (n > 0 && !rows)) {
        set_error("add_items: bad arguments");
        return MORNA_E_INVALID;
    }
    if (h->built) {
        set_error("You can't add an item to a built index");
        return MORNA_E_STATE;
    }
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(ensure_host_rows(h, (int64_t)first_id + n));
    if (n > 0) memcpy(h->host_rows.data() + (size_t)first_id * h->dim, rows, (size_t)n * h->dim * sizeof(float));
    h->host_dirty = true;
    return MORNA_OK;
}

int morna_stage_junctions(morna_index *h, const uint8_t *key_bytes, const int64_t *key_off, int64_t J,
                          const int64_t *row_ptr, const int32_t *item_ids, const int32_t *cov, const double *idf)
{
    CHECK_H(h);
    MORNA_TRY(settle(h));
    if (J < 0 || (J > 0 && (!key_bytes || !key_off || !row_ptr || !item_ids || !cov || !idf))) {
        set_error("stage_junctions: null input");
        return MORNA_E_INVALID;
    }
    HIP_TRY(hipSetDevice(h->device));
    const int64_t nbytes = J ? key_off[J] : 0, nnz = J ? row_ptr[J] : 0;
    if (J && (key_off[0] != 0 || row_ptr[0] != 0)) {
        set_error("stage_junctions: offsets must start at 0");
        return MORNA_E_INVALID;
    }
    for (int64_t j = 0; j < J; j++)
        if (key_off[j + 1] < key_off[j] || row_ptr[j + 1] < row_ptr[j]) {
            set_error("stage_junctions: offsets must be non-decreasing (line %lld)", (long long)j);
            return MORNA_E_INVALID;
        }
    MORNA_TRY(h->s_keys.alloc((size_t)nbytes));
    MORNA_TRY(h->s_key_off.alloc((size_t)J + 1));
    MORNA_TRY(h->s_row_ptr.alloc((size_t)J + 1));
    MORNA_TRY(h->s_ids.alloc((size_t)nnz));
    MORNA_TRY(h->s_cov.alloc((size_t)nnz));
    MORNA_TRY(h->s_idf.alloc((size_t)J));
    if (J > 0) {
        HIP_TRY(hipMemcpyAsync(h->s_keys.p, key_bytes, (size_t)nbytes, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_key_off.p, key_off, (size_t)(J + 1) * 8, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_row_ptr.p, row_ptr, (size_t)(J + 1) * 8, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_ids.p, item_ids, (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_cov.p, cov, (size_t)nnz * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipMemcpyAsync(h->s_idf.p, idf, (size_t)J * 8, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->J = J;
    h->nnz = nnz;
    h->key_bytes_n = nbytes;
    h->staged = true;
    return MORNA_OK;
}

int morna_stage_item_order(morna_index *h, const int64_t *order_key, int64_t n_items)
{
    CHECK_H(h);
    MORNA_TRY(settle(h));
    if (n_items < 0 || n_items >= INT32_MAX || (n_items > 0 && !order_key)) {
        set_error("stage_item_order: bad arguments");
        return MORNA_E_INVALID;
    }
    HIP_TRY(hipSetDevice(h->device));
    h->order_n = 0;
    if (n_items == 0) return MORNA_OK;
    try {
        std::vector<int32_t> at((size_t)n_items), rank((size_t)n_items);
        for (int64_t i = 0; i < n_items; i++) at[(size_t)i] = (int32_t)i;
        std::stable_sort(at.begin(), at.end(), [&](int32_t a, int32_t b) { return order_key[a] < order_key[b]; });
        for (int64_t r = 0; r < n_items; r++) rank[(size_t)at[(size_t)r]] = (int32_t)r;
        MORNA_TRY(h->item_rank.alloc((size_t)n_items));
        MORNA_TRY(h->item_at.alloc((size_t)n_items));
        HIP_TRY(hipMemcpy(h->item_rank.p, rank.data(), (size_t)n_items * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->item_at.p, at.data(), (size_t)n_items * 4, hipMemcpyHostToDevice));
    } catch (const std::exception &e) {
        set_error("stage_item_order: %s", e.what());
        return MORNA_E_INVALID;
    }
    h->order_n = n_items;
    return MORNA_OK;
}

int morna_unstage_junctions(morna_index *h)
{
    CHECK_H(h);
    MORNA_TRY(settle(h));
    h->s_keys.release(); h->s_key_off.release(); h->s_row_ptr.release();
    h->s_ids.release(); h->s_cov.release(); h->s_idf.release();
    for (int i = 0; i < 8; i++) h->scratch[i].release();   // feature-build scratch (fp64 column image ...)
    h->scratch[24].release();                               // tile extents of the lines
    h->scratch[25].release();                               // positions of the entries' items
    h->item_rank.release(); h->item_at.release();
    h->order_n = 0;
    h->staged = false;
    h->J = h->nnz = h->key_bytes_n = 0;
    return MORNA_OK;
}

int morna_build_features(morna_index *h, int64_t n_items)
{
    CHECK_H(h);
    HIP_TRY(hipSetDevice(h->device));
    return build_features(h, n_items);
}

int morna_hash_keys(morna_index *h, const uint8_t *key_bytes, const int64_t *key_off, int64_t J, int32_t *hash_out,
                    int32_t *col_out, int32_t *sign_out)
{
    CHECK_H(h);
    MORNA_TRY(settle(h));
    HIP_TRY(hipSetDevice(h->device));
    return hash_keys_device(h, key_bytes, key_off, J, hash_out, col_out, sign_out);
}

int64_t morna_get_n_items(const morna_index *h)
{
    if (!h) return 0;
    return h->host_dirty ? h->host_n : h->n_items;
}

int morna_get_item_vector(morna_index *h, int32_t id, float *out)
{
    CHECK_H(h);
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(upload_host_rows(h));
    if (id < 0 || id >= h->n_items) {
        set_error("Item index %d out of range [0, %lld)", id, (long long)h->n_items);
        return MORNA_E_RANGE;
    }
    MORNA_TRY(settle(h));
    HIP_TRY(hipMemcpy(out, h->X.p + (size_t)id * h->dpad, (size_t)h->dim * 4, hipMemcpyDeviceToHost));
    return MORNA_OK;
}

static int get_item_vectors_impl(morna_index *h, const int32_t *ids, int64_t n, float *out, bool wait);

int morna_get_item_vectors(morna_index *h, const int32_t *ids, int64_t n, float *out)
{
    CHECK_H(h);
    return get_item_vectors_impl(h, ids, n, out, true);
}

int morna_get_item_vectors_dev(morna_index *h, const int32_t *ids, int64_t n, float *out_dev)
{
    CHECK_H(h);
    if (n > 0 && !out_dev) {
        set_error("get_item_vectors_dev: null buffer");
        return MORNA_E_INVALID;
    }
    return get_item_vectors_impl(h, ids, n, out_dev, false);
}

int morna_get_stream(morna_index *h, void **stream_out)
{
    CHECK_H(h);
    if (!stream_out) {
        set_error("get_stream: null pointer");
        return MORNA_E_INVALID;
    }
    *stream_out = (void *)h->stream;
    return MORNA_OK;
}

static int get_item_vectors_impl(morna_index *h, const int32_t *ids, int64_t n, float *out, bool wait)
{
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(upload_host_rows(h));
    for (int64_t i = 0; i < n; i++)
        if (ids[i] < 0 || ids[i] >= h->n_items) {
            set_error("Item index %d out of range [0, %lld)", ids[i], (long long)h->n_items);
            return MORNA_E_RANGE;
        }
    if (n <= 0) return MORNA_OK;
    // one gather launch into a packed [n][dim] image -- `out` itself when it is device memory handed over in stream
    // order, else a staging image and ONE copy to `out` (host or device memory): a memcpy per row costs microseconds
    // of launch each
    const size_t idb = ((size_t)n * 4 + 255) / 256 * 256;
    MORNA_TRY(h->ws.alloc(idb + (wait ? (size_t)n * h->dim * 4 : 0)));
    int32_t *d_ids = (int32_t *)h->ws.p;
    float *d_rows = wait ? (float *)(h->ws.p + idb) : out;
    HIP_TRY(hipMemcpyAsync(d_ids, ids, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n), dim3(256), 0, h->stream, h->X.p, d_ids, h->dim, h->dpad, d_rows);
    HIP_TRY(hipGetLastError());
    if (wait) {
        HIP_TRY(hipMemcpyAsync(out, d_rows, (size_t)n * h->dim * 4, hipMemcpyDefault, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    } else {
        h->unsettled = true;   // returns with work queued
    }
    return MORNA_OK;
}

int morna_get_items(morna_index *h, float *rows_out)
{
    CHECK_H(h);
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(upload_host_rows(h));
    MORNA_TRY(settle(h));
    if (h->n_items > 0)
        HIP_TRY(hipMemcpy2D(rows_out, (size_t)h->dim * 4, h->X.p, (size_t)h->dpad * 4, (size_t)h->dim * 4,
                            (size_t)h->n_items, hipMemcpyDeviceToHost));
    return MORNA_OK;
}

int morna_get_norms2(morna_index *h, float *out)
{
    CHECK_H(h);
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(upload_host_rows(h));
    MORNA_TRY(settle(h));
    if (h->n_items > 0) HIP_TRY(hipMemcpy(out, h->norm2.p, (size_t)h->n_items * 4, hipMemcpyDeviceToHost));
    return MORNA_OK;
}

// ---- forest ------------------------------------------------------------------

int morna_build(morna_index *h, int32_t n_trees, uint32_t seed)
{
    CHECK_H(h);
    HIP_TRY(hipSetDevice(h->device));
    if (h->built) {
        set_error("You can't build a built index");
        return MORNA_E_STATE;
    }
    return build_forest(h, n_trees, seed);
}

int32_t morna_get_n_trees(const morna_index *h) { return h && h->built ? h->n_trees : 0; }

int morna_get_forest_stats(const morna_index *h, morna_forest_stats *out)
{
    CHECK_H(h);
    if (!out) return MORNA_E_INVALID;
    *out = h->stats;
    return MORNA_OK;
}

int morna_get_forest(morna_index *h, int32_t *node_rec, int32_t *perm, float *hyperplanes, int32_t *hp_node)
{
    CHECK_H(h);
    if (!h->built) {
        set_error("index has not been built");
        return MORNA_E_STATE;
    }
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(settle(h));
    const int64_t n = h->n_nodes;
    if (node_rec)
        for (int64_t i = 0; i < n; i++) {
            const int32_t *r = &h->h_node_rec[(size_t)i * 4];
            int32_t *o = node_rec + i * 6;
            o[0] = r[0] < 0 ? 1 : 0;
            o[1] = h->h_node_tree[(size_t)i];
            o[2] = r[2];
            o[3] = r[3];
            o[4] = r[0];
            o[5] = r[1];
        }
    if (perm)
        HIP_TRY(hipMemcpy(perm, h->perm.p, (size_t)h->n_trees * h->n_items * 4, hipMemcpyDeviceToHost));
    if (hyperplanes && h->n_split > 0)
        HIP_TRY(hipMemcpy2D(hyperplanes, (size_t)h->dim * 4, h->hp.p, (size_t)h->dpad * 4, (size_t)h->dim * 4,
                            (size_t)h->n_split, hipMemcpyDeviceToHost));
    if (hp_node)
        for (int64_t i = 0; i < n; i++)
            if (h->h_node_hp[(size_t)i] >= 0) hp_node[h->h_node_hp[(size_t)i]] = (int32_t)i;
    return MORNA_OK;
}

// ---- search ------------------------------------------------------------------

int morna_get_nns_by_vector(morna_index *h, const float *q, int64_t nq, int32_t k, int32_t search_k, int32_t *ids_out,
                            float *dist_out, int32_t *count_out)
{
    CHECK_H(h);
    if (!q || !ids_out) {
        set_error("get_nns_by_vector: null buffer");
        return MORNA_E_INVALID;
    }
    HIP_TRY(hipSetDevice(h->device));
    return query_batch(h, q, 0, nullptr, nq, k, search_k, ids_out, dist_out, count_out);
}

int morna_get_nns_by_item(morna_index *h, const int32_t *items, int64_t nq, int32_t k, int32_t search_k,
                          int32_t *ids_out, float *dist_out, int32_t *count_out)
{
    CHECK_H(h);
    if (!items || !ids_out) {
        set_error("get_nns_by_item: null buffer");
        return MORNA_E_INVALID;
    }
    HIP_TRY(hipSetDevice(h->device));
    return query_batch(h, nullptr, 0, items, nq, k, search_k, ids_out, dist_out, count_out);
}

int morna_get_nns_by_vector_packed(morna_index *h, const float *q, int64_t nq, int32_t k, int32_t search_k, int64_t id_offset,
                                   int32_t *packed_dev)
{
    CHECK_H(h);
    if (!q || !packed_dev) {
        set_error("get_nns_by_vector_packed: null buffer");
        return MORNA_E_INVALID;
    }
    if (id_offset < 0 || id_offset + h->n_items > (int64_t)INT32_MAX) {
        set_error("get_nns_by_vector_packed: global ids past 2^31 do not fit the packed message");
        return MORNA_E_RANGE;
    }
    HIP_TRY(hipSetDevice(h->device));
    h->unsettled = true;   // returns with work queued: blocking copies on the null stream must settle() first
    return query_batch(h, q, 0, nullptr, nq, k, search_k, nullptr, nullptr, nullptr, packed_dev, id_offset);
}

int morna_merge_topk_packed(morna_index *h, const int32_t *gathered_dev, int32_t world, int64_t nq, int32_t kk, int32_t k,
                            int32_t *ids_out, float *dist_out, int32_t *count_out)
{
    CHECK_H(h);
    HIP_TRY(hipSetDevice(h->device));
    return merge_topk_dev(h, gathered_dev, world, nq, kk, k, ids_out, dist_out, count_out);
}

int morna_exact_search(morna_index *h, const double *q, int64_t nq, int32_t k, int32_t *ids_out, double *dist_out,
                       int32_t *count_out)
{
    CHECK_H(h);
    if (!q || !ids_out) {
        set_error("exact_search: null buffer");
        return MORNA_E_INVALID;
    }
    HIP_TRY(hipSetDevice(h->device));
    return exact_search(h, q, nq, k, ids_out, dist_out, count_out);
}

// ---- persistence ---------------------------------------------------------------
// One little-endian blob: header, matrix rows [n][dim], norms, forest tables.
// It stands in for annoy's mmap file (basename.annoy.mor); byte compatibility
// with annoy's own format is not a goal.

static const char MAGIC[8] = {'M', 'O', 'R', 'N', 'A', 'H', 'I', '1'};

struct FileHeader {
    char magic[8];
    int32_t dim, n_trees;
    int64_t n_items, n_nodes, n_split;
    uint32_t seed, built;
    morna_forest_stats stats;
};

int morna_save(morna_index *h, const char *path)
{
    CHECK_H(h);
    HIP_TRY(hipSetDevice(h->device));
    MORNA_TRY(upload_host_rows(h));
    FILE *f = fopen(path, "wb");
    if (!f) {
        set_error("Unable to open %s for writing", path);
        return MORNA_E_IO;
    }
    FileHeader hd;
    memset(&hd, 0, sizeof(hd));
    memcpy(hd.magic, MAGIC, 8);
    hd.dim = h->dim; hd.n_trees = h->built ? h->n_trees : 0;
    hd.n_items = h->n_items; hd.n_nodes = h->built ? h->n_nodes : 0; hd.n_split = h->built ? h->n_split : 0;
    hd.seed = h->seed; hd.built = h->built ? 1 : 0;
    hd.stats = h->stats;
    bool ok = fwrite(&hd, sizeof(hd), 1, f) == 1;
    std::vector<float> buf;
    if (settle(h) != MORNA_OK) ok = false;
    if (ok && h->n_items > 0) {
        buf.resize((size_t)h->n_items * h->dim);
        if (hipMemcpy2D(buf.data(), (size_t)h->dim * 4, h->X.p, (size_t)h->dpad * 4, (size_t)h->dim * 4,
                        (size_t)h->n_items, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
        ok = ok && fwrite(buf.data(), 4, buf.size(), f) == buf.size();
    }
    if (ok && h->built) {
        std::vector<int32_t> perm((size_t)h->n_trees * h->n_items);
        if (hipMemcpy(perm.data(), h->perm.p, perm.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
        ok = ok && fwrite(perm.data(), 4, perm.size(), f) == perm.size();
        ok = ok && fwrite(h->h_node_rec.data(), 4, h->h_node_rec.size(), f) == h->h_node_rec.size();
        ok = ok && fwrite(h->h_node_tree.data(), 4, h->h_node_tree.size(), f) == h->h_node_tree.size();
        ok = ok && fwrite(h->h_node_hp.data(), 4, h->h_node_hp.size(), f) == h->h_node_hp.size();
        if (ok && h->n_split > 0) {
            buf.resize((size_t)h->n_split * h->dim);
            if (hipMemcpy2D(buf.data(), (size_t)h->dim * 4, h->hp.p, (size_t)h->dpad * 4, (size_t)h->dim * 4,
                            (size_t)h->n_split, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
            ok = ok && fwrite(buf.data(), 4, buf.size(), f) == buf.size();
        }
    }
    if (fclose(f) != 0) ok = false;
    if (!ok) {
        set_error("Unable to write %s", path);
        return MORNA_E_IO;
    }
    return MORNA_OK;
}

// A file is data from outside: everything the kernels index with is checked here, so that a truncated or
// damaged blob is MORNA_E_IO and never an out-of-bounds access on the device or an exception across the C ABI.
// (The .freq.mor / .map.mor files beside it are Python pickles, as in the reference, morna.py:449-455, 545-549:
// an index file set must come from a trusted source.)
static bool forest_tables_ok(const FileHeader &hd, const std::vector<int32_t> &perm, const std::vector<int32_t> &rec,
                             const std::vector<int32_t> &tree, const std::vector<int32_t> &hp)
{
    const int64_t n = hd.n_nodes, N = hd.n_items, T = hd.n_trees;
    std::vector<uint8_t> refs((size_t)n, 0), slot_used((size_t)std::max<int64_t>(hd.n_split, 1), 0);
    for (int64_t i = 0; i < n; i++) {
        const int32_t c0 = rec[(size_t)i * 4], c1 = rec[(size_t)i * 4 + 1], start = rec[(size_t)i * 4 + 2], count = rec[(size_t)i * 4 + 3];
        if (tree[(size_t)i] < 0 || tree[(size_t)i] >= T) return false;
        if (start < 0 || count < 0 || (int64_t)start + count > N) return false;
        if (i < T && (tree[(size_t)i] != i || start != 0 || count != N)) return false;   // roots are nodes 0 .. T-1
        // a node splits exactly when it holds more than a leaf may (annoy's _make_tree; the traversal kernel reads the
        // children and the hyperplane of every root with n_items > K without looking at the record first)
        const bool splits = c0 >= 0 && c1 >= 0;
        if (splits != ((int64_t)count > (int64_t)hd.dim + 2)) return false;
        if (!splits) {
            if (c0 != -1 || c1 != -1 || hp[(size_t)i] != -1) return false;
            continue;
        }
        // children come after their parent (breadth-first ids): no cycles; each node has exactly one parent
        if (c0 <= i || c1 <= i || c0 >= n || c1 >= n || c0 == c1) return false;
        if (refs[(size_t)c0]++ || refs[(size_t)c1]++) return false;
        if (tree[(size_t)c0] != tree[(size_t)i] || tree[(size_t)c1] != tree[(size_t)i]) return false;
        // the children partition the parent's segment of the tree's permutation, left child first, neither empty
        const int32_t s0 = rec[(size_t)c0 * 4 + 2], n0 = rec[(size_t)c0 * 4 + 3], s1 = rec[(size_t)c1 * 4 + 2], n1 = rec[(size_t)c1 * 4 + 3];
        if (s0 != start || n0 <= 0 || n1 <= 0 || (int64_t)n0 + n1 != count || s1 != start + n0) return false;
        if (hp[(size_t)i] < 0 || hp[(size_t)i] >= hd.n_split || slot_used[(size_t)hp[(size_t)i]]++) return false;
    }
    for (int64_t i = 0; i < n; i++)
        if ((i < T) != (refs[(size_t)i] == 0)) return false;
    for (int32_t v : perm)
        if (v < 0 || v >= N) return false;
    // every tree's permutation lists every item once (a leaf's segment is handed to the refine kernel as candidate ids)
    std::vector<uint8_t> seen((size_t)std::max<int64_t>(N, 1));
    for (int64_t t = 0; t < T; t++) {
        std::fill(seen.begin(), seen.end(), 0);
        for (int64_t i = 0; i < N; i++)
            if (seen[(size_t)perm[(size_t)(t * N + i)]]++) return false;
    }
    return true;
}

static int load_impl(morna_index *h, const char *path, FILE *f)
{
    FileHeader hd;
    if (fread(&hd, sizeof(hd), 1, f) != 1 || memcmp(hd.magic, MAGIC, 8) != 0) {
        set_error("%s is not a morna-hip index file", path);
        return MORNA_E_IO;
    }
    if (hd.dim != h->dim) {
        set_error("Index size is not a multiple of vector size: file has dimension %d, index was created with %d", hd.dim, h->dim);
        return MORNA_E_IO;
    }
    // the header against itself and against the size of the file
    const bool built = hd.built != 0;
    bool sane = hd.n_items >= 0 && hd.n_items < INT32_MAX && hd.n_trees >= 0 && hd.n_trees <= (1 << 24) && hd.n_nodes >= 0 &&
                hd.n_split >= 0 && hd.n_nodes < INT32_MAX;
    if (sane && built) sane = hd.n_trees > 0 && hd.n_items > 0 && hd.n_nodes == (int64_t)hd.n_trees + 2 * hd.n_split;
    if (sane && !built) sane = hd.n_nodes == 0 && hd.n_split == 0;
    if (sane) {
        long long expect = (long long)sizeof(hd) + 4ll * hd.n_items * hd.dim;
        if (built) expect += 4ll * hd.n_trees * hd.n_items + 4ll * 6 * hd.n_nodes + 4ll * hd.n_split * hd.dim;
        const long at = ftell(f);
        sane = fseek(f, 0, SEEK_END) == 0 && (long long)ftell(f) == expect && fseek(f, at, SEEK_SET) == 0;
    }
    if (!sane) {
        set_error("%s is truncated or damaged (header and file size disagree)", path);
        return MORNA_E_IO;
    }
    bool ok = true;
    std::vector<float> buf;
    h->host_rows.clear(); h->host_n = 0; h->host_dirty = false; h->built = false;
    h->half_valid = false;
    if (h->n_items != hd.n_items) h->comm_sizes_valid = false;
    h->n_items = hd.n_items;
    MORNA_TRY(h->X.alloc((size_t)std::max<int64_t>(hd.n_items, 1) * h->dpad));
    (void)hipMemset(h->X.p, 0, (size_t)std::max<int64_t>(hd.n_items, 1) * h->dpad * 4);
    if (hd.n_items > 0) {
        buf.resize((size_t)hd.n_items * h->dim);
        ok = fread(buf.data(), 4, buf.size(), f) == buf.size();
        if (ok && hipMemcpy2D(h->X.p, (size_t)h->dpad * 4, buf.data(), (size_t)h->dim * 4, (size_t)h->dim * 4,
                              (size_t)hd.n_items, hipMemcpyHostToDevice) != hipSuccess) ok = false;
    }
    if (ok) MORNA_TRY(compute_norms(h));
    if (ok && built) {
        h->n_trees = hd.n_trees; h->n_nodes = hd.n_nodes; h->n_split = hd.n_split; h->seed = hd.seed; h->stats = hd.stats;
        std::vector<int32_t> perm((size_t)hd.n_trees * hd.n_items);
        h->h_node_rec.resize((size_t)hd.n_nodes * 4);
        h->h_node_tree.resize((size_t)hd.n_nodes);
        h->h_node_hp.resize((size_t)hd.n_nodes);
        ok = fread(perm.data(), 4, perm.size(), f) == perm.size();
        ok = ok && fread(h->h_node_rec.data(), 4, h->h_node_rec.size(), f) == h->h_node_rec.size();
        ok = ok && fread(h->h_node_tree.data(), 4, h->h_node_tree.size(), f) == h->h_node_tree.size();
        ok = ok && fread(h->h_node_hp.data(), 4, h->h_node_hp.size(), f) == h->h_node_hp.size();
        if (ok && !forest_tables_ok(hd, perm, h->h_node_rec, h->h_node_tree, h->h_node_hp)) {
            set_error("%s is damaged: its forest tables are inconsistent", path);
            return MORNA_E_IO;
        }
        if (ok) {
            MORNA_TRY(h->perm.alloc(perm.size()));
            MORNA_TRY(h->node_rec.alloc(h->h_node_rec.size()));
            MORNA_TRY(h->node_tree.alloc(h->h_node_tree.size()));
            MORNA_TRY(h->node_hp.alloc(h->h_node_hp.size()));
            MORNA_TRY(h->hp.alloc((size_t)std::max<int64_t>(hd.n_split, 1) * h->dpad));
            ok = hipMemcpy(h->perm.p, perm.data(), perm.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(h->node_rec.p, h->h_node_rec.data(), h->h_node_rec.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(h->node_tree.p, h->h_node_tree.data(), h->h_node_tree.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(h->node_hp.p, h->h_node_hp.data(), h->h_node_hp.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
            (void)hipMemset(h->hp.p, 0, (size_t)std::max<int64_t>(hd.n_split, 1) * h->dpad * 4);
        }
        if (ok && hd.n_split > 0) {
            buf.resize((size_t)hd.n_split * h->dim);
            ok = fread(buf.data(), 4, buf.size(), f) == buf.size();
            if (ok && hipMemcpy2D(h->hp.p, (size_t)h->dpad * 4, buf.data(), (size_t)h->dim * 4, (size_t)h->dim * 4,
                                  (size_t)hd.n_split, hipMemcpyHostToDevice) != hipSuccess) ok = false;
        }
        if (ok) h->built = true;
    }
    if (!ok) {
        set_error("Unable to read %s (truncated or HIP copy failed)", path);
        return MORNA_E_IO;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MORNA_OK;
}

int morna_load(morna_index *h, const char *path)
{
    CHECK_H(h);
    MORNA_TRY(settle(h));
    HIP_TRY(hipSetDevice(h->device));
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) {
        set_error("Unable to open %s", path ? path : "(null)");
        return MORNA_E_IO;
    }
    int rc;
    try {
        rc = load_impl(h, path, f);
    } catch (const std::exception &e) {   // bad_alloc / length_error must not cross the C ABI
        set_error("Unable to read %s: %s", path, e.what());
        rc = MORNA_E_IO;
    }
    fclose(f);
    if (rc != MORNA_OK) {   // a failed load leaves an empty, unbuilt index rather than half of one
        h->built = false;
        h->n_items = 0;
        h->comm_sizes_valid = false;
        h->norms_valid = false;
    }
    return rc;
}

// ---- measurement -----------------------------------------------------------------

int morna_timer_enable(morna_index *h, int32_t on)
{
    CHECK_H(h);
    if (!on) resolve_timers(h);
    h->timing = on != 0;
    // 1: every group; otherwise bit (which + 1) selects group `which` (an event pair costs the stream a few
    // microseconds of idle, so a benchmark brackets only the group it prices)
    h->timing_mask = on == 1 ? 0xffffffffu : ((uint32_t)on >> 1);
    return MORNA_OK;
}

int morna_timer_reset(morna_index *h)
{
    CHECK_H(h);
    HIP_TRY(hipSetDevice(h->device));
    resolve_timers(h);
    for (int i = 0; i < MORNA_T_COUNT; i++) h->timers[i] = Timer();
    if (h->d_stat.p) HIP_TRY(hipMemset(h->d_stat.p, 0, 2 * sizeof(unsigned long long)));
    return MORNA_OK;
}

int morna_timer_read(morna_index *h, int32_t which, double *ms, int64_t *launches, int64_t *bytes)
{
    CHECK_H(h);
    if (which < 0 || which >= MORNA_T_COUNT) {
        set_error("timer_read: no such timer %d", which);
        return MORNA_E_INVALID;
    }
    HIP_TRY(hipSetDevice(h->device));
    resolve_timers(h);
    if (ms) *ms = h->timers[which].ms;
    if (launches) *launches = h->timers[which].launches;
    if (bytes) *bytes = h->timers[which].bytes;
    return MORNA_OK;
}

int morna_merge_topk(const int64_t *ids, const float *dist, int32_t world, int64_t nq, int32_t kk, int32_t k,
                     int64_t *ids_out, float *dist_out, int32_t *count_out)
{
    if (!ids || !dist || !ids_out || !dist_out || !count_out || world <= 0 || nq < 0 || kk <= 0 || k <= 0) {
        set_error("merge_topk: invalid argument");
        return MORNA_E_INVALID;
    }
    // every list is sorted: a k-way merge by (distance, id); NaN after every number, as in the kernels' keys
    auto key = [](float d) -> uint32_t {
        if (d == 0.f) d = 0.f;
        uint32_t u;
        memcpy(&u, &d, 4);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    };
    std::vector<int32_t> head((size_t)world);
    for (int64_t q = 0; q < nq; q++) {
        std::fill(head.begin(), head.end(), 0);
        int32_t n = 0;
        for (; n < k; n++) {
            int best = -1;
            uint32_t bk = 0;
            int64_t bi = 0;
            for (int w = 0; w < world; w++) {
                if (head[(size_t)w] >= kk) continue;
                const size_t at = ((size_t)w * (size_t)nq + (size_t)q) * (size_t)kk + (size_t)head[(size_t)w];
                const int64_t id = ids[at];
                if (id < 0) continue;   // the rest of this shard's list is empty
                const uint32_t dk = key(dist[at]);
                if (best < 0 || dk < bk || (dk == bk && id < bi)) {
                    best = w;
                    bk = dk;
                    bi = id;
                }
            }
            if (best < 0) break;
            const size_t at = ((size_t)best * (size_t)nq + (size_t)q) * (size_t)kk + (size_t)head[(size_t)best];
            ids_out[q * k + n] = ids[at];
            dist_out[q * k + n] = dist[at];
            head[(size_t)best]++;
        }
        count_out[q] = n;
        for (; n < k; n++) {
            ids_out[q * k + n] = -1;
            dist_out[q * k + n] = INFINITY;
        }
    }
    return MORNA_OK;
}

int morna_synchronize(morna_index *h)
{
    CHECK_H(h);
    MORNA_TRY(settle(h));
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MORNA_OK;
}

}  // extern "C"
