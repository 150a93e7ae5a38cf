/*
 * morna_hip.h -- C ABI of libmorna_hip.so, the MI355X (gfx950) implementation of
 * morna's index-build + nearest-neighbour search hot path.
 *
 * This is the drop-in boundary: the entry points are what a binding for the
 * reference's two native dependencies on this path would call.  Each one cites
 * the reference interface it replaces (file:line under commanderson/morna).
 *
 *   annoy.AnnoyIndex (C++ extension, used angular)  morna.py:26, 166, 543
 *   mmh3.hash        (C extension)                  morna.py:20, 369, 591, 625
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is caller-owned HOST memory
 *     unless the name says _dev; the library copies during the call and never
 *     retains a host pointer.
 *   - every function returns 0 on success or a negative MORNA_E_* code; the
 *     message for the calling thread is available from morna_last_error().
 *   - one host thread per handle; the handle owns its HIP stream.
 *   - ids are dense internal ids 0..n_items-1 (morna.py:378-382).
 *   - distances are annoy "angular": sqrt(max(2 - 2 cos, 0)).
 */
#ifndef MORNA_HIP_H
#define MORNA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MORNA_OK            0
#define MORNA_E_INVALID    -1   /* bad argument              -> ValueError  */
#define MORNA_E_HIP        -2   /* HIP runtime failure       -> RuntimeError */
#define MORNA_E_STATE      -3   /* call order (not built...) -> RuntimeError */
#define MORNA_E_RANGE      -4   /* id out of range           -> IndexError  */
#define MORNA_E_IO         -5   /* save / load               -> IOError     */
#define MORNA_E_EMPTY      -6   /* no items (morna.py:399-403) -> ValueError */

typedef struct morna_index morna_index;

/* AnnoyIndex(dim, metric='angular')                          morna.py:166, 543, 1171 */
int morna_index_create(int32_t dim, int32_t device, morna_index **out);
int morna_index_destroy(morna_index *h);
/* HIP devices the library can create indexes on (0 and MORNA_E_HIP when there is none: the library has no CPU path) */
int morna_device_count(int32_t *count_out);
const char *morna_last_error(void);

/* mmh3.hash(key) -- host mirror of the device hash            morna.py:369, 591, 625 */
int32_t morna_hash32(const uint8_t *key, int64_t len);

/* ---- items --------------------------------------------------------------- */

/* AnnoyIndex.add_item(i, vector): fp64 -> fp32 happens here    morna.py:406, 423 */
int morna_add_item(morna_index *h, int32_t id, const double *v);
/* bulk form: rows[n][dim] fp32 become items first_id .. first_id+n-1 */
int morna_add_items_f32(morna_index *h, int32_t first_id, const float *rows, int64_t n);

/*
 * Fused replacement for the add_junction loop + add_item hand-off
 * (morna.py:344-388, 405-424): J kept junction lines in FILE ORDER.
 *   key_bytes/key_off[J+1]  the "chrom start end" strings        morna.py:849
 *   row_ptr[J+1]            extent of each line's lists
 *   item_ids[nnz]           INTERNAL id of each sample           morna.py:377-382
 *   cov[nnz]                coverages                            morna.py:852
 *   idf[J]                  log(sample_count / cumulative freq)  morna.py:372-374
 *                           (host libm, so that it is bit-identical to Python)
 * stage = host -> HBM copy only; build_features = the kernels (hash, signed
 * column, fp64 accumulation in file order, fp64 -> fp32, row norms).
 * build_features returns once its kernels are enqueued on the handle's stream: what follows on the handle is
 * ordered behind them (entry points that copy with the host wait first); morna_synchronize() waits explicitly.
 */
int morna_stage_junctions(morna_index *h, const uint8_t *key_bytes, const int64_t *key_off, int64_t J,
                          const int64_t *row_ptr, const int32_t *item_ids, const int32_t *cov,
                          const double *idf);
int morna_build_features(morna_index *h, int64_t n_items);
/*
 * Optional, before build_features: an ORDER of the items in which the sample lists of the lines ascend --
 * order_key[i] for internal id i, e.g. the external sample id (an intropolis line lists its samples in ascending
 * order, morna.py:848-853, while internal ids are first-seen, morna.py:377-382).  Performance only: each entry of the
 * staged lists is then read once instead of once per sample tile; lines that do not ascend in the order, or no
 * order at all, give the same matrix.  Dropped by morna_unstage_junctions; ignored unless n_items matches build_features'.
 */
int morna_stage_item_order(morna_index *h, const int64_t *order_key, int64_t n_items);
int morna_unstage_junctions(morna_index *h);
/* device hash of staged or given keys, for tests: hash/col/sign per key */
int morna_hash_keys(morna_index *h, const uint8_t *key_bytes, const int64_t *key_off, int64_t J,
                    int32_t *hash_out, int32_t *col_out, int32_t *sign_out);

/*
 * Native host pre-pass of `morna index` (no GPU work): go_index's line loop,
 * count_samples and the host half of add_junction                morna.py:841-861, 789-822, 357-382
 * Reads a gzipped (or plain) intropolis file; sample_count <= 0 counts the distinct
 * sample-id strings first, as go_index does without -s.  The result holds exactly the
 * arrays morna_stage_junctions takes, plus the external sample id of every internal id
 * and the final junction -> frequency table (what .map.mor / .freq.mor store).
 */
typedef struct morna_lines morna_lines;
int morna_parse_intropolis(const char *path, int64_t sample_count, int64_t sample_threshold, morna_lines **out);
/* counts[8] = kept lines, nnz, n_items, skipped, sample_count, key bytes, distinct keys, lines read */
int morna_lines_counts(const morna_lines *L, int64_t *counts);
/* borrowed pointers, valid until morna_lines_free; any may be NULL */
int morna_lines_arrays(const morna_lines *L, const uint8_t **key_bytes, const int64_t **key_off,
                       const int64_t **row_ptr, const int32_t **item_ids, const int32_t **cov, const double **idf,
                       const int64_t **ext_ids);
int morna_lines_freq_entry(const morna_lines *L, int64_t i, const char **key, int64_t *key_len, int64_t *freq);
int morna_stage_lines(morna_index *h, const morna_lines *L);
/*
 * Binary pre-tokenised cache of a parse: what go_index's line loop
 * (morna.py:841-861) would recompute on every run over the same file.  tag[4] is the
 * caller's identity of the source (size, mtime, sample_count argument, threshold); load
 * hands it back so the caller can decide whether the cache is still valid.  MORNA_E_IO when
 * the file is missing, truncated or not such a cache.
 */
int morna_lines_save(const morna_lines *L, const char *path, const int64_t *tag);
int morna_lines_load(const char *path, int64_t *tag_out, morna_lines **out);
int morna_lines_free(morna_lines *L);
/*
 * Row shards of ONE parsed data set (the reference has no such path).  The parse is the global pass of
 * add_junction -- threshold on a line's whole sample list, cumulative frequency, idf = log(sample_count / freq) with the
 * GLOBAL sample count (morna.py:357-374), first-seen internal ids over the whole file (morna.py:377-382).  Shard `rank`
 * of `world` owns the global ids [rank * ceil(N / world), (rank + 1) * ceil(N / world)) and gets the lines restricted to
 * the entries of its items, in file order, ids renumbered from 0 (lines left without an entry are dropped; the frequency
 * table stays whole).  The matrices built from the shards, stacked by rank, are the matrix of the whole index bit for bit:
 * a cell's terms and their order (morna.py:376-388) do not depend on the other rows.
 *   morna_lines_shard_info  info[4] = {rank, world, id_offset (global id of local id 0), n_items of the whole data set}
 *   morna_lines_from_arrays the same object from arrays the caller tokenised itself (what morna_stage_junctions takes,
 *                           plus ext_ids[n_items], the external sample id of every internal id, and the sample count)
 */
int morna_lines_shard(const morna_lines *L, int32_t rank, int32_t world, morna_lines **out);
int morna_lines_shard_info(const morna_lines *L, int64_t *info);
int morna_lines_from_arrays(const uint8_t *key_bytes, const int64_t *key_off, int64_t J, const int64_t *row_ptr,
                            const int32_t *item_ids, const int32_t *cov, const double *idf, const int64_t *ext_ids,
                            int64_t n_items, int64_t sample_count, morna_lines **out);
/* Benchmark / test utility (no counterpart in the reference): J lines written as an intropolis text file, gzipped when
 * the path ends in ".gz": key words, "+", "GT", "AG", the sample list, the coverage list, tab separated. */
int morna_write_intropolis(const char *path, const uint8_t *key_bytes, const int64_t *key_off, int64_t J, const int64_t *row_ptr,
                           const int64_t *samples, const int32_t *cov);

/* AnnoyIndex.get_n_items()                                     morna.py:1174 */
int64_t morna_get_n_items(const morna_index *h);
/* AnnoyIndex.get_item_vector(i)                                morna.py:702 */
int morna_get_item_vector(morna_index *h, int32_t id, float *out);
/* rows of several items at once: out[n][dim], host or device memory (query vectors of a row-sharded search) */
int morna_get_item_vectors(morna_index *h, const int32_t *ids, int64_t n, float *out);
/* whole matrix / squared norms, for tests */
int morna_get_items(morna_index *h, float *rows_out /* [n][dim] */);
int morna_get_norms2(morna_index *h, float *out /* [n] */);

/* ---- forest -------------------------------------------------------------- */

/* AnnoyIndex.build(n_trees); seed 0 selects annoy's default 123456789   morna.py:425
 * Limit that annoy does not have: the build keeps a hyperplane and a row in registers / LDS and rejects dimensions whose
 * padded row (dim rounded up to 256 floats) exceeds 8192 floats (MORNA_E_INVALID).  BASELINE's configurations use 3000
 * and 8192. */
int morna_build(morna_index *h, int32_t n_trees, uint32_t seed);
int32_t morna_get_n_trees(const morna_index *h);

typedef struct {
    int64_t n_items, dim, leaf_capacity;   /* K = dim + 2 */
    int64_t n_trees, n_nodes, n_split, n_leaves, max_depth;
    int64_t split_attempts;                /* create_split calls (incl. rejected) */
    int64_t split_rows;                    /* sum of |node| over those calls      */
    int64_t fallback_nodes;                /* nodes randomised (imbalance > 0.99) */
} morna_forest_stats;
int morna_get_forest_stats(const morna_index *h, morna_forest_stats *out);
/*
 * Forest dump for structural tests.  node_rec[n_nodes][6] =
 * {kind (0 split, 1 leaf), tree, start, count, child0, child1}; perm[n_trees][n_items];
 * hyperplanes[n_split][dim] in the order given by hp_node[n_split] (node id).
 * Any output pointer may be NULL.
 */
int morna_get_forest(morna_index *h, int32_t *node_rec, int32_t *perm, float *hyperplanes, int32_t *hp_node);

/* ---- search -------------------------------------------------------------- */

/*
 * AnnoyIndex.get_nns_by_vector(v, n, search_k, include_distances)  morna.py:651, 659
 * batched over nq queries; q[nq][dim] fp32.  search_k = -1 -> k * n_trees.
 * ids_out[nq][k] (-1 padded), dist_out[nq][k] (may be NULL), count_out[nq] (may be NULL).
 * q may also point to memory of the handle's device (the row-sharded search hands over the
 * all-gathered query rows without a trip through the host); the outputs are host memory.
 */
int morna_get_nns_by_vector(morna_index *h, const float *q, int64_t nq, int32_t k, int32_t search_k,
                            int32_t *ids_out, float *dist_out, int32_t *count_out);
/* AnnoyIndex.get_nns_by_item(i, n, search_k, include_distances)    morna.py:762, 769, 1191 */
int morna_get_nns_by_item(morna_index *h, const int32_t *items, int64_t nq, int32_t k, int32_t search_k,
                          int32_t *ids_out, float *dist_out, int32_t *count_out);
/*
 * MornaSearch.exact_search_nn + cosine_distance                   morna.py:681-716, 101-114
 * q[nq][dim] fp64 (the un-rounded query_sample); distances fp64, accumulated in
 * the reference's sequential order; ties resolved as bisect_left does.
 * count_out[q] = -1: the reference RAISES for this query -- some indexed row is so nearly parallel to it that
 * cosine_distance's radicand rounds below zero and math.sqrt fails (morna.py:101-114); the reference evaluates every row
 * (morna.py:697-700), so the whole query fails whatever that row's rank would have been.  The lists of such a query are
 * filled in for diagnosis only; a caller must test the count before it slices by it.
 */
int morna_exact_search(morna_index *h, const double *q, int64_t nq, int32_t k,
                       int32_t *ids_out, double *dist_out, int32_t *count_out);
/*
 * The same with STORED rows as the queries (every item of the index against the index: BASELINE configs[4]): the query
 * is the item's fp32 row widened to fp64 on the device -- what exact_search_nn sees when query_sample came out of
 * get_item_vector (morna.py:697-703) -- so nothing but the item numbers crosses PCIe.  The queries are processed in
 * batches inside the call (scan values of a batch: at most 2 GiB).
 */
int morna_exact_search_by_item(morna_index *h, const int32_t *items, int64_t nq, int32_t k,
                               int32_t *ids_out, double *dist_out, int32_t *count_out);

/*
 * Row-sharded search (one handle per GPU; the reference has no such path): merge of the
 * per-shard answers to the same queries after their all-gather.  ids / dist: [world][nq][kk], each
 * [kk] list as get_nns_* returns it -- ascending (distance, id), empty slots id -1 last -- with ids
 * already global.  Writes the k smallest (distance, id) pairs per query; out slots past the
 * count are id -1 / distance +inf.  Host memory, no GPU work.
 */
int morna_merge_topk(const int64_t *ids, const float *dist, int32_t world, int64_t nq, int32_t kk, int32_t k,
                     int64_t *ids_out, float *dist_out, int32_t *count_out);

/*
 * The same exchange with the answers resident in HBM (one handle per GPU, RCCL all-gather between the two calls):
 *   morna_get_nns_by_vector_packed  as morna_get_nns_by_vector (q: host or this device's memory), but the answers are
 *       written to packed_dev -- memory of the handle's device, [nq][2k] int32: the k global ids (local id + id_offset,
 *       -1 = empty) followed by the bits of the k fp32 distances: Q * k * 8 bytes per rank.  ENQUEUED on
 *       the handle's stream when the call returns (q, when it is device memory, is read there too): work the caller
 *       orders behind that stream sees the message; morna_synchronize() waits for it.
 *   morna_merge_topk_packed  gathered_dev[world][nq][2kk] (the all-gathered messages, device memory; read on the
 *       handle's stream: complete before the call, or produced by work ordered on that stream) -> the k smallest
 *       (distance, id) per query, host memory, complete when the call returns.
 *   morna_get_item_vectors_dev  rows of `ids` -> out_dev[n][dim] (device memory), enqueued on the handle's stream.
 *   morna_get_stream  the handle's HIP stream (a hipStream_t), so that the caller can put its collectives between
 *       these calls in stream order instead of waiting on the host (torch: torch.cuda.ExternalStream).
 * world <= 64, kk <= 255, global ids below 2^31.
 */
int morna_get_nns_by_vector_packed(morna_index *h, const float *q, int64_t nq, int32_t k, int32_t search_k, int64_t id_offset,
                                   int32_t *packed_dev);
int morna_get_item_vectors_dev(morna_index *h, const int32_t *ids, int64_t n, float *out_dev);
int morna_get_stream(morna_index *h, void **stream_out);
int morna_merge_topk_packed(morna_index *h, const int32_t *gathered_dev, int32_t world, int64_t nq, int32_t kk, int32_t k,
                            int32_t *ids_out, float *dist_out, int32_t *count_out);

/*
 * ---- row-sharded search with the communicator inside the library ("the RCCL communicator owned by the
 * handle"; 8e).  One handle per GPU and process; rank g holds the rows with global ids [off[g], off[g+1]) -- off from the
 * ranks' own item counts, exchanged by the library -- and its own forest.  Every function below is COLLECTIVE: all ranks
 * call it with the same nq / k / search_k (and the same queries, where queries are passed).  Data path: per-shard search
 * -> ncclAllGather of the per-shard top-k (Q * k * 8 bytes per rank; exact: Q * (12 k + 4)) on the handle's stream ->
 * merge kernel; every rank receives the same merged answer (global ids).  RCCL is loaded at run time; without it
 * morna_comm_init fails and nothing falls back to the host.  Arguments are checked before the first collective of a call;
 * as in any RCCL program, a call that fails on ONE rank only (an item number out of that shard's range) leaves the other
 * ranks waiting in the collective: validate rank-local input before calling.
 *   morna_comm_unique_id   ncclGetUniqueId: ONE rank makes the id, the caller hands the 128 bytes to the others (any
 *                          channel: a file, MPI, torch.distributed's store)
 *   morna_comm_init        ncclCommInitRank on the handle's device; world <= 64
 *   morna_comm_info        rank, world and (when offsets != NULL: collective, always exchanges) offsets[world + 1].  The
 *                          sharded calls exchange the row counts themselves at their first use and whenever THIS handle's
 *                          row count has changed; ranks that resize must do so together (or call this on every rank)
 *   *_by_item_sharded      every rank contributes stored rows of ITS shard as queries (local ids); the answers come back
 *                          for all ranks' queries, rank 0's first.  n_each[world] = every rank's query count, or NULL
 *                          (then the counts are exchanged first).  Query rows travel HBM -> xGMI -> HBM.
 *   exact variants         merged as exact_search_nn's bisect_left scan over ALL rows would (equal distance: higher global
 *                          id first); count -1 as morna_exact_search, true of the whole matrix if true of one shard
 */
#define MORNA_COMM_ID_BYTES 128
int morna_comm_unique_id(uint8_t *id_out /* [MORNA_COMM_ID_BYTES] */);
int morna_comm_init(morna_index *h, const uint8_t *id, int32_t rank, int32_t world);
int morna_comm_destroy(morna_index *h);
int morna_comm_info(morna_index *h, int32_t *rank, int32_t *world, int64_t *offsets);
int morna_get_nns_by_vector_sharded(morna_index *h, const float *q, int64_t nq, int32_t k, int32_t search_k,
                                    int32_t *ids_out, float *dist_out, int32_t *count_out);
int morna_get_nns_by_item_sharded(morna_index *h, const int32_t *local_items, int64_t n_local, const int64_t *n_each,
                                  int32_t k, int32_t search_k, int32_t *ids_out, float *dist_out, int32_t *count_out);
int morna_exact_search_sharded(morna_index *h, const double *q, int64_t nq, int32_t k,
                               int32_t *ids_out, double *dist_out, int32_t *count_out);
int morna_exact_search_by_item_sharded(morna_index *h, const int32_t *local_items, int64_t n_local, const int64_t *n_each,
                                       int32_t k, int32_t *ids_out, double *dist_out, int32_t *count_out);
/*
 * The two halves of the exact exchange on their own (a caller with its own transport; tests that lay several shards'
 * messages side by side): the per-shard answers packed in HBM -- morna_exact_packed_bytes(nq, k) bytes: ids int32
 * [nq][k] (local + id_offset, -1 = empty) | count int32 [nq] | pad to 8 | distances fp64 [nq][k]; queries: exactly one
 * of q (host fp64 [nq][dim]), q_dev (fp32 [nq][dim], this device) and items (stored rows); enqueued on the handle's
 * stream -- and the merge of `world` such messages laid end to end in device memory (host results, waits).
 */
int64_t morna_exact_packed_bytes(int64_t nq, int32_t k);
int morna_exact_search_packed(morna_index *h, const double *q, const float *q_dev, const int32_t *items, int64_t nq, int32_t k,
                              int64_t id_offset, uint8_t *packed_dev);
int morna_merge_exact_packed(morna_index *h, const uint8_t *gathered_dev, int32_t world, int64_t nq, int32_t kk, int32_t k,
                             int32_t *ids_out, double *dist_out, int32_t *count_out);

/* ---- persistence (stands in for AnnoyIndex.save / load)      morna.py:439, 544 */
int morna_save(morna_index *h, const char *path);
int morna_load(morna_index *h, const char *path);

/* ---- measurement --------------------------------------------------------- */

enum {
    MORNA_T_FEATURES = 0,   /* hash + accumulate + transpose/convert + norms   */
    MORNA_T_TWO_MEANS = 1,  /* forest: centroid kernel                         */
    MORNA_T_SPLIT = 2,      /* forest: hyperplane margin / side kernel (dominant) */
    MORNA_T_PARTITION = 3,  /* forest: stable partition                        */
    MORNA_T_QUERY = 4,      /* traversal + refine + top-k                      */
    MORNA_T_EXACT = 5,      /* exact scan + re-rank                            */
    MORNA_T_QUERY_FILTER = 6, /* part of MORNA_T_QUERY: the whole-batch fp16 contraction (bytes = its flops) */
    MORNA_T_EXACT_SCAN = 7, /* part of MORNA_T_EXACT: the fp32 scan (bytes = its flops when it ran on the matrix cores) */
    MORNA_T_SPLIT_MM = 8,   /* part of MORNA_T_SPLIT: the fp16 contraction alone (bytes = the flops of the tiles it LAUNCHED) */
    MORNA_T_TM_STRIP = 9,   /* part of MORNA_T_TWO_MEANS: levels run by two_means_strip_kernel (four waves per node) */
    MORNA_T_TM_WAVE = 10,   /* part of MORNA_T_TWO_MEANS: levels run by two_means_wave_kernel (one wave per node) */
    MORNA_T_COUNT = 11
};
/* HIP-event timing of the kernels on the handle's own stream.  on: 0 off, 1 every group, otherwise a mask with bit
 * (MORNA_T_x + 1) set for each group to time (each event pair costs the stream a few microseconds of idle) */
int morna_timer_enable(morna_index *h, int32_t on);
int morna_timer_reset(morna_index *h);
/* ms = summed event time, launches, bytes = algorithmic bytes of those launches */
int morna_timer_read(morna_index *h, int32_t which, double *ms, int64_t *launches, int64_t *bytes);
int morna_synchronize(morna_index *h);

#ifdef __cplusplus
}
#endif
#endif /* MORNA_HIP_H */
