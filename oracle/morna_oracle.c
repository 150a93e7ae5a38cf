/*
 * oracle/morna_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the arithmetic on morna's index-build + search hot path,
 * used only as the checker by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing under morna_amd/ may import, link or call it.
 *
 * What it restates (reference = /root/reference, read as text only):
 *   - mmh3.hash (MurmurHash3_x86_32, seed 0, signed)    call sites morna.py:369, 591, 625
 *   - MornaIndex.add_junction feature hashing + TF-IDF   morna.py:344-388
 *   - MornaIndex.build fp64 -> fp32 hand-off (add_item)  morna.py:399-424
 *   - MornaSearch.finalize_query                         morna.py:609-629
 *   - cosine_distance + exact_search_nn                  morna.py:101-114, 681-716
 *
 * mmh3 is a third-party dependency that is NOT under /root/reference (unpinned,
 * "pip install mmh3", README.md:12); the function below restates the published
 * MurmurHash3_x86_32 algorithm (Appleby, public domain) and is pinned by mmh3's
 * documented answers and by vectors generated in the authoring container with
 * sklearn.utils.murmurhash3_32 (tests/golden/murmur3_vectors.json).
 *
 * Parity status: PINNED for features / exact ordering by the reference's own
 * embedded known answers (morna.py:1176-1187, 1267-1278, 1312-1323; see
 * tests/test_oracle_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: Python does cov*idf,
 * *mult and += as three separately rounded fp64 operations, so no FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ murmur3 */

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* mmh3.hash(key) with the default seed 0; returns the signed 32-bit value. */
int32_t oracle_mmh3_32(const uint8_t *key, int64_t len, uint32_t seed)
{
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
    uint32_t h1 = seed;
    int64_t nblocks = len / 4;
    for (int64_t i = 0; i < nblocks; i++) {
        uint32_t k1 = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) |
                      ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
        k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
        h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64u;
    }
    const uint8_t *tail = key + nblocks * 4;
    uint32_t k1 = 0;
    switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k1 ^= (uint32_t)tail[1] << 8;  /* fallthrough */
    case 1: k1 ^= tail[0];
            k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint32_t)len;
    h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;
    return (int32_t)h1;
}

/* Python's `h % dim` for dim > 0 (floored modulo, morna.py:371). */
static inline int32_t py_mod(int32_t h, int32_t dim)
{
    int32_t r = h % dim;
    return r < 0 ? r + dim : r;
}

void oracle_hash_col_sign(const uint8_t *keys, const int64_t *key_off, int64_t J, int32_t dim,
                          int32_t *hash_out, int32_t *col_out, int32_t *sign_out)
{
    for (int64_t j = 0; j < J; j++) {
        int32_t h = oracle_mmh3_32(keys + key_off[j], key_off[j + 1] - key_off[j], 0);
        hash_out[j] = h;
        sign_out[j] = h < 0 ? -1 : 1;      /* morna.py:370, taken before the modulo */
        col_out[j] = py_mod(h, dim);       /* morna.py:371 */
    }
}

/* ------------------------------------------------- tiny open-addressing maps */

typedef struct {
    int64_t cap, used;
    int64_t *keys;   /* -1 = empty (sample ids are non-negative) */
    int64_t *vals;
} i64map;

static void i64map_init(i64map *m, int64_t cap)
{
    m->cap = 16;
    while (m->cap < cap * 2) m->cap <<= 1;
    m->used = 0;
    m->keys = (int64_t *)malloc(sizeof(int64_t) * m->cap);
    m->vals = (int64_t *)malloc(sizeof(int64_t) * m->cap);
    for (int64_t i = 0; i < m->cap; i++) m->keys[i] = INT64_MIN;
}
static void i64map_free(i64map *m) { free(m->keys); free(m->vals); }
static uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
static void i64map_grow(i64map *m);
/* returns pointer to the value slot; *fresh set when the key was inserted now */
static int64_t *i64map_get(i64map *m, int64_t key, int *fresh)
{
    if (m->used * 2 >= m->cap) i64map_grow(m);
    uint64_t pos = mix64((uint64_t)key) & (uint64_t)(m->cap - 1);
    for (;;) {
        if (m->keys[pos] == INT64_MIN) {
            m->keys[pos] = key; m->vals[pos] = 0; m->used++;
            *fresh = 1;
            return &m->vals[pos];
        }
        if (m->keys[pos] == key) { *fresh = 0; return &m->vals[pos]; }
        pos = (pos + 1) & (uint64_t)(m->cap - 1);
    }
}
static void i64map_grow(i64map *m)
{
    i64map n;
    n.cap = m->cap * 2; n.used = 0;
    n.keys = (int64_t *)malloc(sizeof(int64_t) * n.cap);
    n.vals = (int64_t *)malloc(sizeof(int64_t) * n.cap);
    for (int64_t i = 0; i < n.cap; i++) n.keys[i] = INT64_MIN;
    for (int64_t i = 0; i < m->cap; i++)
        if (m->keys[i] != INT64_MIN) {
            uint64_t pos = mix64((uint64_t)m->keys[i]) & (uint64_t)(n.cap - 1);
            while (n.keys[pos] != INT64_MIN) pos = (pos + 1) & (uint64_t)(n.cap - 1);
            n.keys[pos] = m->keys[i]; n.vals[pos] = m->vals[i]; n.used++;
        }
    free(m->keys); free(m->vals);
    *m = n;
}

/* string-keyed map junction -> cumulative sample frequency (morna.py:365) */
typedef struct {
    int64_t cap, used;
    int64_t *line;   /* index of a line holding the key bytes, -1 = empty */
    int64_t *freq;
} strmap;

static uint64_t str_hash(const uint8_t *s, int64_t n)
{
    uint64_t h = 1469598103934665603ULL;
    for (int64_t i = 0; i < n; i++) { h ^= s[i]; h *= 1099511628211ULL; }
    return mix64(h);
}

/* ------------------------------------------------------ feature accumulation */

/*
 * Restates go_index's line loop + MornaIndex.add_junction (morna.py:841-861,
 * 344-388) on pre-tokenised input.
 *
 *   keys/key_off      J junction key strings "chrom start end" (morna.py:849/858)
 *   row_ptr[J+1]      extent of each line's sample / coverage lists
 *   samples, cov      as parsed by map(int, ...) (morna.py:851-852)
 *   sample_count      -s value or count_samples() result
 *   threshold         --sample-threshold (morna.py:361)
 *   M                 [max_items][dim] fp64, zero-initialised by the caller:
 *                     sample_feature_matrix (morna.py:184-186) keyed by internal id
 *   ext_ids[max_items] external sample id of each internal id (inverse of
 *                     internal_id_map, morna.py:378-382)
 *   idf_out[J]        idf used for the line, NaN when the line was skipped
 *   freq_out[J]       cumulative frequency after the line (0 when skipped)
 * Returns the number of internal ids assigned (new_internal_id), or -1 when
 * max_items is too small.
 */
int64_t oracle_index_features(const uint8_t *keys, const int64_t *key_off, int64_t J,
                              const int64_t *row_ptr, const int64_t *samples, const int64_t *cov,
                              int64_t sample_count, int64_t threshold, int32_t dim,
                              double *M, int64_t max_items, int64_t *ext_ids,
                              double *idf_out, int64_t *freq_out, int64_t *skipped_out)
{
    i64map ids;
    i64map_init(&ids, 1024);
    strmap fm;
    fm.cap = 16;
    while (fm.cap < J * 2 + 2) fm.cap <<= 1;
    fm.used = 0;
    fm.line = (int64_t *)malloc(sizeof(int64_t) * fm.cap);
    fm.freq = (int64_t *)malloc(sizeof(int64_t) * fm.cap);
    for (int64_t i = 0; i < fm.cap; i++) fm.line[i] = -1;

    int64_t new_internal_id = 0, skipped = 0;
    for (int64_t j = 0; j < J; j++) {
        int64_t b = row_ptr[j], e = row_ptr[j + 1], n = e - b;
        if (idf_out) idf_out[j] = NAN;
        if (freq_out) freq_out[j] = 0;
        if (n < threshold) { skipped++; continue; }              /* morna.py:361-363 */
        const uint8_t *ks = keys + key_off[j];
        int64_t kl = key_off[j + 1] - key_off[j];
        /* sample_frequencies[junction] += len(samples)             morna.py:365 */
        uint64_t pos = str_hash(ks, kl) & (uint64_t)(fm.cap - 1);
        for (;;) {
            int64_t l = fm.line[pos];
            if (l < 0) { fm.line[pos] = j; fm.freq[pos] = 0; break; }
            int64_t ll = key_off[l + 1] - key_off[l];
            if (ll == kl && memcmp(keys + key_off[l], ks, (size_t)kl) == 0) break;
            pos = (pos + 1) & (uint64_t)(fm.cap - 1);
        }
        fm.freq[pos] += n;
        int64_t freq = fm.freq[pos];
        int32_t h = oracle_mmh3_32(ks, kl, 0);                    /* morna.py:369 */
        double mult = h < 0 ? -1.0 : 1.0;                         /* morna.py:370 */
        int32_t col = py_mod(h, dim);                             /* morna.py:371 */
        double idf = log((double)sample_count / (double)freq);    /* morna.py:372-374 */
        if (idf_out) idf_out[j] = idf;
        if (freq_out) freq_out[j] = freq;
        for (int64_t t = b; t < e; t++) {                         /* morna.py:376-388 */
            int fresh;
            int64_t *slot = i64map_get(&ids, samples[t], &fresh);
            if (fresh) {
                if (new_internal_id >= max_items) {
                    i64map_free(&ids); free(fm.line); free(fm.freq);
                    return -1;
                }
                *slot = new_internal_id;
                ext_ids[new_internal_id] = samples[t];
                new_internal_id++;
            }
            int64_t id = *slot;
            double tf_idf = (double)cov[t] * idf;                 /* morna.py:384 */
            double add = mult * tf_idf;                           /* morna.py:388 */
            M[id * (int64_t)dim + col] += add;
        }
    }
    if (skipped_out) *skipped_out = skipped;
    i64map_free(&ids); free(fm.line); free(fm.freq);
    return new_internal_id;
}

/* add_item hand-off: Python float -> C float is a round-to-nearest cast. */
void oracle_f64_to_f32(const double *src, float *dst, int64_t n)
{
    for (int64_t i = 0; i < n; i++) dst[i] = (float)src[i];
}

/*
 * finalize_query (morna.py:609-629) on a de-duplicated query: for each distinct
 * junction key, cov_sum (update_query, morna.py:606) and the stored frequency
 * (0 = unknown junction -> idf 0).  Python iterates dict keys in arbitrary
 * order; cells hit by more than one junction are therefore order-dependent in
 * the reference itself, the caller passes the order it wants pinned.
 */
void oracle_finalize_query(const uint8_t *keys, const int64_t *key_off, int64_t n_keys,
                           const int64_t *cov_sum, const int64_t *freq, int64_t sample_count,
                           int32_t dim, double *q /* [dim], zeroed here */)
{
    for (int32_t z = 0; z < dim; z++) q[z] = 0.0;
    for (int64_t j = 0; j < n_keys; j++) {
        double idf = 0.0;
        if (freq[j] != 0) idf = log((double)sample_count / (double)freq[j]);
        int32_t h = oracle_mmh3_32(keys + key_off[j], key_off[j + 1] - key_off[j], 0);
        double mult = h < 0 ? -1.0 : 1.0;
        double t = (double)cov_sum[j] * idf;
        q[py_mod(h, dim)] += mult * t;
    }
}

/* ------------------------------------------------------------- exact search */

/* cosine_distance (morna.py:101-114): v1 = fp32 row widened, v2 = fp64 query. */
double oracle_cosine_distance(const float *row, const double *q, int32_t dim)
{
    double pp = 0.0, qq = 0.0, pq = 0.0;
    for (int32_t z = 0; z < dim; z++) {
        double i = (double)row[z], j = q[z];
        pp += i * i;
        qq += j * j;
        pq += i * j;
    }
    double ppqq = pp * qq, distance;
    if (ppqq > 0.0) distance = 2.0 - 2.0 * pq / sqrt(ppqq);
    else distance = 2.0;
    return sqrt(distance);   /* NaN where Python's math.sqrt would raise ValueError */
}

/*
 * exact_search_nn (morna.py:697-713): scan i = 0..n-1, bisect_left insertion,
 * truncate to k.  Among equal distances the later (higher) id is placed first.
 * Returns the number of results written (min(k, n)).
 */
int64_t oracle_exact_search(const float *X, int64_t n, int32_t dim, int64_t stride,
                            const double *q, int64_t k, int64_t *ids_out, double *dist_out)
{
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        double d = oracle_cosine_distance(X + i * stride, q, dim);
        /* bisect_left over dist_out[0..m) */
        int64_t lo = 0, hi = m;
        while (lo < hi) {
            int64_t mid = (lo + hi) / 2;
            if (dist_out[mid] < d) lo = mid + 1; else hi = mid;
        }
        if (lo < k) {
            int64_t last = m < k ? m : k - 1;   /* element falling off the end is dropped */
            for (int64_t t = last; t > lo; t--) { dist_out[t] = dist_out[t - 1]; ids_out[t] = ids_out[t - 1]; }
            dist_out[lo] = d; ids_out[lo] = i;
            if (m < k) m++;
        }
    }
    return m;
}
