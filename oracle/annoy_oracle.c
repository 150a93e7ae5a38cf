/*
 * oracle/annoy_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the random-projection forest that morna delegates to the
 * third-party `annoy` package (AnnoyIndex(dim, metric='angular'); call sites
 * /root/reference/morna.py:166, 406, 423, 425, 543, 651, 659, 762, 769, 702).
 *
 * annoy is NOT under /root/reference and is unpinned there ("pip install
 * annoy", README.md:12); this file restates spotify/annoy's published
 * algorithm as summarised in SURVEY.md section 2.1:
 *   Kiss32Random (default seed 123456789), two_means (200 iterations),
 *   Angular::create_split / side / margin / distance, _make_tree with leaf
 *   capacity K = f + 2 and the 3-attempt / 0.95-imbalance rule with random
 *   fallback above 0.99, _get_all_nns (max-heap on (bound, node), sort+unique,
 *   sort by (distance, id)), normalized_distance = sqrt(max(d, 0)).
 *
 * PARITY STATUS: "parity unpinned" for forest structure, search_k behaviour and
 * approximate recall -- no reference test has N > K, the package version is
 * unpinned and its source is absent.  What IS pinned (by morna.py:1176-1187,
 * 1267-1278, 1312-1323) is the N <= K case where every root is one leaf and
 * the result is the exact (distance, id) ordering.
 *
 * Two modes, selected at create time:
 *   mode 0 "faithful"  : one sequential Kiss32 stream, depth-first recursion,
 *                        sequential fp32 dot products (annoy's non-AVX path).
 *   mode 1 "wave order": the SAME algorithm with the three changes a level-
 *                        synchronous GPU build needs, so that the HIP forest can
 *                        be compared node for node, bit for bit:
 *                          (a) dot products summed in the 64-lane canonical
 *                              order of the HIP kernels (cdot below);
 *                          (b) one Kiss32 stream per (tree, level, segment
 *                              start, attempt) instead of one global stream,
 *                              and counter-based coin flips per item position;
 *                          (c) breadth-first node numbering.
 *
 * Build: make -C oracle (gcc -O2 -ffp-contract=off -fno-fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------- RNG */

typedef struct { uint32_t x, y, z, c; } kiss32;

static void kiss_seed(kiss32 *r, uint32_t seed)
{
    r->x = seed; r->y = 362436000u; r->z = 521288629u; r->c = 7654321u;
}
static uint32_t kiss_next(kiss32 *r)
{
    r->x = 69069u * r->x + 12345u;
    r->y ^= r->y << 13; r->y ^= r->y >> 17; r->y ^= r->y << 5;
    uint64_t t = 698769069ULL * r->z + r->c;
    r->c = (uint32_t)(t >> 32);
    r->z = (uint32_t)t;
    return r->x + r->y + r->z;
}
static int kiss_flip(kiss32 *r) { return (int)(kiss_next(r) & 1u); }
static size_t kiss_index(kiss32 *r, size_t n) { return (size_t)(kiss_next(r) % (uint32_t)n); }

static uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
/* mode 1: seed of the stream owned by one split attempt of one node */
static uint32_t node_seed(uint32_t seed, uint32_t tree, uint32_t level, uint32_t start, uint32_t attempt)
{
    uint32_t h = fmix32(seed + 0x9E3779B9u * (tree + 1u));
    h = fmix32(h ^ (level * 0x85ebca6bu + 0x27d4eb2fu));
    h = fmix32(h ^ start);
    h = fmix32(h + attempt * 0xc2b2ae35u + 0x165667b1u);
    return h ? h : 1u;
}
/* mode 1: coin flip for the item at position i of the node's segment */
static int pos_flip(uint32_t nseed, uint32_t i)
{
    return (int)(fmix32(nseed ^ fmix32(i + 0x632BE5ABu)) & 1u);
}

/* ------------------------------------------------------------ dot products */

static float dot_seq(const float *a, const float *b, int f)
{
    float s = 0;
    for (int z = 0; z < f; z++) s += a[z] * b[z];
    return s;
}

/*
 * Canonical 64-lane order of the HIP kernels: lane l owns elements
 * 256*k + 4*l + c (c = 0..3), keeps one fmaf chain per c over k, folds them as
 * (a0 + a1) + (a2 + a3), then the lanes are combined by an xor butterfly
 * (offsets 32, 16, 8, 4, 2, 1).
 */
static float dot_canon(const float *a, const float *b, int f)
{
    float t[64];
    for (int l = 0; l < 64; l++) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int base = 4 * l; base < f; base += 256)
            for (int c = 0; c < 4; c++) {
                int idx = base + c;
                if (idx < f) acc[c] = fmaf(a[idx], b[idx], acc[c]);
            }
        t[l] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float u[64];
        for (int l = 0; l < 64; l++) u[l] = t[l] + t[l ^ off];
        memcpy(t, u, sizeof(t));
    }
    return t[0];
}

/* Angular::distance: T=float fields, double literals (2.0 - 2.0 * pq / sqrt(ppqq)) */
static float ang_dist(float pp, float qq, float pq)
{
    float ppqq = pp * qq;
    if (ppqq > 0) return (float)(2.0 - 2.0 * (double)pq / (double)sqrtf(ppqq));
    return 2.0f;
}

/* ------------------------------------------------------------------ index */

enum { NODE_SPLIT = 0, NODE_LEAF = 1 };

typedef struct {
    int kind;
    int n_desc;
    int child[2];      /* split: node ids (mode 0: ids < n_items are items) */
    float *v;          /* split: hyperplane [f] */
    int *items;        /* leaf: item ids */
    int tree, level;   /* mode 1 bookkeeping */
} onode;

typedef struct {
    int f, K, mode;
    int n_items;
    float *X;          /* [n_items][f] */
    float *norm2;      /* dot(x, x) per item (Node::norm cache, annoy >= 1.16) */
    onode *nodes;      /* tree nodes; mode 0 ids are offset by n_items */
    int n_nodes, cap_nodes;
    int *roots;
    int n_roots;
    kiss32 rng;        /* mode 0 global stream */
    uint32_t seed;
    /* counters for the roofline bookkeeping of the GPU build */
    int64_t split_rows;   /* sum over accepted+rejected split attempts of |node| */
    int64_t split_nodes;
} annoyo;

static float xdot(const annoyo *a, const float *x, const float *y)
{
    return a->mode ? dot_canon(x, y, a->f) : dot_seq(x, y, a->f);
}

annoyo *annoyo_create(int f, int mode)
{
    annoyo *a = (annoyo *)calloc(1, sizeof(annoyo));
    a->f = f; a->K = f + 2; a->mode = mode;
    a->seed = 123456789u;
    kiss_seed(&a->rng, a->seed);
    return a;
}

void annoyo_set_seed(annoyo *a, uint32_t seed)
{
    a->seed = seed;
    kiss_seed(&a->rng, seed);
}

void annoyo_destroy(annoyo *a)
{
    if (!a) return;
    for (int i = 0; i < a->n_nodes; i++) { free(a->nodes[i].v); free(a->nodes[i].items); }
    free(a->nodes); free(a->roots); free(a->X); free(a->norm2); free(a);
}

/* add_item for ids 0..n-1 in one call (morna adds dense internal ids, morna.py:405-424) */
void annoyo_set_items(annoyo *a, const float *X, int n)
{
    a->n_items = n;
    a->X = (float *)malloc(sizeof(float) * (size_t)n * a->f);
    memcpy(a->X, X, sizeof(float) * (size_t)n * a->f);
    a->norm2 = (float *)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; i++) a->norm2[i] = xdot(a, a->X + (size_t)i * a->f, a->X + (size_t)i * a->f);
}

static int new_node(annoyo *a)
{
    if (a->n_nodes == a->cap_nodes) {
        a->cap_nodes = a->cap_nodes ? a->cap_nodes * 2 : 256;
        a->nodes = (onode *)realloc(a->nodes, sizeof(onode) * a->cap_nodes);
    }
    memset(&a->nodes[a->n_nodes], 0, sizeof(onode));
    return a->n_nodes++;
}

static void normalize(const annoyo *a, float *v)
{
    float norm = sqrtf(xdot(a, v, v));
    if (norm > 0)
        for (int z = 0; z < a->f; z++) v[z] /= norm;
}

/* two_means + create_split: fills n[f] with the unit normal of the split */
static void create_split(annoyo *a, const int *items, size_t count, kiss32 *rng, float *n)
{
    const int f = a->f;
    float *p = (float *)malloc(sizeof(float) * f), *q = (float *)malloc(sizeof(float) * f);
    size_t i = kiss_index(rng, count);
    size_t j = kiss_index(rng, count - 1);
    j += (j >= i);
    memcpy(p, a->X + (size_t)items[i] * f, sizeof(float) * f);
    memcpy(q, a->X + (size_t)items[j] * f, sizeof(float) * f);
    normalize(a, p); normalize(a, q);
    float pp = xdot(a, p, p), qq = xdot(a, q, q);
    int ic = 1, jc = 1;
    for (int l = 0; l < 200; l++) {
        size_t k = kiss_index(rng, count);
        const float *x = a->X + (size_t)items[k] * f;
        float nk2 = a->norm2[items[k]];
        float di = ic * ang_dist(pp, nk2, xdot(a, p, x));
        float dj = jc * ang_dist(qq, nk2, xdot(a, q, x));
        float norm = sqrtf(nk2);
        if (!(norm > 0)) continue;
        if (di < dj) {
            for (int z = 0; z < f; z++) p[z] = (p[z] * ic + x[z] / norm) / (ic + 1);
            pp = xdot(a, p, p);
            ic++;
        } else if (dj < di) {
            for (int z = 0; z < f; z++) q[z] = (q[z] * jc + x[z] / norm) / (jc + 1);
            qq = xdot(a, q, q);
            jc++;
        }
    }
    for (int z = 0; z < f; z++) n[z] = p[z] - q[z];
    normalize(a, n);
    free(p); free(q);
}

static double split_imbalance(size_t left, size_t right)
{
    double ls = (float)left, rs = (float)right;
    float fr = (float)(ls / (ls + rs + 1e-9));
    return fr > 1 - fr ? fr : 1 - fr;
}

/* ------------------------------------------------ mode 0: depth-first build */

static int make_tree0(annoyo *a, const int *idx, size_t n, int is_root)
{
    if (n == 1 && !is_root) return idx[0];
    if (n <= (size_t)a->K && (!is_root || (size_t)a->n_items <= (size_t)a->K || n == 1)) {
        int id = new_node(a);
        onode *m = &a->nodes[id];
        m->kind = NODE_LEAF;
        m->n_desc = is_root ? a->n_items : (int)n;
        m->items = (int *)malloc(sizeof(int) * (n ? n : 1));
        memcpy(m->items, idx, sizeof(int) * n);
        /* a root leaf advertises n_items descendants; keep the real count too */
        m->child[0] = (int)n;
        return id + a->n_items;
    }
    const int f = a->f;
    float *hv = (float *)malloc(sizeof(float) * f);
    int *side_of = (int *)malloc(sizeof(int) * n);
    size_t cnt[2] = {0, 0};
    for (int attempt = 0; attempt < 3; attempt++) {
        create_split(a, idx, n, &a->rng, hv);
        a->split_rows += (int64_t)n; a->split_nodes++;
        cnt[0] = cnt[1] = 0;
        for (size_t i = 0; i < n; i++) {
            float d = xdot(a, hv, a->X + (size_t)idx[i] * f);
            int s = d != 0 ? (d > 0) : kiss_flip(&a->rng);
            side_of[i] = s; cnt[s]++;
        }
        if (split_imbalance(cnt[0], cnt[1]) < 0.95) break;
    }
    while (split_imbalance(cnt[0], cnt[1]) > 0.99) {
        for (int z = 0; z < f; z++) hv[z] = 0;
        cnt[0] = cnt[1] = 0;
        for (size_t i = 0; i < n; i++) { int s = kiss_flip(&a->rng); side_of[i] = s; cnt[s]++; }
    }
    int *ch[2];
    ch[0] = (int *)malloc(sizeof(int) * (cnt[0] ? cnt[0] : 1));
    ch[1] = (int *)malloc(sizeof(int) * (cnt[1] ? cnt[1] : 1));
    size_t w[2] = {0, 0};
    for (size_t i = 0; i < n; i++) ch[side_of[i]][w[side_of[i]]++] = idx[i];
    free(side_of);
    int flip = cnt[0] > cnt[1];
    int child[2];
    for (int side = 0; side < 2; side++)     /* smallest child first */
        child[side ^ flip] = make_tree0(a, ch[side ^ flip], cnt[side ^ flip], 0);
    free(ch[0]); free(ch[1]);
    int id = new_node(a);
    onode *m = &a->nodes[id];
    m->kind = NODE_SPLIT;
    m->n_desc = is_root ? a->n_items : (int)n;
    m->child[0] = child[0]; m->child[1] = child[1];
    m->v = hv;
    return id + a->n_items;
}

/* ---------------------------------------------- mode 1: breadth-first build */

typedef struct { int tree, level, start, count, node; } seg;

static void build1(annoyo *a, int n_trees)
{
    const int f = a->f, N = a->n_items;
    int *perm = (int *)malloc(sizeof(int) * (size_t)N * n_trees);
    int *tmp = (int *)malloc(sizeof(int) * (size_t)N);
    uint8_t *side_of = (uint8_t *)malloc((size_t)N);
    seg *cur = (seg *)malloc(sizeof(seg) * n_trees), *nxt = NULL;
    int n_cur = n_trees;
    a->roots = (int *)malloc(sizeof(int) * n_trees);
    a->n_roots = n_trees;
    for (int t = 0; t < n_trees; t++) {
        for (int i = 0; i < N; i++) perm[(size_t)t * N + i] = i;
        int id = new_node(a);
        a->roots[t] = id;
        cur[t].tree = t; cur[t].level = 0; cur[t].start = 0; cur[t].count = N; cur[t].node = id;
    }
    while (n_cur > 0) {
        /* children are numbered base + 2*i + side for the i-th SPLIT node of this level */
        int n_split = 0;
        for (int i = 0; i < n_cur; i++) if (cur[i].count > a->K) n_split++;
        nxt = (seg *)malloc(sizeof(seg) * (size_t)(2 * n_split + 1));
        int n_nxt = 0;
        for (int i = 0; i < n_cur; i++) {
            seg s = cur[i];
            int *items = perm + (size_t)s.tree * N + s.start;
            onode *m = &a->nodes[s.node];
            m->tree = s.tree; m->level = s.level; m->n_desc = s.count;
            if (s.count <= a->K) {
                m->kind = NODE_LEAF;
                m->items = (int *)malloc(sizeof(int) * (size_t)(s.count ? s.count : 1));
                memcpy(m->items, items, sizeof(int) * (size_t)s.count);
                m->child[0] = s.count;
                continue;
            }
            float *hv = (float *)malloc(sizeof(float) * f);
            size_t cnt[2] = {0, 0};
            uint32_t ns = 0;
            for (int attempt = 0; attempt < 3; attempt++) {
                kiss32 rng;
                ns = node_seed(a->seed, (uint32_t)s.tree, (uint32_t)s.level, (uint32_t)s.start, (uint32_t)attempt);
                kiss_seed(&rng, ns);
                create_split(a, items, (size_t)s.count, &rng, hv);
                a->split_rows += s.count; a->split_nodes++;
                cnt[0] = cnt[1] = 0;
                for (int p = 0; p < s.count; p++) {
                    float d = xdot(a, hv, a->X + (size_t)items[p] * f);
                    int sd = d != 0 ? (d > 0) : pos_flip(ns, (uint32_t)p);
                    side_of[p] = (uint8_t)sd; cnt[sd]++;
                }
                if (split_imbalance(cnt[0], cnt[1]) < 0.95) break;
            }
            for (int round = 0; split_imbalance(cnt[0], cnt[1]) > 0.99; round++) {
                for (int z = 0; z < f; z++) hv[z] = 0;
                cnt[0] = cnt[1] = 0;
                if (round >= 32) {   /* give up on chance: halve by position */
                    for (int p = 0; p < s.count; p++) { int sd = p >= s.count / 2; side_of[p] = (uint8_t)sd; cnt[sd]++; }
                    break;
                }
                ns = node_seed(a->seed, (uint32_t)s.tree, (uint32_t)s.level, (uint32_t)s.start, (uint32_t)(3 + round));
                for (int p = 0; p < s.count; p++) { int sd = pos_flip(ns, (uint32_t)p); side_of[p] = (uint8_t)sd; cnt[sd]++; }
            }
            /* stable partition: side 0 first */
            size_t w0 = 0, w1 = cnt[0];
            for (int p = 0; p < s.count; p++) tmp[side_of[p] ? w1++ : w0++] = items[p];
            memcpy(items, tmp, sizeof(int) * (size_t)s.count);
            m->kind = NODE_SPLIT; m->v = hv;
            for (int sd = 0; sd < 2; sd++) {
                int id = new_node(a);
                m = &a->nodes[s.node];             /* realloc may have moved nodes */
                m->child[sd] = id;
                nxt[n_nxt].tree = s.tree; nxt[n_nxt].level = s.level + 1;
                nxt[n_nxt].start = s.start + (sd ? (int)cnt[0] : 0);
                nxt[n_nxt].count = (int)cnt[sd];
                nxt[n_nxt].node = id;
                n_nxt++;
            }
        }
        free(cur);
        cur = nxt; n_cur = n_nxt;
    }
    free(cur); free(perm); free(tmp); free(side_of);
}

void annoyo_build(annoyo *a, int n_trees)
{
    if (a->mode) { build1(a, n_trees); return; }
    a->roots = (int *)malloc(sizeof(int) * n_trees);
    int *all = (int *)malloc(sizeof(int) * (size_t)(a->n_items ? a->n_items : 1));
    for (int i = 0; i < a->n_items; i++) all[i] = i;
    for (int t = 0; t < n_trees; t++) a->roots[a->n_roots++] = make_tree0(a, all, (size_t)a->n_items, 1);
    free(all);
}

/* ----------------------------------------------------------------- search */

typedef struct { float d; int id; } pqe;

static int pq_less(pqe x, pqe y) { return x.d < y.d || (x.d == y.d && x.id < y.id); }

static void pq_push(pqe **h, int *n, int *cap, pqe e)
{
    if (*n == *cap) { *cap = *cap ? *cap * 2 : 64; *h = (pqe *)realloc(*h, sizeof(pqe) * *cap); }
    int i = (*n)++;
    (*h)[i] = e;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!pq_less((*h)[p], (*h)[i])) break;
        pqe t = (*h)[p]; (*h)[p] = (*h)[i]; (*h)[i] = t;
        i = p;
    }
}
static pqe pq_pop(pqe *h, int *n)
{
    pqe top = h[0];
    h[0] = h[--(*n)];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && pq_less(h[m], h[l])) m = l;
        if (r < *n && pq_less(h[m], h[r])) m = r;
        if (m == i) break;
        pqe t = h[m]; h[m] = h[i]; h[i] = t;
        i = m;
    }
    return top;
}

typedef struct { float d; int id; } dist_id;
static int cmp_dist_id(const void *x, const void *y)
{
    const dist_id *a = (const dist_id *)x, *b = (const dist_id *)y;
    if (a->d < b->d) return -1;
    if (a->d > b->d) return 1;
    return (a->id > b->id) - (a->id < b->id);
}
static int cmp_int(const void *x, const void *y) { return (*(const int *)x > *(const int *)y) - (*(const int *)x < *(const int *)y); }

/*
 * _get_all_nns.  Returns the number of results (<= n); if cand_out is not NULL
 * it receives the number of unique candidates whose distance was evaluated.
 */
int annoyo_get_nns_by_vector(const annoyo *a, const float *v, int n, int search_k,
                             int *ids_out, float *dist_out, int *cand_out)
{
    const int f = a->f;
    pqe *heap = NULL; int hn = 0, hcap = 0;
    if (search_k == -1) search_k = n * a->n_roots;
    for (int i = 0; i < a->n_roots; i++) { pqe e = {INFINITY, a->roots[i]}; pq_push(&heap, &hn, &hcap, e); }
    int *nns = NULL; int nn = 0, ncap = 0;
    while (nn < search_k && hn > 0) {
        pqe top = pq_pop(heap, &hn);
        float d = top.d; int i = top.id;
        if (a->mode == 0 && i < a->n_items) {           /* a bare item */
            if (nn == ncap) { ncap = ncap ? ncap * 2 : 256; nns = (int *)realloc(nns, sizeof(int) * ncap); }
            nns[nn++] = i;
            continue;
        }
        const onode *nd = &a->nodes[a->mode ? i : i - a->n_items];
        if (nd->kind == NODE_LEAF) {
            int cnt = nd->child[0];
            if (nn + cnt > ncap) { ncap = (nn + cnt) * 2; nns = (int *)realloc(nns, sizeof(int) * ncap); }
            memcpy(nns + nn, nd->items, sizeof(int) * (size_t)cnt);
            nn += cnt;
        } else {
            float margin = xdot(a, nd->v, v);
            pqe e1 = {d < margin ? d : margin, nd->child[1]};
            pqe e0 = {d < -margin ? d : -margin, nd->child[0]};
            pq_push(&heap, &hn, &hcap, e1);
            pq_push(&heap, &hn, &hcap, e0);
        }
    }
    qsort(nns, (size_t)nn, sizeof(int), cmp_int);
    dist_id *nd = (dist_id *)malloc(sizeof(dist_id) * (size_t)(nn ? nn : 1));
    int m = 0, last = -1;
    float vv = xdot(a, v, v);
    for (int i = 0; i < nn; i++) {
        int j = nns[i];
        if (j == last) continue;
        last = j;
        nd[m].d = ang_dist(vv, a->norm2[j], xdot(a, v, a->X + (size_t)j * f));
        nd[m].id = j;
        m++;
    }
    qsort(nd, (size_t)m, sizeof(dist_id), cmp_dist_id);
    int p = n < m ? n : m;
    for (int i = 0; i < p; i++) {
        ids_out[i] = nd[i].id;
        if (dist_out) dist_out[i] = sqrtf(nd[i].d > 0 ? nd[i].d : 0);
    }
    if (cand_out) *cand_out = m;
    free(nd); free(nns); free(heap);
    return p;
}

int annoyo_get_nns_by_item(const annoyo *a, int item, int n, int search_k,
                           int *ids_out, float *dist_out, int *cand_out)
{
    return annoyo_get_nns_by_vector(a, a->X + (size_t)item * a->f, n, search_k, ids_out, dist_out, cand_out);
}

/* ------------------------------------------------- introspection for tests */

int annoyo_n_nodes(const annoyo *a) { return a->n_nodes; }
int annoyo_n_roots(const annoyo *a) { return a->n_roots; }
int annoyo_root(const annoyo *a, int t) { return a->roots[t]; }
int64_t annoyo_split_rows(const annoyo *a) { return a->split_rows; }
int64_t annoyo_split_nodes(const annoyo *a) { return a->split_nodes; }
float annoyo_norm2(const annoyo *a, int i) { return a->norm2[i]; }

/* node record: kind, n_desc, child0 (leaf: item count), child1, tree, level */
void annoyo_node_info(const annoyo *a, int id, int *out6)
{
    const onode *m = &a->nodes[id];
    out6[0] = m->kind; out6[1] = m->n_desc; out6[2] = m->child[0]; out6[3] = m->child[1];
    out6[4] = m->tree; out6[5] = m->level;
}
void annoyo_node_vector(const annoyo *a, int id, float *out) { memcpy(out, a->nodes[id].v, sizeof(float) * a->f); }
void annoyo_node_items(const annoyo *a, int id, int *out) { memcpy(out, a->nodes[id].items, sizeof(int) * (size_t)a->nodes[id].child[0]); }

float annoyo_dot(int mode, const float *x, const float *y, int f) { return mode ? dot_canon(x, y, f) : dot_seq(x, y, f); }
