/* A C caller of libmorna_hip.so with nothing but include/morna_hip.h: the drop-in boundary as a C program would use it
 * (no Python, no torch).  Exercises the AnnoyIndex-shaped entry points (morna.py:166, 406, 425, 439, 544, 651, 762, 702, 1174),
 * mmh3.hash (morna.py:369), the exact search (morna.py:681-716) with vectors and with stored rows, and the row-sharded
 * search through a communicator made from the C ABI alone on one rank.  Prints "abi caller ok" and
 * exits 0; any mismatch exits 1 with a message.  Built and run by tests/test_gpu_abi_c.py.  */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "morna_hip.h"

#define CK(call)                                                                  \
    do {                                                                          \
        int rc_ = (call);                                                         \
        if (rc_ != MORNA_OK) {                                                    \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, morna_last_error());    \
            return 1;                                                             \
        }                                                                         \
    } while (0)
#define EXPECT(cond)                                              \
    do {                                                          \
        if (!(cond)) {                                            \
            fprintf(stderr, "line %d: %s\n", __LINE__, #cond);    \
            return 1;                                             \
        }                                                         \
    } while (0)

static unsigned long long rng_state = 88172645463325252ull;
static double rnd(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0 - 0.5;
}

int main(int argc, char **argv)
{
    enum { F = 72, N = 3000, T = 8, K = 10, NQ = 40 };
    const char *path = argc > 1 ? argv[1] : "/tmp/abi_caller.annoy.mor";
    /* mmh3.hash("foo") = -156908512, mmh3's documented answer */
    EXPECT(morna_hash32((const uint8_t *)"foo", 3) == -156908512);
    morna_index *h = NULL, *g = NULL;
    int32_t n_dev = 0;
    CK(morna_device_count(&n_dev));
    EXPECT(n_dev >= 1);
    EXPECT(morna_index_create(F, n_dev, &h) == MORNA_E_INVALID);   /* a device that is not there */
    CK(morna_index_create(F, 0, &h));
    static double row[F];
    static float centers[12][F];
    for (int c = 0; c < 12; c++)
        for (int z = 0; z < F; z++) centers[c][z] = (float)rnd();
    for (int i = 0; i < N; i++) {
        const int c = (int)((rnd() + 0.5) * 12) % 12;
        for (int z = 0; z < F; z++) row[z] = centers[c][z] + 0.3 * rnd();
        CK(morna_add_item(h, i, row));
    }
    EXPECT(morna_get_n_items(h) == N);
    CK(morna_build(h, T, 0));
    EXPECT(morna_get_n_trees(h) == T);
    EXPECT(morna_add_item(h, N, row) == MORNA_E_STATE); /* "You can't add an item to a built index" */
    morna_forest_stats st;
    CK(morna_get_forest_stats(h, &st));
    EXPECT(st.n_items == N && st.leaf_capacity == F + 2 && st.n_nodes == st.n_trees + 2 * st.n_split);

    static int32_t items[NQ], ids[NQ][K], ids2[NQ][K], cnt[NQ], cnt2[NQ];
    static float dist[NQ][K], dist2[NQ][K], q[NQ][F];
    static double qd[NQ][F], ed[NQ][K], ed2[NQ][K];
    for (int i = 0; i < NQ; i++) items[i] = (i * 71) % N;
    CK(morna_get_nns_by_item(h, items, NQ, K, -1, &ids[0][0], &dist[0][0], cnt));
    CK(morna_get_item_vectors(h, items, NQ, &q[0][0]));
    CK(morna_get_nns_by_vector(h, &q[0][0], NQ, K, -1, &ids2[0][0], &dist2[0][0], cnt2));
    for (int i = 0; i < NQ; i++) {
        EXPECT(cnt[i] == K && ids[i][0] == items[i] && dist[i][0] < 1e-3f); /* a row is its own nearest neighbour */
        for (int r = 1; r < K; r++) EXPECT(dist[i][r] >= dist[i][r - 1]);
    }
    EXPECT(memcmp(ids, ids2, sizeof(ids)) == 0 && memcmp(dist, dist2, sizeof(dist)) == 0); /* by item = by its stored vector */

    /* exact search: stored rows as queries == the same rows handed over as fp64 vectors */
    for (int i = 0; i < NQ; i++)
        for (int z = 0; z < F; z++) qd[i][z] = (double)q[i][z];
    CK(morna_exact_search(h, &qd[0][0], NQ, K, &ids[0][0], &ed[0][0], cnt));
    CK(morna_exact_search_by_item(h, items, NQ, K, &ids2[0][0], &ed2[0][0], cnt2));
    EXPECT(memcmp(ids, ids2, sizeof(ids)) == 0 && memcmp(ed, ed2, sizeof(ed)) == 0 && memcmp(cnt, cnt2, sizeof(cnt)) == 0);
    for (int i = 0; i < NQ; i++) EXPECT(cnt[i] == K && ids[i][0] == items[i] && ed[i][0] == 0.0);

    /* the row-sharded search on a communicator of one rank: same answers, ids global (= local here) */
    uint8_t uid[MORNA_COMM_ID_BYTES];
    int32_t rank = -1, world = -1;
    int64_t off[2] = {-1, -1}, n_each[1] = {NQ};
    EXPECT(morna_get_nns_by_vector_sharded(h, &q[0][0], NQ, K, -1, &ids2[0][0], &dist2[0][0], cnt2) == MORNA_E_STATE);
    CK(morna_comm_unique_id(uid));
    CK(morna_comm_init(h, uid, 0, 1));
    CK(morna_comm_info(h, &rank, &world, off));
    EXPECT(rank == 0 && world == 1 && off[0] == 0 && off[1] == N);
    CK(morna_exact_search_by_item_sharded(h, items, NQ, n_each, K, &ids2[0][0], &ed2[0][0], cnt2));
    EXPECT(memcmp(ids, ids2, sizeof(ids)) == 0 && memcmp(ed, ed2, sizeof(ed)) == 0);
    CK(morna_get_nns_by_item(h, items, NQ, K, 50, &ids[0][0], &dist[0][0], cnt));
    CK(morna_get_nns_by_item_sharded(h, items, NQ, n_each, K, 50, &ids2[0][0], &dist2[0][0], cnt2));
    EXPECT(memcmp(ids, ids2, sizeof(ids)) == 0 && memcmp(dist, dist2, sizeof(dist)) == 0);
    CK(morna_comm_destroy(h));

    /* save -> load into a second handle: same items, same forest, same answers */
    CK(morna_save(h, path));
    CK(morna_index_create(F, 0, &g));
    CK(morna_load(g, path));
    EXPECT(morna_get_n_items(g) == N && morna_get_n_trees(g) == T);
    static float v1[F], v2[F];
    CK(morna_get_item_vector(h, 1234, v1));
    CK(morna_get_item_vector(g, 1234, v2));
    EXPECT(memcmp(v1, v2, sizeof(v1)) == 0);
    EXPECT(morna_get_item_vector(g, N, v2) == MORNA_E_RANGE);
    CK(morna_get_nns_by_item(g, items, NQ, K, 50, &ids2[0][0], &dist2[0][0], cnt2));
    EXPECT(memcmp(ids, ids2, sizeof(ids)) == 0 && memcmp(dist, dist2, sizeof(dist)) == 0);
    EXPECT(morna_load(g, "/nonexistent/file") == MORNA_E_IO);
    CK(morna_index_destroy(g));
    CK(morna_index_destroy(h));
    remove(path);
    printf("abi caller ok\n");
    return 0;
}
