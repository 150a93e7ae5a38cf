"""Full-size and edge-case checks of the HIP path through size-independent
properties (the oracle is too slow at BASELINE.json's sizes).  -m gpu

  * forest: every tree is a permutation of the items, leaves hold <= K = D + 2 ids,
    children partition their parent, hyperplanes are unit vectors (or zero after the
    random fallback), split counts add up;
  * approximate search: every returned distance is the true angular distance of the
    returned id; results are sorted by (distance, id); recall against the exact search;
  * features: exactly linear under power-of-two scaling of the coverages; invariant
    under a permutation of the junction lines that keeps each column's line order;
  * exact search: agrees with the approximate search on distances of common ids and
    with numpy's fp64 ordering away from ties.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check_forest(a, N, T, K):
    f = a.get_forest()
    st = a.forest_stats()
    rec, perm = f["node_rec"], f["perm"]
    assert perm.shape == (T, N)
    assert (np.sort(perm, axis=1) == np.arange(N)[None, :]).all()              # each tree: a permutation
    leaves = rec[rec[:, 0] == 1]
    splits = np.nonzero(rec[:, 0] == 0)[0]
    assert (leaves[:, 3] <= K).all() and (leaves[:, 3] >= 1).all()
    assert st["n_nodes"] == len(rec) and st["n_split"] == len(splits) and st["n_leaves"] == len(leaves)
    for tree in range(T):                                                      # leaves tile [0, N) of their tree
        lt = leaves[leaves[:, 1] == tree]
        order = np.argsort(lt[:, 2])
        starts, counts = lt[order, 2], lt[order, 3]
        assert starts[0] == 0 and (starts[1:] == (starts + counts)[:-1]).all() and starts[-1] + counts[-1] == N
    c0, c1 = rec[splits, 4], rec[splits, 5]
    assert (rec[c0, 3] + rec[c1, 3] == rec[splits, 3]).all()                   # children partition the parent
    assert (rec[c0, 2] == rec[splits, 2]).all() and (rec[c1, 2] == rec[splits, 2] + rec[c0, 3]).all()
    assert (rec[c0, 3] > 0).all() and (rec[c1, 3] > 0).all()
    assert (rec[splits, 3] > K).all()
    hn = np.linalg.norm(f["hyperplanes"].astype(np.float64), axis=1)
    assert np.all((np.abs(hn - 1.0) < 1e-4) | (hn == 0.0))
    # segments of sorted ids: stable partitions of the identity keep each node's list ascending
    for nid in leaves[:50, :]:
        seg = perm[nid[1], nid[2]:nid[2] + nid[3]]
        assert (np.diff(seg) > 0).all()
    return st


def _true_ang(X, q, ids):
    X64, q64 = X[ids].astype(np.float64), q.astype(np.float64)
    pq = X64 @ q64
    ppqq = (X64 * X64).sum(1) * (q64 @ q64)
    d2 = np.where(ppqq > 0, 2.0 - 2.0 * pq / np.sqrt(np.where(ppqq > 0, ppqq, 1.0)), 2.0)
    return np.sqrt(np.maximum(d2, 0))


@pytest.mark.parametrize("N,D,T", [(10_000, 3000, 200), (9_000, 8192, 4)])
def test_full_size_forest_and_search_properties(N, D, T):
    """BASELINE.json configs[1] (10k x 3000, 200 trees) and the D = 8192 row length of configs[4]
    (LDS two_means kernel, 32 k-steps per row)."""
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(8675309)
    centers = rng.standard_normal((64, D)).astype(np.float32)
    X = centers[rng.integers(0, 64, N)] * (rng.random((N, D), dtype=np.float32) < 0.3)
    X += 0.05 * rng.standard_normal((N, D), dtype=np.float32)
    a = AnnoyIndex(D)
    a.add_items(X)
    a.build(T)
    K = D + 2
    st = _check_forest(a, N, T, K)
    assert st["split_rows"] >= st["split_attempts"] * (K + 1)
    items = rng.choice(N, 200, replace=False).astype(np.int32)
    for sk in (100, -1):
        ids, d, cnt = a.get_nns_by_item_batch(items, 20, sk)
        assert (cnt == 20).all()
        for qi, it in enumerate(items):
            assert ids[qi, 0] == it and d[qi, 0] < 1e-3                        # an item is its own nearest neighbour
            assert np.allclose(d[qi], _true_ang(X, X[it], ids[qi]), atol=2e-5)
            # ranked by (2 - 2 cos, id) as annoy does; the square root that is reported can make two different keys
            # print equal: ascending distances, distinct ids
            assert (np.diff(d[qi]) >= 0).all() and len(set(ids[qi].tolist())) == 20
    eids, ed, _ = a.exact_search_batch(X[items].astype(np.float64), 20)
    rec = np.mean([len(set(ids[i].tolist()) & set(eids[i].tolist())) / 20.0 for i in range(len(items))])
    assert rec > 0.9                                                           # clustered data, search_k = 20 * T
    for qi in range(len(items)):                                               # exact: fp64 ordering
        assert (np.diff(ed[qi]) >= 0).all()
        # compare 2 - 2cos (the square): near 0 the square root turns 1e-16 of summation-order noise into 1e-8
        assert np.allclose(ed[qi] ** 2, _true_ang(X, X[items[qi]], eids[qi]) ** 2, atol=1e-12)


def test_many_items_tiny_rows_deep_trees():
    """530k items x 4 features: K = 6, so trees are ~17 levels deep with ~10^5 nodes each, and the
    sample bitmap of the query kernel no longer fits LDS (global-memory bitmap path).  Checked
    against the oracle's wave-order restatement on a handful of queries, plus the invariants."""
    from morna_amd.annoy import AnnoyIndex
    from oracle import capi
    rng = np.random.default_rng(7)
    N, D, T = 530_000, 4, 2
    X = rng.standard_normal((N, D)).astype(np.float32)
    a = AnnoyIndex(D)
    a.add_items(X)
    a.build(T)
    st = _check_forest(a, N, T, D + 2)
    assert st["max_depth"] >= 15
    o = capi.AnnoyOracle(D, mode=1)
    o.set_items(X)
    o.build(T)
    assert st["split_rows"] == o.split_rows() and st["split_attempts"] == o.split_nodes()
    assert st["n_nodes"] == o.n_nodes()
    items = rng.choice(N, 24, replace=False).astype(np.int32)
    for n, sk in ((10, -1), (10, 5000), (100, 20000)):
        ids, d, cnt = a.get_nns_by_item_batch(items, n, sk)
        for qi, it in enumerate(items):
            rid, rd = o.get_nns_by_item(int(it), n, sk, include_distances=True)
            assert ids[qi, :int(cnt[qi])].tolist() == rid
            assert d[qi, :int(cnt[qi])].tobytes() == np.array(rd, np.float32).tobytes()


def test_feature_build_linearity_and_order_invariance():
    """A 20 000-sample, 6 000-line synthetic intropolis at D = 3000: size-independent properties of the feature build."""
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.index import prepare_csr
    from morna_amd.synth import synthetic_intropolis
    data = synthetic_intropolis(20_000, J=6_000)
    prep = prepare_csr(data["keys"], data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100)
    D = 3000

    def build(ids, cov, row_ptr, key_bytes, key_off, idf):
        a = AnnoyIndex(D)
        a.stage_junctions(key_bytes, key_off, row_ptr, ids, cov, idf)
        a.build_features(prep["n_items"])
        return a.get_items()
    base = build(prep["ids"], prep["cov"], prep["row_ptr"], prep["key_bytes"], prep["key_off"], prep["idf"])
    assert np.isfinite(base).all() and (np.abs(base).sum(1) > 0).all()
    # exact linearity: coverages x 4 -> every fp64 partial sum x 4 -> every fp32 cell x 4
    x4 = build(prep["ids"], prep["cov"] * 4, prep["row_ptr"], prep["key_bytes"], prep["key_off"], prep["idf"])
    assert (x4 == base * 4.0).all()
    # a sample listed in another position of the same line lands in the same cell: reverse every line
    rp = prep["row_ptr"]
    rev_ids, rev_cov = prep["ids"].copy(), prep["cov"].copy()
    for j in range(len(rp) - 1):
        rev_ids[rp[j]:rp[j + 1]] = prep["ids"][rp[j]:rp[j + 1]][::-1]
        rev_cov[rp[j]:rp[j + 1]] = prep["cov"][rp[j]:rp[j + 1]][::-1]
    assert build(rev_ids, rev_cov, rp, prep["key_bytes"], prep["key_off"], prep["idf"]).tobytes() == base.tobytes()
    # signed feature hashing roughly preserves norms (reference tests/test_norm_estimator.py)
    nz = (base != 0).sum(1)
    assert nz.min() > 100


def test_edge_cases_small_and_ragged():
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(3)
    # k larger than the index, search_k = 0 / 1 / huge, a single item, two items
    for N in (1, 2, 5):
        X = rng.standard_normal((N, 7)).astype(np.float32)
        a = AnnoyIndex(7)
        a.add_items(X)
        a.build(3)
        ids, d, cnt = a.get_nns_by_item_batch(np.zeros(1, np.int32), 10, -1)
        assert cnt[0] == N and sorted(ids[0, :N].tolist()) == list(range(N)) and (ids[0, N:] == -1).all()
        assert a.get_nns_by_item(0, 10, 0) == []                               # search_k = 0: nothing inspected
        assert len(a.get_nns_by_item(0, 10, 10**6)) == N
        eids, ed, ecnt = a.exact_search_batch(X[:1].astype(np.float64), 10)
        assert ecnt[0] == N and eids[0, 0] == 0
    # empty query batch, empty stage
    a = AnnoyIndex(16)
    a.add_items(rng.standard_normal((40, 16)).astype(np.float32))
    a.build(2)
    ids, d, cnt = a.get_nns_by_vector_batch(np.zeros((0, 16), np.float32), 5)
    assert ids.shape == (0, 5)
    with pytest.raises(IndexError):
        a.get_nns_by_vector([0.0] * 15, 5)
    b = AnnoyIndex(16)
    b.stage_junctions(np.zeros(0, np.uint8), np.zeros(1, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int32),
                      np.zeros(0, np.int32), np.zeros(0, np.float64))
    with pytest.raises(ValueError):
        b.build_features(0)                                                    # morna.py:399-403
    # a junction line shared by every sample gives idf = log(1) = 0: all-zero rows are legal items
    b.stage_junctions(np.frombuffer(b"chr1 1 2", np.uint8), np.array([0, 8]), np.array([0, 3]), np.array([0, 1, 2], np.int32),
                      np.array([5, 6, 7], np.int32), np.array([0.0]))
    b.build_features(3)
    assert (b.get_items() == 0).all()
    b.build(2)
    assert sorted(b.get_nns_by_item(1, 3, -1)) == [0, 1, 2]


_FOREST_DIGEST = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, {root!r})
from morna_amd.annoy import AnnoyIndex
rng = np.random.default_rng(20261004)
N, D, T = 30000, 700, 24
C = rng.standard_normal((40, D)).astype(np.float32)
X = (C[rng.integers(0, 40, N)] * rng.uniform(1e-3, 1e3, (N, 1)).astype(np.float32)
     + 0.4 * rng.standard_normal((N, D)).astype(np.float32)).astype(np.float32)
X[5] = 0.0
X[7] = X[6]
X[11] *= np.float32(1e-30)          # a row far below the fp16 range after scaling fails: its own scale handles it
# 600 rows that differ from row 50 in the 4th digit: for queries among them, hundreds of candidates lie within
# the filter's 2 * delta of the k-th distance and must all reach the fp32 ranking
X[1000:1600] = X[50] * (1.0 + 1e-4 * rng.standard_normal((600, D))).astype(np.float32)
X[1600:1700] = X[50] * np.float32(3.0)                      # exact scaled duplicates: ties broken by id
X[13, 0] = np.float32(np.inf)                               # a row with no fp16 image at all
a = AnnoyIndex(D)
a.add_items(X)
a.build(T)
f = a.get_forest()
h = hashlib.sha256()
for k in ("perm", "node_rec", "hyperplanes", "hp_node"):
    h.update(np.ascontiguousarray(f[k]).tobytes())
items = np.concatenate([np.arange(64), np.arange(1000, 1064), np.arange(1600, 1616), [13]]).astype(np.int32)
for n, sk in ((10, -1), (40, 100), (3, 5000)):
    ids, d, cnt = a.get_nns_by_item_batch(items, n, sk)
    h.update(ids.tobytes()); h.update(d.tobytes()); h.update(cnt.tobytes())
print("DIGEST", h.hexdigest(), a.forest_stats()["n_split"], a.forest_stats()["max_depth"])
"""


def test_filters_do_not_change_results(tmp_path):
    """The fp16 MFMA split filter + exact fallback (splitmm.hip) must write exactly the sides of the plain fp32
    kernels, and the fp16 candidate filter of the approximate search must return exactly what the all-fp32
    search returns; likewise the four-wave strip form of two_means and the one-wave form: the same seeded build
    and searches in four processes -- defaults, MORNA_SPLIT_MM=0 (row-window / chunk forms), MORNA_QUERY_FILTER=0,
    MORNA_TM_STRIP=0 -- on rows spanning six orders of magnitude in norm, a zero
    row, duplicates, near-duplicates in the 4th digit, exact scaled duplicates, a row of denormal scale and a row
    holding an infinity."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = str(tmp_path / "digest.py")
    with open(script, "w") as fh:
        fh.write(_FOREST_DIGEST.format(root=root))
    out = {}
    for name, extra in (("default", {}), ("no_mm", {"MORNA_SPLIT_MM": "0"}), ("no_qf", {"MORNA_QUERY_FILTER": "0"}),
                        ("no_strip", {"MORNA_TM_STRIP": "0"}), ("no_dense", {"MORNA_QUERY_DENSE": "0"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0].split()
        out[name] = line[1:]
    assert out["default"] == out["no_mm"] == out["no_qf"] == out["no_strip"] == out["no_dense"], out
    assert int(out["default"][2]) >= 4                                      # deep enough for every form to run
