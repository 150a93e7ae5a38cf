"""Parity of the HIP path (through the C ABI) with the CPU oracle.  -m gpu

Bars: bit-exact for hashes, the fp32 feature matrix, exact-search ids AND fp64
distances, forest structure / hyperplanes / approximate results against oracle
mode 1 (same algorithm, same seeds, same summation order); stated tolerances
only where a test says so.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, angular64, assert_tie_aware_order

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from oracle import capi as c
    c.lib()
    return c


@pytest.fixture(scope="module")
def morna_ref():
    from oracle import morna_ref as m
    return m


def _tokenized(morna_ref, lines):
    keys, rp, s, c = [], [0], [], []
    for ln in lines:
        k, ss, cc = morna_ref.tokenize_line(ln)
        keys.append(k)
        s += ss
        c += cc
        rp.append(len(s))
    return keys, np.array(rp, np.int64), np.array(s, np.int64), np.array(c, np.int64)


def _gpu_features(keys, rp, s, c, sample_count, threshold, D, order=None):
    """order: None -> no item order staged (tile-by-tile accumulation); "ext" -> the external sample ids (the order the
    host paths hand over: lines listed by ascending sample id are then read once, the others take the flagged paths);
    "reversed" -> an order in which NO ascending line ascends (every line goes the flagged way)."""
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.index import prepare_csr
    prep = prepare_csr(keys, rp, s, c, sample_count, threshold)
    a = AnnoyIndex(D)
    a.stage_junctions(prep["key_bytes"], prep["key_off"], prep["row_ptr"], prep["ids"], prep["cov"], prep["idf"])
    if order == "ext":
        a.stage_item_order(prep["ext_ids"])
    elif order == "reversed":
        a.stage_item_order(-np.asarray(prep["ext_ids"], np.int64))
    a.build_features(prep["n_items"])
    return a, prep


# --------------------------------------------------------------------- hashing

def test_device_hash_golden():
    from morna_amd.annoy import AnnoyIndex
    from oracle.capi import pack_keys
    with open(os.path.join(GOLDEN, "murmur3_vectors.json")) as fh:
        g = json.load(fh)
    keys = [k for k, _ in g["vectors"] if k]
    buf, off = pack_keys(keys)
    for D in (128, 3000):
        a = AnnoyIndex(D)
        h, col, sign = a.hash_keys(buf, off)
        want = np.array([hv for k, hv in g["vectors"] if k], np.int64)
        assert h.astype(np.int64).tolist() == want.tolist()
        assert col.tolist() == [int(x) % D for x in want]            # Python floored modulo
        assert sign.tolist() == [-1 if x < 0 else 1 for x in want]


# -------------------------------------------------------------------- features

@pytest.mark.parametrize("name", ["simple", "lossy", "lose_sample"])
@pytest.mark.parametrize("D", [40, 128, 3000])
def test_features_embedded_fixtures(embedded, embedded_mats, morna_ref, name, D):
    spec = embedded["expected"][name]
    keys, rp, s, c = _tokenized(morna_ref, embedded[spec["input"]])
    a, prep = _gpu_features(keys, rp, s, c, spec["sample_count"], spec["sample_threshold"], D)
    assert prep["n_items"] == spec["n_items"] == a.get_n_items()
    X = a.get_items()
    assert X.tobytes() == embedded_mats["%s_D%d_f32" % (name, D)].tobytes()
    b, _ = _gpu_features(keys, rp, s, c, spec["sample_count"], spec["sample_threshold"], D, "ext")   # the lossy fixture lists
    assert b.get_items().tobytes() == X.tobytes()                                                    # its samples descending
    assert prep["ext_ids"].tolist() == embedded_mats["%s_D%d_ext_ids" % (name, D)].tolist()


def test_features_tiny_intropolis(morna_ref):
    g = np.load(os.path.join(GOLDEN, "tiny_intropolis_D128.npz"))
    with open(os.path.join(GOLDEN, "tiny_intropolis.tsv")) as fh:
        lines = fh.readlines()
    keys, rp, s, c = _tokenized(morna_ref, lines)
    a, prep = _gpu_features(keys, rp, s, c, 6850, 100, 128, "ext")
    assert a.get_items().tobytes() == g["X"].tobytes()
    assert _gpu_features(keys, rp, s, c, 6850, 100, 128)[0].get_items().tobytes() == g["X"].tobytes()
    assert prep["ext_ids"].tolist() == g["ext_ids"].tolist()


def _synthetic_lines(rng, n_samples, J, lo, hi, dup_keys=True, weird=True):
    keys, rp, s, c = [], [0], [], []
    for j in range(J):
        if dup_keys and j > 10 and rng.random() < 0.05:
            keys.append(keys[int(rng.integers(0, j))])              # duplicate key: cumulative freq
        else:
            keys.append("chr%d %d %d" % (rng.integers(1, 23), rng.integers(10**4, 2 * 10**8), rng.integers(10**4, 2 * 10**8)))
        n = int(rng.integers(lo, hi))
        ids = np.sort(rng.choice(n_samples, size=min(n, n_samples), replace=False)) + 1
        if weird:
            r = rng.random()
            if r < 0.1:
                ids = ids[::-1]                                     # descending, as the lossy fixture
            elif r < 0.2:
                ids = rng.permutation(ids)                          # unordered -> serial path
            elif r < 0.25 and len(ids) > 4:
                ids = np.concatenate([ids, ids[:3]])                # a sample repeated inside a line
        s += ids.tolist()
        c += rng.integers(1, 200, size=len(ids)).tolist()
        rp.append(len(s))
    return keys, np.array(rp, np.int64), np.array(s, np.int64), np.array(c, np.int64)


@pytest.mark.parametrize("D,n_samples,J", [(64, 700, 1500), (3000, 2000, 3000), (257, 300, 50),
                                           (40, 70000, 4500)])   # > 65536 samples: the duplicate flags of the lines
                                                                 # come from the workgroup-per-line kernel
@pytest.mark.parametrize("order", [None, "ext", "reversed"])
def test_features_synthetic_vs_oracle(capi, D, n_samples, J, order):
    rng = np.random.default_rng(8675309 + D)
    keys, rp, s, c = _synthetic_lines(rng, n_samples, J, 5, 120)
    buf, off = capi.pack_keys(keys)
    ref = capi.index_features(buf, off, rp, s, c, n_samples, 20, D)
    a, prep = _gpu_features(keys, rp, s, c, n_samples, 20, D, order)
    assert prep["n_items"] == ref["n_items"]
    assert n_samples < 65536 or prep["n_items"] > 65536
    assert prep["ext_ids"].tolist() == ref["ext_ids"].tolist()
    assert prep["skipped"] == ref["skipped"]
    assert a.get_items().tobytes() == ref["X"].tobytes()


def test_features_long_lines_with_item_order(capi):
    """Lines of up to 9000 samples (several 2048-entry pieces per wave in line_prep_kernel, pieces of a tile longer
    than one wave in accumulate_wave_kernel), 5000 samples = 3 tiles of the item order, the last one ragged."""
    rng = np.random.default_rng(31)
    n_samples, J, D = 9500, 260, 96
    keys, rp, s, c = _synthetic_lines(rng, n_samples, J, 40, 9000, dup_keys=True, weird=True)
    buf, off = capi.pack_keys(keys)
    ref = capi.index_features(buf, off, rp, s, c, n_samples, 20, D)
    for order in ("ext", None):
        a, prep = _gpu_features(keys, rp, s, c, n_samples, 20, D, order)
        assert a.get_items().tobytes() == ref["X"].tobytes(), order


def test_row_norms_canonical(capi):
    rng = np.random.default_rng(1)
    for D in (40, 256, 3000):
        X = rng.standard_normal((37, D)).astype(np.float32)
        from morna_amd.annoy import AnnoyIndex
        a = AnnoyIndex(D)
        a.add_items(X)
        n2 = a.get_norms2()
        want = np.array([capi.dot(1, x, x) for x in X], np.float32)
        assert n2.tobytes() == want.tobytes()


def test_add_item_roundtrip_and_errors():
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(2)
    a = AnnoyIndex(24)
    V = rng.standard_normal((5, 24))
    for i in range(5):
        a.add_item(i, V[i].tolist())
    assert a.get_n_items() == 5
    for i in range(5):
        assert a.get_item_vector(i) == [float(np.float32(x)) for x in V[i]]
    with pytest.raises(IndexError):
        a.get_item_vector(5)
    with pytest.raises(RuntimeError):
        a.get_nns_by_item(0, 3)          # not built
    a.build(3)
    with pytest.raises(RuntimeError):
        a.add_item(5, V[0].tolist())     # annoy: can't add to a built index
    with pytest.raises(IndexError):
        a.get_nns_by_item(9, 3)
    e = AnnoyIndex(8)
    with pytest.raises(ValueError):
        e.build(2)                       # no items


# ---------------------------------------------------------------- exact search

def test_exact_search_golden():
    from morna_amd.annoy import AnnoyIndex
    g = np.load(os.path.join(GOLDEN, "exact_search_512x128.npz"))
    a = AnnoyIndex(128)
    a.add_items(g["X"])
    ids, d, cnt = a.exact_search_batch(g["Q"], 20)
    assert cnt.tolist() == [20] * len(g["Q"])
    assert ids.astype(np.int64).tolist() == g["ids"].tolist()
    assert d.tobytes() == g["dists"].tobytes()


@pytest.mark.parametrize("N,D,k", [(3000, 300, 20), (1500, 3000, 10), (40, 16, 64)])
def test_exact_search_vs_oracle(capi, N, D, k):
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(N + D)
    X = (rng.standard_normal((N, D)) * (rng.random((N, D)) < 0.2)).astype(np.float32)
    X[N // 2] = X[3]
    X[N // 2 + 1] = 2.0 * X[3]        # exact scaling: cos is exactly 1, no negative radicand
    Q = rng.standard_normal((9, D))
    Q[0] = X[3]
    Q[1] = X[5] * 0.5
    a = AnnoyIndex(D)
    a.add_items(X)
    ids, d, cnt = a.exact_search_batch(Q, k)
    for qi in range(len(Q)):
        rid, rd = capi.exact_search(X, Q[qi], k)
        m = int(cnt[qi])
        assert m == len(rid)
        assert ids[qi, :m].astype(np.int64).tolist() == rid.tolist()
        assert d[qi, :m].tobytes() == rd.tobytes()


# ------------------------------------------------------------ forest / approx

@pytest.mark.parametrize("name", ["simple", "lossy", "lose_sample"])
@pytest.mark.parametrize("D", [128, 3000])
def test_embedded_orderings_gpu(embedded, embedded_mats, name, D):
    """The reference's own known answers (morna.py:1176-1187, 1267-1278, 1312-1323):
    20 trees, get_nns_by_item(i, 10, search_k=100)."""
    from morna_amd.annoy import AnnoyIndex
    spec = embedded["expected"][name]
    X = embedded_mats["%s_D%d_f32" % (name, D)]
    a = AnnoyIndex(D)
    for i in range(X.shape[0]):
        a.add_item(i, [float(x) for x in embedded_mats["%s_D%d_f64" % (name, D)][i]])
    a.build(20)
    assert a.get_n_items() == spec["n_items"]
    for i, exp in enumerate(spec["orderings"]):
        got = a.get_nns_by_item(i, 10, 100, include_distances=False)
        assert_tie_aware_order(got, exp, angular64(X, i), tol=2e-6)


def _clustered(rng, N, f, nc=12, noise=0.3):
    centers = rng.standard_normal((nc, f))
    return (centers[rng.integers(0, nc, N)] + noise * rng.standard_normal((N, f))).astype(np.float32)


def _compare_forest(a, o, N, T):
    """GPU forest vs oracle mode 1, node for node."""
    f = a.get_forest()
    rec = f["node_rec"]
    assert o.n_nodes() == rec.shape[0]
    hp_of = {int(n): i for i, n in enumerate(f["hp_node"])}
    for nid in range(rec.shape[0]):
        nd = o.node(nid)
        kind, tree, start, count, c0, c1 = [int(x) for x in rec[nid]]
        assert kind == nd["kind"], nid
        assert tree == nd["tree"], nid
        assert count == nd["n_desc"], nid
        if kind == 1:
            assert f["perm"][tree, start:start + count].tolist() == nd["items"].tolist(), nid
        else:
            assert (c0, c1) == (nd["child0"], nd["child1"]), nid
            assert f["hyperplanes"][hp_of[nid]].tobytes() == nd["v"].tobytes(), nid


@pytest.mark.parametrize("f,N,T", [(16, 3000, 8), (40, 5000, 5), (300, 4000, 3),
                                   (1000, 3000, 3), (2000, 4500, 2),   # 4 and 8 float4 per lane: four-wave two_means
                                   (3000, 8000, 11),    # the bench's row width: register two_means + row-window split,
                                                        # 11 trees = one full and one partial tree group
                                   (5000, 6000, 2),     # 20 float4 per lane: strip two_means only (too long for one wave)
                                   (8192, 9000, 2)])    # config 5's width: strip two_means with 32 float4 per lane
def test_forest_bit_exact_vs_oracle_wave_order(capi, f, N, T):
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(8675309 + f)
    X = _clustered(rng, N, f)
    X[17] = 0.0                       # a zero row: norm == 0 paths, margin == 0 coin flips
    X[18] = X[19]
    o = capi.AnnoyOracle(f, mode=1)
    o.set_items(X)
    o.build(T)
    a = AnnoyIndex(f)
    a.add_items(X)
    a.build(T)
    st = a.forest_stats()
    assert st["split_rows"] == o.split_rows() and st["split_attempts"] == o.split_nodes()
    _compare_forest(a, o, N, T)
    # approximate search: same candidates, same order, same fp32 distances
    items = np.arange(0, 120 if f < 1000 else 24, dtype=np.int32)      # the oracle is one thread
    for n, sk in ((10, -1), (20, 100), (5, 1), (50, 3000)):
        ids, d, cnt = a.get_nns_by_item_batch(items, n, sk)
        for qi, it in enumerate(items):
            (rid, rd) = o.get_nns_by_item(int(it), n, sk, include_distances=True)
            m = int(cnt[qi])
            assert ids[qi, :m].tolist() == rid, (it, n, sk)
            assert d[qi, :m].tobytes() == np.array(rd, np.float32).tobytes(), (it, n, sk)
    Q = rng.standard_normal((16, f)).astype(np.float32)
    ids, d, cnt = a.get_nns_by_vector_batch(Q, 10, -1)
    for qi in range(len(Q)):
        rid, rd = o.get_nns_by_vector(Q[qi], 10, -1, include_distances=True)
        assert ids[qi, :int(cnt[qi])].tolist() == rid
        assert d[qi, :int(cnt[qi])].tobytes() == np.array(rd, np.float32).tobytes()


@pytest.mark.parametrize("f,N,T", [(40, 3000, 6), (1000, 2500, 2), (3000, 7000, 2)])
def test_forest_subnormal_centroid_elements(capi, f, N, T):
    """Columns scaled down to the bottom of the float range: x / |x| and (c * n + x / |x|) / (n + 1) land among the
    subnormals, where a quotient CAN be an exact rounding tie.  two_means takes its quotients through fp64
    reciprocals and must notice these cases and divide for real (devutil.hpp centroid_step4); the oracle divides."""
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(424242 + f)
    X = _clustered(rng, N, f).astype(np.float64)
    q = f // 4
    X[:, :q] *= 2.0 ** rng.integers(-146, -132, q)          # subnormal inputs, a few significant bits each
    X[:, q:2 * q] *= 2.0 ** rng.integers(-126, -120, q)     # normal inputs whose quotient by |x| is subnormal
    X = X.astype(np.float32)
    assert (np.abs(X[:, :q]) < 1.2e-38).all() and (X[:, :q] != 0).any()
    o = capi.AnnoyOracle(f, mode=1)
    o.set_items(X)
    o.build(T)
    a = AnnoyIndex(f)
    a.add_items(X)
    a.build(T)
    _compare_forest(a, o, N, T)
    hp = a.get_forest()["hyperplanes"]
    tiny = np.abs(hp[:, :2 * q])
    assert ((tiny > 0) & (tiny < 1.2e-38)).any()            # the case under test did occur


def test_forest_degenerate_inputs(capi):
    """All rows identical (every split is 100% imbalanced -> random fallback) and
    a matrix of zeros (every margin is exactly 0 -> coin flips)."""
    from morna_amd.annoy import AnnoyIndex
    f, N, T = 8, 700, 3
    for X in (np.ones((N, f), np.float32), np.zeros((N, f), np.float32)):
        o = capi.AnnoyOracle(f, mode=1)
        o.set_items(X)
        o.build(T)
        a = AnnoyIndex(f)
        a.add_items(X)
        a.build(T)
        _compare_forest(a, o, N, T)
        assert a.forest_stats()["n_leaves"] > T
        got = a.get_nns_by_item(5, 10, -1)
        assert got == o.get_nns_by_item(5, 10, -1)


def test_recall_matches_faithful_annoy_restatement(capi):
    """Approximate search quality against the faithful (sequential Kiss32,
    depth-first) restatement: recall@10 of the GPU forest must be within 0.05
    of it at the same n_trees / search_k; every returned distance must be the
    true angular distance within 1e-5."""
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(11)
    f, N, T, k = 32, 6000, 10, 10
    X = _clustered(rng, N, f, nc=40, noise=0.6)
    o = capi.AnnoyOracle(f, mode=0)
    o.set_items(X)
    o.build(T)
    a = AnnoyIndex(f)
    a.add_items(X)
    a.build(T)
    items = np.arange(300, dtype=np.int32)
    ids, d, cnt = a.get_nns_by_item_batch(items, k, -1)
    hit_g = hit_o = 0
    for qi, it in enumerate(items):
        dd = angular64(X, int(it))
        true = set(np.argsort(dd, kind="stable")[:k].tolist())
        hit_g += len(true & set(ids[qi, :int(cnt[qi])].tolist()))
        hit_o += len(true & set(o.get_nns_by_item(int(it), k, -1)))
        want = np.sqrt(np.maximum(dd[ids[qi, :int(cnt[qi])]], 0))
        assert np.allclose(d[qi, :int(cnt[qi])], want, atol=1e-5)
    assert hit_g / (k * len(items)) >= hit_o / (k * len(items)) - 0.05


def test_sharded_search_device_path_one_rank_rccl():
    """The RCCL path of morna_amd/dist.py on a 1-rank group: query rows stay in HBM
    (device-pointer hand-over), top-k all-gather, merge -- must equal the plain by-item search."""
    import socket
    import torch
    import torch.distributed as dist
    from morna_amd.annoy import AnnoyIndex
    from morna_amd.dist import ShardedSearch
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(12)
        X = _clustered(rng, 3000, 40)
        a = AnnoyIndex(40)
        a.add_items(X)
        a.build(6)
        ss = ShardedSearch(a, 0, 1, 3000)
        items = rng.choice(3000, 77, replace=False).astype(np.int32)
        ids, d, cnt = ss.get_nns_by_local_items(items, 10, -1)
        want = a.get_nns_by_item_batch(items, 10, -1)
        assert ids.tolist() == want[0].astype(np.int64).tolist()
        assert d.astype(np.float32).tobytes() == want[1].tobytes() and cnt.tolist() == want[2].tolist()
        ids2, _, _ = ss.get_nns_by_vector(X[items], 10, -1)
        assert ids2.tolist() == ids.tolist()
    finally:
        dist.destroy_process_group()


def test_save_load_roundtrip(tmp_path):
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(4)
    X = _clustered(rng, 2500, 24)
    a = AnnoyIndex(24)
    a.add_items(X)
    a.build(6)
    want = a.get_nns_by_item_batch(np.arange(50, dtype=np.int32), 10, -1)
    a.save(str(tmp_path / "t.annoy.mor"))
    b = AnnoyIndex(24)
    b.load(str(tmp_path / "t.annoy.mor"))
    assert b.get_n_items() == 2500 and b.get_n_trees() == 6
    got = b.get_nns_by_item_batch(np.arange(50, dtype=np.int32), 10, -1)
    assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes()
    with pytest.raises(IOError):
        AnnoyIndex(25).load(str(tmp_path / "t.annoy.mor"))


def test_load_rejects_damaged_files(tmp_path):
    """A truncated or inconsistent index blob is an IOError, not a crash, an out-of-bounds read on the device or an
    exception thrown across the C ABI: header against the file size, node tables, permutation."""
    import struct
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(6)
    X = _clustered(rng, 900, 12)
    a = AnnoyIndex(12)
    a.add_items(X)
    a.build(3)
    path = str(tmp_path / "ok.annoy.mor")
    a.save(path)
    blob = open(path, "rb").read()
    st = a.forest_stats()
    hdr = 8 + 4 + 4 + 8 + 8 + 8 + 4 + 4 + 11 * 8
    assert len(blob) == hdr + 4 * 900 * 12 + 4 * 3 * 900 + 4 * 6 * st["n_nodes"] + 4 * st["n_split"] * 12

    def attempt(data):
        p = str(tmp_path / "bad.annoy.mor")
        with open(p, "wb") as fh:
            fh.write(data)
        b = AnnoyIndex(12)
        with pytest.raises(IOError):
            b.load(p)
        assert b.get_n_items() == 0                       # nothing half-loaded is left behind
        with pytest.raises(RuntimeError):
            b.get_nns_by_item(0, 3)

    attempt(blob[:len(blob) // 2])                        # truncated
    attempt(blob + b"\0\0\0\0")                           # trailing bytes
    attempt(blob[:16] + struct.pack("<q", 2 ** 40) + blob[24:])          # n_items the file cannot hold
    attempt(blob[:24] + struct.pack("<q", st["n_nodes"] + 2) + blob[32:])  # n_nodes != n_trees + 2 n_split
    rows_end = hdr + 4 * 900 * 12
    bad = bytearray(blob)
    bad[rows_end:rows_end + 4] = struct.pack("<i", 900)   # perm entry == n_items
    attempt(bytes(bad))
    rec0 = rows_end + 4 * 3 * 900
    bad = bytearray(blob)
    bad[rec0:rec0 + 4] = struct.pack("<i", 0)             # root 0's first child is itself: a cycle
    attempt(bytes(bad))
    bad = bytearray(blob)
    bad[rec0 + 12:rec0 + 16] = struct.pack("<i", 901)     # root count beyond the items
    attempt(bytes(bad))
    hp0 = rec0 + 4 * 4 * st["n_nodes"] + 4 * st["n_nodes"]
    bad = bytearray(blob)
    bad[hp0:hp0 + 4] = struct.pack("<i", st["n_split"])   # hyperplane slot out of range
    attempt(bytes(bad))
    # what the traversal kernel relies on without looking -- a node has children exactly when it holds more
    # than K = dim + 2 items, and the children cut the parent's segment in two
    nodes = a.get_forest()["node_rec"]                    # [n_nodes][6] = kind, tree, start, count, child0, child1
    leaf = int(np.nonzero(nodes[:, 0] == 1)[0][0])
    bad = bytearray(blob)
    bad[rec0 + 16 * leaf + 12:rec0 + 16 * leaf + 16] = struct.pack("<i", 12 + 3)   # a leaf that claims K + 1 items
    attempt(bytes(bad))
    split = int(np.nonzero(nodes[:, 0] == 0)[0][-1])
    c1 = int(nodes[split, 5])
    bad = bytearray(blob)
    bad[rec0 + 16 * c1 + 8:rec0 + 16 * c1 + 16] = struct.pack("<ii", int(nodes[c1, 2]) + 1, int(nodes[c1, 3]) - 1)   # a gap between the children
    attempt(bytes(bad))
    bad = bytearray(blob)
    bad[rows_end + 4:rows_end + 8] = bad[rows_end:rows_end + 4]       # an item twice in a tree's permutation
    attempt(bytes(bad))
    b = AnnoyIndex(12)
    b.load(path)                                          # and the intact file still loads
    assert b.get_nns_by_item(5, 4, -1) == a.get_nns_by_item(5, 4, -1)


@pytest.mark.parametrize("D,order", [(3000, "ext"), (257, None), (40, "reversed")])
def test_feature_build_norms_are_canonical(capi, D, order):
    """The rows' squared norms that come out of the feature build (transpose_norms_kernel: the chains of the canonical dot
    filled while the column image is turned into rows) are wave_dot(x, x) bit for bit -- what row_norms_kernel computes
    from finished rows and the oracle's mode-1 dot."""
    rng = np.random.default_rng(77 + D)
    keys, rp, s, c = _synthetic_lines(rng, 900, 700, 5, 200)
    a, prep = _gpu_features(keys, rp, s, c, 900, 20, D, order)
    X = a.get_items()
    n2 = a.get_norms2()
    want = np.array([capi.dot(1, x, x) for x in X], np.float32)
    assert n2.tobytes() == want.tobytes()
    assert (X != 0).any()


def test_feature_build_ignores_ids_past_n_items():
    """Entries whose internal id is not below the n_items of build_features have no cell: both forms of the accumulation
    skip them (the one that looks positions up in the item order must not read past its table)."""
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(8)
    D, n_items, J = 64, 300, 120
    keys = ["chr1 %d %d" % (i, i + 5) for i in range(J)]
    kb = [k.encode() for k in keys]
    key_off = np.zeros(J + 1, np.int64)
    key_off[1:] = np.cumsum([len(k) for k in kb])
    rp, ids, cov = [0], [], []
    for j in range(J):
        line = np.sort(rng.choice(n_items + 50, size=int(rng.integers(5, 80)), replace=False))   # some ids in [300, 350)
        ids += line.tolist()
        cov += rng.integers(1, 9, size=len(line)).tolist()
        rp.append(len(ids))
    args = (np.frombuffer(b"".join(kb), np.uint8), key_off, np.array(rp, np.int64), np.array(ids, np.int32), np.array(cov, np.int32),
            rng.random(J) + 0.5)
    out = []
    for order in (None, np.arange(n_items, dtype=np.int64), np.arange(n_items, dtype=np.int64)[::-1].copy()):
        a = AnnoyIndex(D)
        a.stage_junctions(*args)
        if order is not None:
            a.stage_item_order(order)
        a.build_features(n_items)
        out.append(a.get_items())
    assert out[0].tobytes() == out[1].tobytes() == out[2].tobytes()
    keep = np.array(ids) < n_items
    assert (out[0] != 0).any() and keep.sum() < len(ids)


def test_exact_search_large_shard_selection_edges(capi):
    """Shards of more than 8192 rows take the two-pass selection (exact_select2_kernel).  Its edges: a tie group far larger than
    the candidate room (5000 identical rows: the first pass collects them all, the host makes room and repeats), k above the
    256 the two-pass form handles (falls back to the k-round form), k larger than a thread's share -- each against the
    oracle's bisect_left order (equal distance: higher id first) and against the k-round selection (MORNA_EXACT_SELECT2=0)."""
    import os
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(12)
    N, D = 9000, 40
    X = rng.standard_normal((N, D)).astype(np.float32)
    dup = rng.choice(N, 5000, replace=False)
    X[dup] = X[dup[0]]                                   # 5000 copies of one row
    X[dup[:50]] *= np.float32(2.0)                       # some of them scaled by a power of two: still distance 0
    a = AnnoyIndex(D)
    a.add_items(X)
    items = np.array([int(dup[0]), int(dup[77]), 3, 4], np.int32)
    for k in (10, 300, 6000):
        ids, d, cnt = a.exact_search_by_item_batch(items, k)
        os.environ["MORNA_EXACT_SELECT2"] = "0"
        try:
            ids0, d0, cnt0 = a.exact_search_by_item_batch(items, k)
        finally:
            del os.environ["MORNA_EXACT_SELECT2"]
        assert ids.tolist() == ids0.tolist() and d.tobytes() == d0.tobytes() and cnt.tolist() == cnt0.tolist(), k
        for qi, it in enumerate(items):
            rid, rd = capi.exact_search(X, X[int(it)].astype(np.float64), k)
            assert ids[qi, :len(rid)].astype(np.int64).tolist() == rid.tolist(), (k, qi)
            assert d[qi, :len(rid)].tobytes() == rd.tobytes(), (k, qi)
    ids, d, cnt = a.exact_search_by_item_batch(items[:1], 10)
    assert d[0].tolist() == [0.0] * 10 and ids[0].tolist() == sorted(dup.tolist(), reverse=True)[:10]
