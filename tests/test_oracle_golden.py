"""Pins the CPU oracle (oracle/) against the reference's own known answers.

Nothing here needs a GPU.  Sources: tests/golden/ (see make_golden.py).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, angular64, assert_tie_aware_order
from oracle import capi, morna_ref


def _lines(embedded, spec):
    return embedded[spec["input"]]


def _tokenized(lines):
    keys, rp, s, c = [], [0], [], []
    for ln in lines:
        k, ss, cc = morna_ref.tokenize_line(ln)
        keys.append(k)
        s += ss
        c += cc
        rp.append(len(s))
    return keys, np.array(rp, np.int64), np.array(s, np.int64), np.array(c, np.int64)


def test_murmur3_known_answers():
    with open(os.path.join(GOLDEN, "murmur3_vectors.json")) as fh:
        g = json.load(fh)
    for k, h in g["documented"].items():          # mmh3's documented answers
        assert morna_ref.mmh3_hash(k) == h
        assert capi.mmh3_32(k) == h
    for k, h in g["vectors"]:
        assert morna_ref.mmh3_hash(k) == h, k
        assert capi.mmh3_32(k) == h, k


def test_tiny_intropolis_hash_col_sign():
    # "chr1 14830 14929" -> h=-28859081, sign -1, col 919@3000 / 55@128
    h = capi.mmh3_32("chr1 14830 14929")
    assert h == -28859081
    assert h % 3000 == 919 and h % 128 == 55
    buf, off = capi.pack_keys(["chr1 14830 14929"])
    hh, col, sign = capi.hash_col_sign(buf, off, 3000)
    assert (int(hh[0]), int(col[0]), int(sign[0])) == (-28859081, 919, -1)


def test_sample_count(embedded):
    # morna.py:1140-1149
    assert morna_ref.count_samples(embedded["generic"]) == embedded["sample_count_expected"]


@pytest.mark.parametrize("name", ["simple", "lossy", "lose_sample"])
@pytest.mark.parametrize("D", [128, 3000])
def test_embedded_orderings_python_ref(embedded, name, D):
    """The reference's hard-coded neighbour orderings, reproduced from the py3
    restatement of go_index/add_junction at a collision-free dimension."""
    spec = embedded["expected"][name]
    idx = morna_ref.go_index_lines(_lines(embedded, spec), D, spec["sample_count"], spec["sample_threshold"])
    assert idx.new_internal_id == spec["n_items"]
    X = idx.matrix32()
    for i, exp in enumerate(spec["orderings"]):
        d = angular64(X, i)
        got, _ = morna_ref.angular_order_exact(X, i, 10)
        assert_tie_aware_order(got, exp, d)


@pytest.mark.parametrize("name", ["simple", "lossy", "lose_sample"])
@pytest.mark.parametrize("D", [40, 128, 3000])
def test_features_c_oracle_equals_python_ref(embedded, embedded_mats, name, D):
    spec = embedded["expected"][name]
    keys, rp, s, c = _tokenized(_lines(embedded, spec))
    buf, off = capi.pack_keys(keys)
    r = capi.index_features(buf, off, rp, s, c, spec["sample_count"], spec["sample_threshold"], D)
    assert r["n_items"] == spec["n_items"]
    M = embedded_mats["%s_D%d_f64" % (name, D)]
    X = embedded_mats["%s_D%d_f32" % (name, D)]
    assert r["M"].tobytes() == M.tobytes()           # bit-exact fp64 accumulation order
    assert r["X"].tobytes() == X.tobytes()
    assert r["ext_ids"].tolist() == embedded_mats["%s_D%d_ext_ids" % (name, D)].tolist()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name", ["simple", "lossy", "lose_sample"])
@pytest.mark.parametrize("D", [128, 3000])
def test_embedded_orderings_annoy_oracle(embedded, embedded_mats, name, D, mode):
    """get_nns_by_item(i, 10, search_k=100) on a 20-tree forest (morna.py:1164-1193)."""
    spec = embedded["expected"][name]
    X = embedded_mats["%s_D%d_f32" % (name, D)]
    a = capi.AnnoyOracle(D, mode=mode)
    a.set_items(X)
    a.build(20)
    assert len(a.roots()) == 20
    for i, exp in enumerate(spec["orderings"]):
        got = a.get_nns_by_item(i, 10, 100)
        assert_tie_aware_order(got, exp, angular64(X, i), tol=2e-6)


def test_tiny_intropolis_features():
    g = np.load(os.path.join(GOLDEN, "tiny_intropolis_D128.npz"))
    with open(os.path.join(GOLDEN, "tiny_intropolis.tsv")) as fh:
        lines = fh.readlines()
    assert morna_ref.count_samples(lines) == int(g["sample_count"]) == 6850
    keys, rp, s, c = _tokenized(lines)
    buf, off = capi.pack_keys(keys)
    r = capi.index_features(buf, off, rp, s, c, 6850, 100, 128)
    assert r["X"].tobytes() == g["X"].tobytes()
    assert r["M"].tobytes() == g["M"].tobytes()
    assert r["ext_ids"].tolist() == g["ext_ids"].tolist()


def test_exact_search_golden():
    g = np.load(os.path.join(GOLDEN, "exact_search_512x128.npz"))
    X, Q = g["X"], g["Q"]
    for qi in range(Q.shape[0]):
        ids, d = capi.exact_search(X, Q[qi], 20)
        assert ids.tolist() == g["ids"][qi].tolist()
        assert d.tobytes() == g["dists"][qi].tobytes()
    # the bisect_left tie rule: duplicates of row 7 come out highest id first
    assert g["ids"][1][:3].tolist() == [101, 100, 7]


def test_finalize_query_c_equals_python():
    rng = np.random.default_rng(3)
    keys = ["chr%d %d %d" % (rng.integers(1, 23), rng.integers(1, 10**8), rng.integers(1, 10**8)) for _ in range(300)]
    freq = {k: int(rng.integers(0, 50)) for k in keys}
    s = morna_ref.RefSearch(1000, 64, freq, np.zeros((1, 64), np.float32))
    for k in keys:
        c, a, b = k.split(" ")
        s.update_query((c, int(a), int(b), int(rng.integers(1, 30))))
    s.finalize_query()
    order = list(s.query.keys())
    q = capi.finalize_query([" ".join(str(t) for t in j) for j in order], [s.query[j] for j in order],
                            [freq[" ".join(str(t) for t in j)] for j in order], 1000, 64)
    assert q.tobytes() == np.array(s.query_sample).tobytes()


def test_canonical_dot_matches_numpy():
    rng = np.random.default_rng(5)
    for f in (1, 3, 64, 255, 256, 257, 3000):
        x = rng.standard_normal(f).astype(np.float32)
        y = rng.standard_normal(f).astype(np.float32)
        ref = float(x.astype(np.float64) @ y.astype(np.float64))
        for mode in (0, 1):
            assert abs(capi.dot(mode, x, y) - ref) <= 1e-5 * max(1.0, np.abs(x * y).sum())


@pytest.mark.parametrize("mode", [0, 1])
def test_annoy_oracle_forest_invariants(mode):
    """Structure the algorithm guarantees whatever the RNG: every tree
    partitions the items, leaves hold <= K = f + 2 ids, split children are
    non-empty, and the search returns true neighbours with decent recall."""
    rng = np.random.default_rng(8675309)
    f, N, T = 16, 3000, 8
    centers = rng.standard_normal((12, f))
    X = (centers[rng.integers(0, 12, N)] + 0.3 * rng.standard_normal((N, f))).astype(np.float32)
    a = capi.AnnoyOracle(f, mode=mode)
    a.set_items(X)
    a.build(T)
    K = f + 2

    def walk(nid):
        if mode == 0 and nid < N:
            return [nid]
        nd = a.node(nid - N if mode == 0 else nid)
        if nd["kind"] == 1:
            assert nd["child0"] <= K
            return list(nd["items"])
        left, right = walk(nd["child0"]), walk(nd["child1"])
        assert left and right
        assert abs(float(np.linalg.norm(nd["v"])) - 1.0) < 1e-4 or not np.any(nd["v"])
        return left + right

    for r in a.roots():
        items = walk(r)
        assert sorted(items) == list(range(N))
    # recall@10 with annoy's default search_k = n * n_trees
    hits = 0
    for i in range(0, 200):
        got = a.get_nns_by_item(i, 10, -1)
        d = angular64(X, i)
        true = set(np.argsort(d, kind="stable")[:10].tolist())
        hits += len(true & set(got))
    assert hits / 2000.0 > 0.5
    assert a.split_nodes() > 0 and a.split_rows() >= a.split_nodes() * (K + 1)
