"""Statistical bridge between the two restatements of annoy (CPU only).

Forest parity for N > K is UNPINNED: no reference fixture has N > K and annoy itself is absent.  The HIP
forest is compared bit for bit with oracle mode 1, which differs from the faithful mode 0 in ONE respect: a random stream
per (tree, level, segment, attempt) instead of annoy's single sequential Kiss32 stream (a level-synchronous build cannot
replay a sequential stream).  That changes WHICH random hyperplanes are drawn; it must not change the algorithm.  This
test cannot make parity green -- nothing can, here -- but it is the evidence that the distribution of what the two modes
build and find is the same: over 6 seeds on clustered data with N ~ 30 K,
  leaf sizes (mean, deciles), leaf depth, split attempts per split node, rows per split attempt, fallback nodes,
  candidates inspected per query at morna's search_k = 100, recall@20
agree within the tolerances stated at each assertion (seed-to-seed spread of mode 0 itself is the yardstick), and on data
that forces the fallback (a block of identical rows larger than a leaf) both modes take it."""
import numpy as np
import pytest

from conftest import angular64
from oracle import capi

D, N, T, KNN, SEARCH_K = 30, 1000, 30, 20, 100      # K = D + 2 = 32 ids per leaf, N ~ 31 K
SEEDS = range(6)


def _data(seed, dup=0):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((12, D)).astype(np.float32)
    X = (cent[rng.integers(0, 12, N)] + 0.35 * rng.standard_normal((N, D))).astype(np.float32)
    if dup:
        X[:dup] = X[0]                               # more identical rows than a leaf may hold: no hyperplane separates them
    return X


def _stats(mode, X, seed):
    o = capi.AnnoyOracle(D, mode=mode, seed=1234 + seed)
    o.set_items(X)
    o.build(T)
    leaf_sizes, leaf_depths, n_split, n_fallback = [], [], 0, 0
    for r in o.roots():
        stack = [(r, 0)]
        while stack:
            nid, dep = stack.pop()
            if mode == 0 and nid < N:                # mode 0 keeps annoy's convention: ids below n_items are single items
                leaf_sizes.append(1)
                leaf_depths.append(dep)
                continue
            nd = o.node(nid - N if mode == 0 else nid)
            if nd["kind"] == 1:
                leaf_sizes.append(nd["child0"])
                leaf_depths.append(dep)
            else:
                n_split += 1
                n_fallback += 0 if nd["v"].any() else 1
                stack.append((nd["child0"], dep + 1))
                stack.append((nd["child1"], dep + 1))
    q = np.random.default_rng(seed + 100).choice(N, 60, replace=False)
    cands, hits = [], 0
    for it in q:
        res, c = o.get_nns_by_item(int(it), KNN, SEARCH_K, return_cand=True)
        cands.append(c)
        hits += len(set(res) & set(np.argsort(angular64(X, int(it)), kind="stable")[:KNN].tolist()))
    ls = np.array(leaf_sizes, float)
    return dict(leaf_mean=ls.mean(), leaf_q=np.quantile(ls, [0.1, 0.5, 0.9]), leaf_depth=float(np.mean(leaf_depths)),
                n_split=n_split, attempts=o.split_nodes() / n_split, rows_per_attempt=o.split_rows() / o.split_nodes(),
                fallback=n_fallback, cand=float(np.mean(cands)), recall=hits / (KNN * len(q)), leaf_hist=np.bincount(ls.astype(int), minlength=D + 3))


@pytest.fixture(scope="module")
def runs():
    return {(mode, seed): _stats(mode, _data(seed), seed) for mode in (0, 1) for seed in SEEDS}


def _mean(runs, mode, key):
    return float(np.mean([runs[(mode, s)][key] for s in SEEDS]))


def _spread(runs, mode, key):
    return float(np.std([runs[(mode, s)][key] for s in SEEDS]))


def test_leaf_sizes_and_depth_have_one_distribution(runs):
    for s in SEEDS:
        for mode in (0, 1):
            assert runs[(mode, s)]["leaf_hist"][D + 3:].sum() == 0        # no leaf above K = D + 2, either mode
    # pooled histograms of the 6 x 30 trees: total-variation distance of the leaf-size distributions below 0.06
    h0 = sum(runs[(0, s)]["leaf_hist"] for s in SEEDS).astype(float)
    h1 = sum(runs[(1, s)]["leaf_hist"] for s in SEEDS).astype(float)
    tv = 0.5 * np.abs(h0 / h0.sum() - h1 / h1.sum()).sum()
    assert tv < 0.06, tv
    # means within 4 % of each other, deciles within 2 ids, mean leaf depth within 0.15 levels
    assert abs(_mean(runs, 0, "leaf_mean") - _mean(runs, 1, "leaf_mean")) < 0.04 * _mean(runs, 0, "leaf_mean")
    q0 = np.mean([runs[(0, s)]["leaf_q"] for s in SEEDS], axis=0)
    q1 = np.mean([runs[(1, s)]["leaf_q"] for s in SEEDS], axis=0)
    assert np.all(np.abs(q0 - q1) <= 2.0), (q0, q1)
    assert abs(_mean(runs, 0, "leaf_depth") - _mean(runs, 1, "leaf_depth")) < 0.15


def test_split_work_is_the_same(runs):
    # create_split attempts per split node (1 unless an attempt is rejected for imbalance >= 0.95): within 0.02;
    # rows streamed per attempt: within 3 %; split nodes per forest: within 3 %
    assert abs(_mean(runs, 0, "attempts") - _mean(runs, 1, "attempts")) < 0.02
    assert abs(_mean(runs, 0, "rows_per_attempt") - _mean(runs, 1, "rows_per_attempt")) < 0.03 * _mean(runs, 0, "rows_per_attempt")
    assert abs(_mean(runs, 0, "n_split") - _mean(runs, 1, "n_split")) < 0.03 * _mean(runs, 0, "n_split")
    assert sum(runs[(m, s)]["fallback"] for m in (0, 1) for s in SEEDS) == 0      # separable data: nobody falls back


def test_search_inspects_and_finds_the_same(runs):
    # candidates per query at search_k = 100: means within 6 %; recall@20: means within 0.03 (mode 0's own seed-to-seed
    # standard deviation is of that order)
    c0, c1 = _mean(runs, 0, "cand"), _mean(runs, 1, "cand")
    assert abs(c0 - c1) < 0.06 * c0, (c0, c1)
    r0, r1 = _mean(runs, 0, "recall"), _mean(runs, 1, "recall")
    assert abs(r0 - r1) < 0.03, (r0, r1, _spread(runs, 0, "recall"))
    assert min(r0, r1) > 0.5


def test_both_modes_take_the_random_fallback_when_no_hyperplane_can_split():
    """120 identical rows, K = 32: every two_means split of a node that holds only them is rejected three times and
    annoy assigns the sides at random (imbalance > 0.99 -> zero hyperplane).  Both modes do so, about as often."""
    f0 = f1 = 0
    for s in range(3):
        X = _data(s, dup=120)
        a, b = _stats(0, X, s), _stats(1, X, s)
        assert a["fallback"] > 0 and b["fallback"] > 0
        assert a["leaf_hist"][D + 3:].sum() == 0 and b["leaf_hist"][D + 3:].sum() == 0
        f0 += a["fallback"]
        f1 += b["fallback"]
    assert 0.6 < f0 / f1 < 1.0 / 0.6, (f0, f1)
