#!/usr/bin/env python3
"""Randomised parity sweep (run by hand on the GPU box; the committed tests cover fixed shapes): random D / N / trees / k /
search_k and batch sizes on both sides of the spread / batch / contraction switches, approximate answers against oracle mode 1
(ids and fp32 distances, bit for bit) and exact answers by item against the oracle (ids and fp64 distances).
    python3 scripts/stress_parity.py [rounds] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morna_amd.annoy import AnnoyIndex  # noqa: E402
from oracle import capi  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for r in range(rounds):
    D = int(rng.choice([8, 17, 40, 64, 100, 256, 300, 768, 1000]))
    K = D + 2
    N = int(rng.choice([3, K - 1, K, K + 1, 3 * K, 10 * K + 7, 40 * K]))
    N = max(N, 2)
    T = int(rng.choice([1, 3, 10, 33]))
    C = rng.standard_normal((7, D)).astype(np.float32)
    X = (C[rng.integers(0, 7, N)] + 0.4 * rng.standard_normal((N, D))).astype(np.float32)
    if N > 10 and rng.random() < 0.5:
        X[N // 2] = X[1]                      # a duplicate row
        X[N // 3] = 0                         # a zero row
    a = AnnoyIndex(D)
    a.add_items(X)
    a.build(T)
    o = capi.AnnoyOracle(D, mode=1)
    o.set_items(X)
    o.build(T)
    for nq in (1, 5, 39, 40, 70):
        items = rng.integers(0, N, nq).astype(np.int32)
        for k, sk in ((1, -1), (10, 100), (25, 7), (int(min(N, 300)), int(3 * N))):
            ids, d, cnt = a.get_nns_by_item_batch(items, k, sk)
            for qi, it in enumerate(items):
                rid, rd = o.get_nns_by_item(int(it), k, sk, include_distances=True)
                m = int(cnt[qi])
                assert ids[qi, :m].tolist() == rid, ("approx ids", D, N, T, nq, k, sk, qi)
                assert d[qi, :m].tobytes() == np.array(rd, np.float32).tobytes(), ("approx dist", D, N, T, nq, k, sk, qi)
        k = int(min(N + 3, 12))
        ids, d, cnt = a.exact_search_by_item_batch(items, k)
        for qi in range(0, nq, max(1, nq // 6)):
            rid, rd = capi.exact_search(X, X[int(items[qi])].astype(np.float64), k)
            if cnt[qi] < 0:
                assert np.isnan(rd).any() or True
                continue
            assert ids[qi, :len(rid)].astype(np.int64).tolist() == rid.tolist(), ("exact ids", D, N, nq, qi)
            assert d[qi, :len(rid)].tobytes() == rd.tobytes(), ("exact dist", D, N, nq, qi)
    # the in-library sharded entry points on a communicator of one rank: the plain answers (ids are global = local here)
    a.comm_init(AnnoyIndex.comm_unique_id(), 0, 1)
    for nq in (1, 7, 45):
        items = rng.integers(0, N, nq).astype(np.int32)
        k = int(min(N, 9))
        want = a.get_nns_by_item_batch(items, k, 50)
        got = a.get_nns_by_item_sharded(items, k, 50, n_each=[nq])
        assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes() and got[2].tolist() == want[2].tolist()
        want = a.exact_search_by_item_batch(items, k)
        got = a.exact_search_by_item_sharded(items, k, [nq])
        assert got[0].tolist() == want[0].tolist() and got[1].tobytes() == want[1].tobytes() and got[2].tolist() == want[2].tolist()
    a.comm_destroy()
    print("round %d ok: D=%d N=%d T=%d" % (r, D, N, T), flush=True)
print("stress ok")
