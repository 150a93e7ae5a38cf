import sys, numpy as np
sys.path.insert(0, '.')
from morna_amd.annoy import AnnoyIndex
from oracle import capi
N, D, k = 1500, 3000, 10
rng = np.random.default_rng(N + D)
X = (rng.standard_normal((N, D)) * (rng.random((N, D)) < 0.2)).astype(np.float32)
X[N // 2] = X[3]
X[N // 2 + 1] = 3.0 * X[3]
Q = rng.standard_normal((9, D))
Q[0] = X[3]
a = AnnoyIndex(D); a.add_items(X)
ids, d, cnt = a.exact_search_batch(Q[:1], 30)
print("gpu", ids[0][:12], d[0][:12])
rid, rd = capi.exact_search(X, Q[0], 30)
print("ora", rid[:12], rd[:12])
for j in (3, 750, 751):
    print(j, repr(capi.cosine_distance(X[j], Q[0])))
