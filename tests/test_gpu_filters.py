"""The fp16 matrix-core filters at the row widths that ship (D = 3000: the bench; D = 8192: configs[4]) on rows built
to defeat them.  -m gpu

The split of a level (splitmm.hip) and the candidate filter of the approximate search decide from fp16 copies only
what they can PROVE and hand everything else to the canonical fp32 arithmetic, so switching them off must change
nothing: whole-forest and search digests are compared across MORNA_SPLIT_MM = 0 / 1, MORNA_QUERY_FILTER = 0 / 1 and
MORNA_QUERY_DENSE = 0 / 1 (the filter dots of a whole batch as one contraction, or candidate by candidate), and across
MORNA_QUERY_SPLIT_TRAVERSE = 0 / 1 and MORNA_QUERY_SPREAD = 0 / 1 (how a batch's, and a small batch's, traversal is dealt out),
MORNA_SPLIT_ORDER = 0 / 1 and MORNA_SPLIT_LISTS = 0 / 1 (the order of the rows of the split contraction, and the per-tile
task lists: both only choose which products are computed; the 256-wide case with 150 trees has levels with enough split
nodes for the lists) -- separate processes: the switches are read once.  The rows:

  * midpoints  s (a / |a| + b / |b|) of rows a, b from two different clusters: against the hyperplane that separates
    those clusters (the normalised difference of their centroids) the dot product cancels to ~1e-7 of |row| |h|, so
    the side is decided by the rounding of the canonical fp32 dot;
  * near-mirror pairs  x, -x (1 + 1e-6 noise);
  * spikes: one element 1e9 times larger than the rest -- the fp16 copy keeps the spike alone, the sign of the dot
    is in what it dropped whenever the hyperplane is small at the spike;
  * rows at the edges of the fp16 scaling: largest element exactly a power of two, and one ulp below one;
  * norms from 1e-18 to 1e18, a zero row, exact duplicates, exact scaled duplicates, near-duplicates in the 5th digit.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, {root!r})
from morna_amd.annoy import AnnoyIndex
N, D, T = {N}, {D}, {T}
rng = np.random.default_rng(2026 + D)
nc = 5
C = rng.standard_normal((nc, D)).astype(np.float32)
lab = rng.integers(0, nc, N)
X = C[lab] + np.float32(0.25) * rng.standard_normal((N, D), dtype=np.float32)
X *= (10.0 ** rng.uniform(-3, 3, (N, 1))).astype(np.float32)
unit = X / np.linalg.norm(X.astype(np.float64), axis=1, keepdims=True).astype(np.float32)
# midpoints of rows from different clusters
m0 = 2000
for i in range(m0, m0 + 1500):
    a, b = rng.integers(0, m0, 2)
    while lab[a] == lab[b]:
        b = rng.integers(0, m0)
    X[i] = (unit[a] + unit[b]) * np.float32(10.0 ** rng.uniform(-2, 2))
# exact midpoints of the cluster centres themselves, at many scales
for i in range(3500, 3600):
    a, b = rng.choice(nc, 2, replace=False)
    ca, cb = C[a] / np.linalg.norm(C[a]), C[b] / np.linalg.norm(C[b])
    X[i] = (ca + cb) * np.float32(2.0 ** rng.integers(-20, 20))
# near-mirror pairs
for i in range(3600, 4000, 2):
    X[i + 1] = -X[i] * (1.0 + 1e-6 * rng.standard_normal(D)).astype(np.float32)
# spikes
for i in range(4000, 4200):
    X[i] = (1e-9 * rng.standard_normal(D)).astype(np.float32)
    X[i, rng.integers(0, D)] = np.float32(rng.choice([-1.0, 1.0]))
    X[i] *= np.float32(10.0 ** rng.uniform(-3, 3))
# fp16 scaling edges
for i in range(4200, 4300):
    j = rng.integers(0, D)
    X[i] = np.clip(X[i], -1, 1) * np.float32(0.4)
    X[i, j] = np.float32(1.0) if i % 2 else np.nextafter(np.float32(2.0), np.float32(0.0))
    X[i] *= np.float32(2.0 ** rng.integers(-30, 30))
X[4300] = 0.0
X[4301] = X[4302]
X[4303:4400] = X[50] * (1.0 + 1e-5 * rng.standard_normal((97, D))).astype(np.float32)
X[4400:4450] = X[50] * np.float32(4.0)
X[4450] *= np.float32(1e-18)
X[4451] *= np.float32(1e15)
a = AnnoyIndex(D)
a.add_items(X)
a.build(T)
f = a.get_forest()
h = hashlib.sha256()
for k in ("perm", "node_rec", "hyperplanes", "hp_node"):
    h.update(np.ascontiguousarray(f[k]).tobytes())
items = np.concatenate([np.arange(40), np.arange(2000, 2040), np.arange(3500, 3520), np.arange(3600, 3640),
                        np.arange(4000, 4020), np.arange(4200, 4220), np.arange(4300, 4320), np.arange(4400, 4410),
                        [4450, 4451]]).astype(np.int32)
for n, sk in ((20, 100), (10, -1), (50, 4000)):
    ids, d, cnt = a.get_nns_by_item_batch(items, n, sk)
    h.update(ids.tobytes()); h.update(d.tobytes()); h.update(cnt.tobytes())
Q = (X[items[:32]].astype(np.float64) * (1 + 1e-3 * rng.standard_normal((32, D)))).astype(np.float32)
ids, d, cnt = a.get_nns_by_vector_batch(Q, 20, 100)
h.update(ids.tobytes()); h.update(d.tobytes()); h.update(cnt.tobytes())
st = a.forest_stats()
print("DIGEST", h.hexdigest(), st["n_split"], st["max_depth"])
"""


@pytest.mark.parametrize("D,N,T", [(3000, 16000, 6), (8192, 34000, 3), (256, 40000, 150)])
def test_adversarial_rows_filters_change_nothing(tmp_path, D, N, T):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = str(tmp_path / "digest.py")
    with open(script, "w") as fh:
        fh.write(_SCRIPT.format(root=root, N=N, D=D, T=T))
    out, open_lines = {}, []
    for name, extra in (("default", {"MORNA_DEBUG_OPEN": "1"}), ("no_mm", {"MORNA_SPLIT_MM": "0"}),
                        ("no_qf", {"MORNA_QUERY_FILTER": "0"}), ("no_dense", {"MORNA_QUERY_DENSE": "0"}),
                        ("no_order", {"MORNA_SPLIT_ORDER": "0"}), ("no_lists", {"MORNA_SPLIT_LISTS": "0", "MORNA_SPLIT_ORDER": "0"}),
                        # round 3: the traversal of a batch as root margins by query groups + one-wave descents, or fused (one
                        # workgroup per query); small batches dealt out over the chip, or one workgroup per query
                        ("fused_traverse", {"MORNA_QUERY_SPLIT_TRAVERSE": "0"}), ("no_spread", {"MORNA_QUERY_SPREAD": "0"})):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0].split()[1:]
        if name == "default":
            open_lines = [ln for ln in r.stderr.splitlines() if ln.startswith("[morna] split_mm level")]
    assert out["default"] == out["no_mm"] == out["no_qf"] == out["no_dense"] == out["no_order"] == out["no_lists"], out
    assert out["default"] == out["fused_traverse"] == out["no_spread"], out
    assert int(out["default"][2]) >= 2                       # at least two levels went through the contraction
    assert open_lines, "the matrix-core split did not run"
    print("\n".join(open_lines))                             # pytest -s: the open-pair share per level
    # the filter must have left pairs open (the rows above) and still decided most
    shares = [float(ln.split("(")[-1].split("%")[0]) for ln in open_lines]
    assert max(shares) > 0.0 and min(shares) < 50.0, open_lines
