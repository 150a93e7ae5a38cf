#!/usr/bin/env python3
"""BASELINE.json configs[3] on the box we have: the 200k-sample x 3000-feature synthetic intropolis, cut 8 ways by rows,
ONE of the eight shards built and searched on one MI355X (the 8-GPU run is the driver's).

  * the 200k data set is generated once and passed through the global pre-pass (threshold, cumulative frequency,
    idf = log(200000 / freq), first-seen ids over the whole file: morna.py:357-382);
  * shard g's lines are cut by the library (morna_lines_shard) -- 25 000 rows;
  * its feature matrix is compared BYTE FOR BYTE with rows [25000 g, 25000 (g + 1)) of the oracle's matrix of the WHOLE
    data set (oracle/morna_oracle.c, the C restatement of morna.py:344-388);
  * a step as bench.py times it (features + 200-tree forest + 1000 by-item queries through the in-library sharded
    path on a 1-rank RCCL communicator), and the exact search of 64 queries against the oracle.

    python3 scripts/c4_200k_shard.py [shard] > profiles/r03_c4_200k_shard.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (before the library: see tests/conftest.py)
from morna_amd.annoy import AnnoyIndex  # noqa: E402
from morna_amd.index import ParsedLines, prepare_csr, shard_bounds  # noqa: E402
from morna_amd.synth import query_items, synthetic_intropolis  # noqa: E402
from oracle import capi  # noqa: E402

N, D, T, Q, K, SEARCH_K, WORLD = 200_000, 3000, 200, 1000, 20, 100, 8
g = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out = {"workload": "synthetic intropolis %d samples x %d features cut into %d row shards (BASELINE.json configs[3]); shard %d on one GPU"
                   % (N, D, WORLD, g)}
t0 = time.perf_counter()
data = synthetic_intropolis(N, J=70_000)
out["generate_s"] = time.perf_counter() - t0
t0 = time.perf_counter()
prep = prepare_csr(data["keys"], data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100)
out["global_prepass_s"] = time.perf_counter() - t0
out["n_items"], out["nnz"], out["lines"] = prep["n_items"], int(len(prep["ids"])), int(len(prep["idf"]))
t0 = time.perf_counter()
lines = ParsedLines.from_arrays(prep, data["sample_count"])
part = lines.shard(g, WORLD)
out["shard_cut_s"] = time.perf_counter() - t0
bounds = shard_bounds(prep["n_items"], WORLD)
out["shard"] = {"rank": g, "world": WORLD, "id_offset": part.id_offset, "rows": part.n_items, "nnz": part.nnz, "lines": part.n_lines}
assert (part.id_offset, part.n_items) == (bounds[g], bounds[g + 1] - bounds[g])
print("[c4] data %.0fs, pre-pass %.0fs, cut %.1fs" % (out["generate_s"], out["global_prepass_s"], out["shard_cut_s"]), file=sys.stderr, flush=True)

a = AnnoyIndex(D)
part.stage(a)
a.build_features(part.n_items)
X = a.get_items()

# the oracle's matrix of the WHOLE data set, rows of this shard
t0 = time.perf_counter()
buf, off = capi.pack_keys(data["keys"])
ref = capi.index_features(buf, off, data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100, D, max_items=prep["n_items"])
out["oracle_features_s"] = time.perf_counter() - t0
lo, hi = bounds[g], bounds[g + 1]
same = X.tobytes() == ref["X"][lo:hi].tobytes()
out["shard_matrix_equals_rows_of_the_whole_oracle_matrix"] = bool(same)
assert ref["ext_ids"].tolist() == prep["ext_ids"].tolist()
Xw = ref["X"]
del ref
print("[c4] oracle %.0fs, matrix identical: %s" % (out["oracle_features_s"], same), file=sys.stderr, flush=True)

# a step as bench.py times it, queries through the in-library sharded path (1-rank RCCL communicator)
a.comm_init(AnnoyIndex.comm_unique_id(), 0, 1)
items = query_items(part.n_items, Q)


def step():
    a.build_features(part.n_items)
    a.build(T, seed=0)
    return a.get_nns_by_item_sharded(items, K, SEARCH_K, n_each=[len(items)])


for _ in range(2):
    step()
a.synchronize()
t0 = time.perf_counter()
R = 5
for _ in range(R):
    ids, d, cnt = step()
a.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / R
st = a.forest_stats()
out["step_ms_per_gpu"] = ms
out["samples_per_sec_if_8_gpus_ran_like_this_one"] = N / (ms / 1e3)
out["forest"] = {k: st[k] for k in ("n_nodes", "n_split", "max_depth", "split_attempts", "fallback_nodes")}
assert (cnt == K).all() and (ids[:, 0] == items).all()
# exact search of the shard: 64 queries taken from the WHOLE matrix (rows of other shards too) against the oracle
rng = np.random.default_rng(4)
qi = rng.choice(prep["n_items"], 64, replace=False)
eids, ed, ecnt = a.exact_search_sharded(Xw[qi].astype(np.float64), K)
ok = True
for j in range(0, 64, 4):
    rid, rd = capi.exact_search(X, Xw[qi[j]].astype(np.float64), K)
    ok = ok and eids[j].tolist() == rid.tolist() and ed[j].tobytes() == rd.tobytes()
out["exact_search_of_the_shard_equals_oracle"] = bool(ok)
own = ids[:, :]  # approximate answers are local ids here (one rank): recall against the shard's exact search
ex = a.exact_search_by_item_batch(items[:200], K)[0]
out["recall_at_20_vs_exact_within_the_shard"] = float(np.mean([len(set(own[i].tolist()) & set(ex[i].tolist())) / float(K) for i in range(200)]))
a.comm_destroy()
print(json.dumps(out, indent=1))
assert same and ok
