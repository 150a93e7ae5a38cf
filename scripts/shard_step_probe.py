#!/usr/bin/env python3
"""One GPU's step of the 8-way (or G-way) row cut of the 50k-sample set, timed on one GPU: what a rank of
`bench.py --gpus G` does between the collectives (shard features + forest + its share of the queries against its shard).
    python3 scripts/shard_step_probe.py [G] [rank]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morna_amd.annoy import AnnoyIndex  # noqa: E402
from morna_amd.index import ParsedLines, prepare_csr  # noqa: E402
from morna_amd.synth import query_items, synthetic_intropolis  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = int(sys.argv[2]) if len(sys.argv) > 2 else 3
data = synthetic_intropolis(50_000, J=70_000)
prep = prepare_csr(data["keys"], data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100)
part = ParsedLines.from_arrays(prep, data["sample_count"]).shard(g, G)
a = AnnoyIndex(3000)
part.stage(a)
n = part.n_items
Q = np.ascontiguousarray(np.random.default_rng(1).standard_normal((1000, 3000)), np.float32)
items = query_items(n, 1000)


def step():
    a.build_features(n)
    a.build(200, seed=0)
    return a.get_nns_by_item_batch(items, 20, 100)


for _ in range(3):
    step()
a.synchronize()
R = 8
t0 = time.perf_counter()
for _ in range(R):
    step()
a.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / R
a.timer_reset()
a.timer_enable(True)
for _ in range(R):
    step()
a.timer_enable(False)
tm = a.timers()
print("shard %d of %d: %d rows, %d lines, %d entries: step %.3f ms; %s" % (
    g, G, n, part.n_lines, part.nnz, ms, {k: round(v["ms"] / R, 3) for k, v in tm.items() if v["ms"] > 0}))
