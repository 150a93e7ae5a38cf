#!/usr/bin/env python3
"""The approximate search of ONE query (and of small batches) at C3: wall-clock per call and, under
`rocprofv3 --kernel-trace`, the launches of one call with the device's idle time between them.

    python3 scripts/bench_single_query.py [nq ...]                        # host clock + HIP-event time
    rocprofv3 --kernel-trace --output-format csv -d out -o p -- python3 scripts/bench_single_query.py 1
    python3 scripts/bench_single_query.py --trace out/.../p_kernel_trace.csv   # timeline of the last call
"""
import csv
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeline(path):
    rows = [r for r in csv.DictReader(open(path)) if "query_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = [i for i, r in enumerate(rows) if "query_roots" in r["Kernel_Name"]][-1]
    t0 = int(rows[first]["Start_Timestamp"])
    end = t0
    for r in rows[first:]:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("morna::", "")
        print("%8.2f us  %-36s %7.2f us   idle before %6.2f us" % ((a - t0) / 1e3, name[:36], (b - a) / 1e3, max(a - end, 0) / 1e3))
        end = max(end, b)
    print("span %.2f us" % ((end - t0) / 1e3))


if len(sys.argv) > 2 and sys.argv[1] == "--trace":
    timeline(sys.argv[2])
    sys.exit(0)

from morna_amd.annoy import AnnoyIndex  # noqa: E402
from morna_amd.index import prepare_csr  # noqa: E402
from morna_amd.synth import query_items, synthetic_intropolis  # noqa: E402

N, D, T = 50_000, 3000, 200
data = synthetic_intropolis(N, J=70_000)
prep = prepare_csr(data["keys"], data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100)
a = AnnoyIndex(D)
a.stage_junctions(prep["key_bytes"], prep["key_off"], prep["row_ptr"], prep["ids"], prep["cov"], prep["idf"])
a.stage_item_order(prep["ext_ids"])
a.build_features(prep["n_items"])
a.build(T)
items = query_items(prep["n_items"], 1000)
X = a.get_items()
for nq in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 63]:
    sub = items[:nq]
    for _ in range(5):
        a.get_nns_by_item_batch(sub, 20, 100)
    a.synchronize()
    R = 50
    t0 = time.perf_counter()
    for _ in range(R):
        a.get_nns_by_item_batch(sub, 20, 100)
    wall = (time.perf_counter() - t0) / R
    Q = np.ascontiguousarray(X[sub])
    t0 = time.perf_counter()
    for _ in range(R):
        a.get_nns_by_vector_batch(Q, 20, 100)
    wall_v = (time.perf_counter() - t0) / R
    a.timer_reset()
    a.timer_enable(True, only=["query"])
    for _ in range(R):
        a.get_nns_by_item_batch(sub, 20, 100)
    a.timer_enable(False)
    print("nq %3d: by item %7.1f us per call, by vector %7.1f us, kernels (HIP events) %7.1f us" % (
        nq, 1e6 * wall, 1e6 * wall_v, 1e3 * a.timers()["query"]["ms"] / R))
