#!/usr/bin/env python3
"""Timing of the approximate search alone at C3 (index built once): per-kernel-group HIP-event times for a few settings.
    python scripts/bench_query.py            # MORNA_LIB=<.so> times an experimental build of the library
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morna_amd.annoy import AnnoyIndex  # noqa: E402
from morna_amd.index import prepare_csr  # noqa: E402
from morna_amd.synth import query_items, synthetic_intropolis  # noqa: E402

N, D, T = 50_000, 3000, int(os.environ.get("TREES", "40"))
data = synthetic_intropolis(N, J=70_000)
prep = prepare_csr(data["keys"], data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100)
a = AnnoyIndex(D)
a.stage_junctions(prep["key_bytes"], prep["key_off"], prep["row_ptr"], prep["ids"], prep["cov"], prep["idf"])
a.stage_item_order(prep["ext_ids"])
a.build_features(prep["n_items"])
a.build(T)
items = query_items(prep["n_items"], 1000)
for _ in range(3):
    a.get_nns_by_item_batch(items, 20, 100)
a.timer_reset()
a.timer_enable(True)
t0 = time.perf_counter()
R = 10
for _ in range(R):
    res = a.get_nns_by_item_batch(items, 20, 100)
wall = (time.perf_counter() - t0) / R
a.timer_enable(False)
tm = a.timers()
print("lib=%s trees=%d: wall %.3f ms, query group %.3f ms, filter contraction %.3f ms (%.0f TFLOP/s)" % (
    os.path.basename(os.environ.get("MORNA_LIB", "libmorna_hip.so")), T, 1e3 * wall, tm["query"]["ms"] / R, tm["query_filter"]["ms"] / R,
    tm["query_filter"]["bytes"] / 1e12 / (tm["query_filter"]["ms"] / 1e3)))
