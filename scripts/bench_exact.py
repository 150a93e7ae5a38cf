#!/usr/bin/env python3
"""Times the exact (brute-force, reference-order fp64) search.

    python scripts/bench_exact.py --samples 50000 --features 3000 --queries 1000 --k 20
    python scripts/bench_exact.py --samples 6250 --features 8192 --queries 6250    # one GPU's share of config 5
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=50000)
    ap.add_argument("--features", type=int, default=3000)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--check", type=int, default=3, help="queries verified against the CPU oracle")
    a = ap.parse_args()
    from morna_amd.annoy import AnnoyIndex
    rng = np.random.default_rng(8675309)
    centers = rng.standard_normal((64, a.features)).astype(np.float32)
    X = (centers[rng.integers(0, 64, a.samples)] * (rng.random((a.samples, a.features), dtype=np.float32) < 0.3)
         + 0.1 * rng.standard_normal((a.samples, a.features), dtype=np.float32)).astype(np.float32)
    idx = AnnoyIndex(a.features)
    idx.add_items(X)
    qi = rng.choice(a.samples, size=min(a.queries, a.samples), replace=False)
    Q = X[qi].astype(np.float64)
    idx.exact_search_batch(Q[:8], a.k)                   # warm-up (uploads the rows)
    idx.timer_reset()
    idx.timer_enable(True)
    t0 = time.perf_counter()
    ids, d, cnt = idx.exact_search_batch(Q, a.k)
    dt = time.perf_counter() - t0
    idx.timer_enable(False)
    tm = idx.timers()["exact"]
    out = dict(samples=a.samples, features=a.features, queries=len(Q), k=a.k, seconds=dt,
               queries_per_sec=len(Q) / dt, scan_ms=tm["ms"], scan_alg_bytes=tm["bytes"],
               scan_alg_GBps=(tm["bytes"] / 1e9) / (tm["ms"] / 1e3) if tm["ms"] else None,
               scan_TFLOPs=2.0 * len(Q) * a.samples * a.features / (tm["ms"] / 1e3) / 1e12 if tm["ms"] else None)
    if a.check:
        from oracle import capi
        for i in range(min(a.check, len(Q))):
            rid, rd = capi.exact_search(X, Q[i], a.k)
            assert ids[i].astype(np.int64).tolist() == rid.tolist(), i
            assert d[i].tobytes() == rd.tobytes(), i
        out["verified_vs_oracle"] = min(a.check, len(Q))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
