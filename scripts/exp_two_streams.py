#!/usr/bin/env python3
"""Experiment: does building two half-forests on two streams (two handles, two host
threads) overlap two_means (VALU / latency bound) with split (memory bound)?

Compares one 200-tree build with two concurrent 100-tree builds over the same rows.
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from morna_amd.annoy import AnnoyIndex
    N, D, T = 50000, 3000, 200
    rng = np.random.default_rng(8675309)
    centers = rng.standard_normal((64, D)).astype(np.float32)
    X = centers[rng.integers(0, 64, N)] * (rng.random((N, D), dtype=np.float32) < 0.3)
    X += 0.05 * rng.standard_normal((N, D), dtype=np.float32)

    def fresh():
        a = AnnoyIndex(D)
        a.add_items(X)
        a.get_norms2()          # upload now
        return a

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    one = fresh()
    one.build(T)                 # warm-up (allocations)

    def build_one():
        one_again = fresh_holder[0]
        one_again.build(T)

    # rebuilding needs an unbuilt index: re-add rows each time outside the timed part
    res = []
    for _ in range(3):
        a = fresh()
        a.build(2)               # grow scratch
        b = fresh()
        t0 = time.perf_counter()
        b.build(T)
        res.append((time.perf_counter() - t0) * 1e3)
    print("one handle, %d trees: %.2f ms (min of %s)" % (T, min(res), ["%.2f" % r for r in res]))

    res2 = []
    for delay_ms in (0.0, 0.5, 1.0, 2.0):
        for _ in range(2):
            h1, h2 = fresh(), fresh()

            def run(h, d):
                if d:
                    time.sleep(d / 1e3)
                h.build(T // 2)
            t1 = threading.Thread(target=run, args=(h1, 0.0))
            t2 = threading.Thread(target=run, args=(h2, delay_ms))
            t0 = time.perf_counter()
            t1.start(); t2.start(); t1.join(); t2.join()
            res2.append((delay_ms, (time.perf_counter() - t0) * 1e3))
    for d, ms in res2:
        print("two handles x %d trees, second delayed %.1f ms: %.2f ms" % (T // 2, d, ms))


if __name__ == "__main__":
    fresh_holder = [None]
    main()
