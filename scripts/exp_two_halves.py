import sys, time, threading
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from morna_amd.annoy import AnnoyIndex
from morna_amd.index import prepare_csr
from morna_amd.synth import SEED, synthetic_intropolis

N, D = 50000, 3000
data = synthetic_intropolis(N, J=70000, seed=SEED)
prep = prepare_csr(data["keys"], data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100)
n_items = prep["n_items"]

def make():
    a = AnnoyIndex(D)
    a.stage_junctions(prep["key_bytes"], prep["key_off"], prep["row_ptr"], prep["ids"], prep["cov"], prep["idf"])
    a.build_features(n_items)
    a.synchronize()
    return a

full = make()
for _ in range(3):
    full.unbuild() if hasattr(full, "unbuild") else None
    t = time.perf_counter(); full.build(200, seed=0); full.synchronize(); t_full = time.perf_counter() - t
    full = make()
print("one build of 200 trees: %.2f ms" % (1e3 * t_full))

for delay_ms in (0.0, 0.2, 0.4, 0.8):
    best = 1e9
    for rep in range(3):
        A, B = make(), make()
        def run(ix, d):
            if d: 
                t_end = time.perf_counter() + d * 1e-3
                while time.perf_counter() < t_end: pass
            ix.build(100, seed=0); ix.synchronize()
        t = time.perf_counter()
        tb = threading.Thread(target=run, args=(B, delay_ms)); tb.start()
        run(A, 0.0)
        tb.join()
        best = min(best, time.perf_counter() - t)
    print("two concurrent builds of 100 trees, second delayed %.1f ms: %.2f ms" % (delay_ms, 1e3 * best))
