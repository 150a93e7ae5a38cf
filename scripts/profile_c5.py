#!/usr/bin/env python3
"""configs[4] exactly as bench.py's extra keys run it (its own data: the 50k-sample synthetic intropolis at 8192 features),
for `rocprofv3 --kernel-trace --stats`: the full-size all-pairs pass (50 000 by-item queries x 50 000 rows) and one GPU's
diagonal block of the 8-way cut.  Prints the two bench entries as one JSON object.

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d out -o p -- python3 <repo>/scripts/profile_c5.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from morna_amd.index import prepare_csr  # noqa: E402
from morna_amd.synth import synthetic_intropolis  # noqa: E402


class Args(object):
    junctions = 70_000


data = synthetic_intropolis(50_000, J=Args.junctions)
prep = prepare_csr(data["keys"], data["row_ptr"], data["samples"], data["cov"], data["sample_count"], 100)
out = {"exact_all_pairs": bench.exact_all_pairs(Args, 0, prep, data["sample_count"]),
       "exact_all_pairs_shard": bench.exact_all_pairs(Args, 0, prep, data["sample_count"], shard_of=(0, 8))}
print(json.dumps(out, indent=1))
