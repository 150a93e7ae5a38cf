#!/usr/bin/env python3
"""Condenses rocprofv3 output into the files kept under profiles/.

    python scripts/profile_summary.py --stats <dir>/x_kernel_stats.csv \
        --fetch <dir>/fetch_counter_collection.csv --write <dir>/write_counter_collection.csv \
        --steps 1 --tag r01_c3 --config '{"samples":50000,"features":3000,"trees":200}'

Writes profiles/<tag>_kernel_stats.csv (copy), profiles/<tag>_traffic.json (HBM-side bytes per kernel
from the FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md section HBM prescribes
for wide coalesced reads on gfx950; both counters are in KiB).
"""
import argparse
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("_ZN5morna"):   # a name rocprofv3 left mangled: _ZN5morna<len><name>...
                rest = k[len("_ZN5morna"):]
                digits = "".join(ch for ch in rest[:3] if ch.isdigit())
                k = "morna::" + rest[len(digits):len(digits) + int(digits)]
            a = agg[k]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--steps", type=int, default=1, help="steps (incl. warmup) the PMC runs executed")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--config", default="{}")
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    if a.stats:
        shutil.copyfile(a.stats, os.path.join(out_dir, a.tag + "_kernel_stats.csv"))
    if a.fetch and a.write:
        f, w = per_kernel(a.fetch), per_kernel(a.write)
        kernels = {}
        for k in sorted(set(f) | set(w)):
            calls = max(f.get(k, [0])[0], w.get(k, [0])[0])
            fetch_b = 2.0 * f[k][1] * 1024 if k in f else 0.0      # gfx950: FETCH_SIZE reads 1/2 of wide streams
            write_b = w[k][1] * 1024 if k in w else 0.0
            kernels[k] = dict(calls=calls, launches_per_step=calls / float(a.steps),
                              fetch_bytes_corrected=fetch_b, write_bytes=write_b,
                              hbm_bytes_per_launch=(fetch_b + write_b) / max(calls, 1),
                              ms_in_pmc_run=(f.get(k) or w.get(k))[2])
        doc = dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace",
                   correction="FETCH_SIZE x2 (MI355X_MICROARCH.md: gfx950 reports half the bytes of wide coalesced reads); KiB units",
                   config=json.loads(a.config), steps=a.steps, kernels=kernels)
        with open(os.path.join(out_dir, a.tag + "_traffic.json"), "w") as fh:
            json.dump(doc, fh, indent=1)
        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["fetch_bytes_corrected"])[:6]:
            print("%-40s calls %3d  HBM-side %.2f GB/launch" % (k, v["calls"], v["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
