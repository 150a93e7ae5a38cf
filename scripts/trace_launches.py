#!/usr/bin/env python3
"""Print per-launch durations of the forest and query kernels from a rocprofv3 kernel trace csv (last step only);
with --timeline after the file name: every launch of the last step with the device idle time before it."""
import csv
import sys


def name_of(raw):
    n = raw.split("(")[0].replace("void ", "")
    if n.startswith("_ZN5morna"):   # a name rocprofv3 left mangled: _ZN5morna<len><name>...
        rest = n[len("_ZN5morna"):]
        digits = "".join(ch for ch in rest[:3] if ch.isdigit())
        n = "morna::" + rest[len(digits):len(digits) + int(digits)]
    return n.replace("morna::", "")


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
timeline = len(sys.argv) > 2 and sys.argv[2] == "--timeline"
if timeline:
    # every launch of the last step, with the idle time of the device before it (start - latest end so far)
    first = [i for i, r in enumerate(rows) if "hash_keys" in r["Kernel_Name"]][-1]
    t0, busy_end, idle = int(rows[first]["Start_Timestamp"]), int(rows[first]["Start_Timestamp"]), 0
    for r in rows[first:]:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = max(a - busy_end, 0)
        idle += gap
        print("%9.3f ms  %-30s %8.3f ms   idle before %7.3f ms" % ((a - t0) / 1e6, name_of(r["Kernel_Name"])[:30], (b - a) / 1e6, gap / 1e6))
        busy_end = max(busy_end, b)
    print("step span %.3f ms, device idle %.3f ms" % ((busy_end - t0) / 1e6, idle / 1e6))
    sys.exit(0)
last_iota = max(i for i, r in enumerate(rows) if "iota_perm" in r["Kernel_Name"])
for r in rows[last_iota:]:
    n = name_of(r["Kernel_Name"])
    if n.startswith(("split", "two_means", "query", "invert", "sched", "partition", "rows_to_half")):
        print("%-28s %8.3f ms" % (n[:28], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
