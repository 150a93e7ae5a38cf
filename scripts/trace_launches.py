#!/usr/bin/env python3
"""Print per-launch durations of the forest and query kernels from a rocprofv3 kernel trace csv (last step only)."""
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
last_iota = max(i for i, r in enumerate(rows) if "iota_perm" in r["Kernel_Name"])
for r in rows[last_iota:]:
    n = name_of(r["Kernel_Name"])
    if n.startswith(("split", "two_means", "query", "invert", "sched", "partition", "rows_to_half")):
        print("%-28s %8.3f ms" % (n[:28], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
