#!/usr/bin/env python3
"""Print per-launch durations of the forest kernels from a rocprofv3 kernel trace csv (last step only)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
last_iota = max(i for i, r in enumerate(rows) if "iota_perm" in r["Kernel_Name"])
for r in rows[last_iota:]:
    n = r["Kernel_Name"].split("(")[0].replace("morna::", "").replace("void ", "")
    if n.startswith(("split", "two_means", "query", "invert", "sched", "partition")):
        print("%-28s %8.3f ms  grid %s" % (n[:28], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size", "")))
