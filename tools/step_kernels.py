#!/usr/bin/env python3
"""Kernels of the LAST training step in a rocprofv3 kernel trace (steps are delimited by the optimizer's launches), longer than a
threshold: tools/step_kernels.py <kernel_trace.csv> [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
ends = [i for j, i in enumerate(idx) if j + 1 == len(idx) or idx[j + 1] - i > 50]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
busy = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    if d > thr:
        print("%8.1f  +%8.1f us  grid %-10s %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r["Kernel_Name"][:100]))
print("step span ms %.3f, kernels %d, sum of kernel times ms %.3f" % ((int(rows[b - 1]["End_Timestamp"]) - t0) / 1e6, b - a, busy / 1e3))
