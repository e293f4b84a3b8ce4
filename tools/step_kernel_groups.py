#!/usr/bin/env python3
"""Kernel time of the LAST training step in a rocprofv3 kernel trace, grouped by kernel name (steps are delimited by the optimizer's
launches): tools/step_kernel_groups.py <kernel_trace.csv> [rows]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 26
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"].lower()]
ends = [i for j, i in enumerate(idx) if j + 1 == len(idx) or idx[j + 1] - i > 50]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = r["Kernel_Name"][:72]; agg[k][0] += 1; agg[k][1] += d; busy += d
print("step span ms %.2f, kernels %d, sum of kernel times ms %.2f" % ((int(rows[b - 1]["End_Timestamp"]) - t0) / 1e6, b - a, busy / 1e3))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-74s n %4d total %8.1f us avg %7.1f" % (k, n, t, t / n))
if len(sys.argv) > 3:                      # every launch of the step in order: offset, duration, gap before it, grid, name
    prev = t0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if sys.argv[3] == "all" or sys.argv[3] in r["Kernel_Name"]:
            print("%9.1f us  dur %7.1f  gap %6.1f  grid %7s wg %4s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")),
                                                                    r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r["Kernel_Name"][:90]))
        prev = max(prev, e)
