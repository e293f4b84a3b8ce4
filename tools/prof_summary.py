#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV pair: per-kernel totals and the last step's timeline."""
import csv, sys
d = sys.argv[1]
pre = sys.argv[2] if len(sys.argv) > 2 else "r01"
rows = list(csv.DictReader(open("%s/%s_kernel_stats.csv" % (d, pre))))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms %.2f" % (tot / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print("%-86s calls=%6s avg_us=%9.2f tot_ms=%8.2f %5.1f%%" % (r['Name'][:86], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot))
