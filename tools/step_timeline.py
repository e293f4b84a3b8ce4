#!/usr/bin/env python3
"""Timeline of one replayed training step out of a rocprofv3 kernel trace: tools/step_timeline.py <trace.csv> [step]."""
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if 'adam_step' in r['Kernel_Name']]
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[n], idx[n + 1]
t0 = int(tr[a]['End_Timestamp'])
prev_end = t0
for r in tr[a + 1:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f %8.1f dur %7.1f gap %6.1f q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3,
                                                     r.get('Queue_Id', '?'), r['Kernel_Name'][:60]))
    prev_end = max(prev_end, e)
