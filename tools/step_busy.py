#!/usr/bin/env python3
"""Per-step GPU busy time out of a rocprofv3 kernel trace (steps end at adam_step_kernel): tools/step_busy.py <trace.csv>
prints, for the last few steps, the span, the union of kernel intervals (busy), the launch count and the largest kernels."""
import csv, sys, collections
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(tr) if 'adam_step' in r['Kernel_Name']]
for a, b in list(zip(idx[:-1], idx[1:]))[-4:]:
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in tr[a + 1:b + 1]]
    t0 = int(tr[a]['End_Timestamp'])
    busy, cur = 0, t0
    for s, e, _ in ev:
        s = max(s, cur)
        if e > s:
            busy += e - s; cur = e
    by = collections.Counter()
    for s, e, n in ev:
        by[n[:60]] += e - s
    print("span %.2f ms busy %.2f ms launches %d aten %d" % ((ev[-1][1] - t0) / 1e6, busy / 1e6, len(ev), sum('at::native' in n for _, _, n in ev)))
for n, t in by.most_common(12):
    print("   %8.2f ms %s" % (t / 1e6, n))
