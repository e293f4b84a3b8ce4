"""Per-kernel time inside the last `ms` milliseconds of a rocprofv3 kernel trace (the timed steps of a bench run):
python tools/trace_window.py <kernel_trace.csv> <ms> [top]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) * 1e6
top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows]
t1 = max(e[1] for e in ev)
sel = [e for e in ev if e[0] >= t1 - win]
agg = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    agg[n][0] += 1
    agg[n][1] += e - s
busy = sum(v[1] for v in agg.values())
print("window %.0f ms: %d launches, busy %.1f ms (%.0f%%)" % (win / 1e6, len(sel), busy / 1e6, 100 * busy / win))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-90s calls %5d total %8.2f ms avg %8.1f us" % (n[:90], c, t / 1e6, t / c / 1e3))
