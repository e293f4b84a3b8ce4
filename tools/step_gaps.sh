#!/bin/bash
# idle time between consecutive replayed c2 steps (adam_step end -> next step's first kernel) from a kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
d=/tmp/sg; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $d -o t -- python bench.py --steps 80 --warmup 20 --iwae-images 0 --cpu-baseline-steps 0 --probe-steps 0 --probe-warmup 0 --no-amdahl --no-graph-profile > $d/stdout.txt 2>&1
f=$(find $d -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(tr) if "adam_step" in r["Kernel_Name"]]
gaps = []
for a, b in zip(ad[40:], ad[41:]):
    end = int(tr[a]["End_Timestamp"]); nxt = int(tr[a + 1]["Start_Timestamp"])
    gaps.append(((nxt - end) / 1e3, tr[a + 1]["Kernel_Name"][:30], (int(tr[b]["End_Timestamp"]) - end) / 1e3))
print("gap to next step's first kernel (us), first kernel, step period (us):")
for g in gaps[:24]: print("  %8.1f  %-30s %8.1f" % g)
P
m=$(find $d -name "*memory_copy_trace.csv" | head -1); [ -n "$m" ] && (head -1 $m; tail -6 $m) | cut -c1-200
grep '^{' $d/stdout.txt | cut -c1-120
