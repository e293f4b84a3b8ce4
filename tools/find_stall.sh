#!/bin/bash
# where does the one-off 5-7 ms step come from: a kernel trace of 600 replayed c2 steps; prints gaps > 1 ms between consecutive
# kernel starts and kernels longer than 1 ms during the replay phase
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
d=/tmp/stall; rm -rf $d; mkdir -p $d
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python bench.py --steps 600 --warmup 20 --iwae-images 0 --cpu-baseline-steps 0 --probe-steps 0 --probe-warmup 0 --no-amdahl --no-graph-profile > $d/stdout.txt 2>&1
f=$(find $d -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(tr) if "adam_step" in r["Kernel_Name"]]
print("kernels", len(tr), "steps", len(adam))
first = adam[60] if len(adam) > 60 else 0
prev_end = int(tr[first]["End_Timestamp"])
for i in range(first + 1, len(tr)):
    r = tr[i]; s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > 1e6:
        print("LONG kernel %.2f ms: %s (step ~%d)" % ((e - s) / 1e6, r["Kernel_Name"][:70], sum(1 for a in adam if a < i)))
    if s - prev_end > 1e6:
        print("GAP %.2f ms before %s (after %s) (step ~%d)" % ((s - prev_end) / 1e6, r["Kernel_Name"][:50], tr[i - 1]["Kernel_Name"][:50], sum(1 for a in adam if a < i)))
    prev_end = max(prev_end, e)
P
grep '^{' $d/stdout.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms'])"
