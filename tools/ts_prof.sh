#!/bin/bash
# kernel-trace stats of tools/ts_abl.py under each EVAE_TS_ABL value -> stdout (kernel name, calls, avg us)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in ${ABLS:-0 1 3 7}; do
  d=/tmp/tsprof_$a; rm -rf $d
  EVAE_TS_ABL=$a timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python tools/ts_abl.py > /dev/null 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== abl $a"; python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "topk" in r["Name"] or "Memset" in r["Name"] or "fill" in r["Name"].lower():
        print("%-70s calls %5s avg %8.1f us min %8.1f max %8.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
P
done
