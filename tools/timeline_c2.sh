#!/bin/bash
# kernel timeline of one replayed c2 step -> gpurun_out/<tag>/timeline_C25000.txt (extra env vars are passed through)
tag=${1:-tl}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
d=$out/tl; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python bench.py --exemplars 25000 --steps 60 --warmup 20 --iwae-images 0 --cpu-baseline-steps 0 --probe-steps 0 --probe-warmup 0 --no-amdahl > $d/stdout.txt 2>&1
f=$(find $d -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 40 > $out/timeline_C25000.txt
grep '^{' $d/stdout.txt | cut -c1-160
rm -rf $d
