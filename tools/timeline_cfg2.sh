#!/bin/bash
# kernel timeline of one replayed step of --config $1 -> gpurun_out/$2/timeline_$1.txt
cfg=$1; tag=${2:-tl}; out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
d=$out/tlc; rm -rf $d; mkdir -p $d
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python bench.py --config $cfg --steps 12 --warmup 6 --cpu-baseline-steps 0 > $d/stdout.txt 2>&1
f=$(find $d -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 10 > $out/timeline_$cfg.txt
grep '^{' $d/stdout.txt | cut -c1-160
rm -rf $d
