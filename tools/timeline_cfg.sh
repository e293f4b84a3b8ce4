# kernel timeline of one replayed step of any bench configuration: tools/timeline_cfg.sh <tag> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -o t -- python bench.py "$@" --steps 60 --warmup 20 --iwae-images 0 --cpu-baseline-steps 0 --probe-steps 0 --probe-warmup 0 > gpurun_out/$tag/stdout.txt 2>&1
f=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f 40 > gpurun_out/$tag/timeline.txt
tail -1 gpurun_out/$tag/stdout.txt | cut -c1-200
