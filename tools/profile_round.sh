#!/bin/bash
# Round evidence, run on the GPU box through gpurun:  tools/profile_round.sh <tag>
#   1. default bench line (un-profiled)            -> gpurun_out/<tag>/bench_line.json
#   2. rocprofv3 kernel stats of the training steps -> gpurun_out/<tag>/stats/*
#   3. PMC passes on the dominant kernel (one counter group per pass, each under its own timeout)
tag=${1:-r01}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
timeout 600 python bench.py > $out/bench_stdout.txt 2>$out/bench_stderr.txt; tail -1 $out/bench_stdout.txt > $out/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python bench.py --iwae-images 0 --cpu-baseline-steps 0 > $out/bench_under_rocprof_stdout.txt 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python tools/gemm_probe.py fwd1 3 > /dev/null 2>&1; echo fetch rc=$?
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python tools/gemm_probe.py fwd1 3 > /dev/null 2>&1; echo write rc=$?
timeout 150 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_sq -o p -- python tools/gemm_probe.py fwd1 3 > /dev/null 2>&1; echo sq rc=$?
timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_sq2 -o p -- python tools/gemm_probe.py fwd1 3 > /dev/null 2>&1; echo sq2 rc=$?
find $out -type f | head -40
