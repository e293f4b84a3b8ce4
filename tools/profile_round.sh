#!/bin/bash
# One round's evidence, run on the GPU box through gpurun:  tools/profile_round.sh <round tag, e.g. r04> [what...]
#   bench    : bench lines (un-profiled) of every --config                       -> gpurun_out/<tag>/bench_<cfg>.json
#   stats    : rocprofv3 --kernel-trace --stats of every --config                -> gpurun_out/<tag>/stats_<cfg>/
#   pmc      : FETCH_SIZE / WRITE_SIZE / SQ counter passes (one group per pass, each under its own timeout) on
#              tools/kernel_probe.py launches                                     -> gpurun_out/<tag>/pmc_<probe>_<group>/
#   timeline : kernel trace of one replayed step at C = 200 / 3125 / 25000       -> gpurun_out/<tag>/timeline_C<n>.txt
# tools/profile_collect.py <tag> then copies the judged summaries into profiles/<tag>_* and stamps them with the commit.
tag=${1:?round tag}; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
what=${@:-bench stats pmc timeline}
cd /tmp && export TMPDIR=/tmp
cd $root
CFGS=${CFGS:-c2 c2a c1 c4 c3 c5 iwae topk}
PROBES=${PROBES:-u8fwd1 u8fwd1_img fwd2_p6 hdgrad2_img dgrad2_p6 wgrad2_p6 u8wgrad1 hwgrad fwd2 dgrad2 wgrad2 prior_iwae prior_c5 prior_train prior_train1 topk_c5 topk_c2 cw5_fwd cw5_bwd cw5_wgrad cw2_bwd cw1_fwd cw1_wgrad res96_fwd res96_bwd res96_wgrad}
steps_of() { case $1 in c3) echo "--steps 20 --warmup 6";; c5) echo "--steps 20 --warmup 4";; *) echo "";; esac; }
for w in $what; do
  if [ $w = bench ]; then
    for c in $CFGS; do
      timeout 900 python bench.py --config $c $(steps_of $c) > $out/bench_${c}_stdout.txt 2> $out/bench_${c}_stderr.txt
      grep '^{' $out/bench_${c}_stdout.txt | tail -1 > $out/bench_$c.json; echo "bench $c rc=$? $(cut -c1-160 $out/bench_$c.json)"
    done
  elif [ $w = stats ]; then
    for c in $CFGS; do
      extra="--iwae-images 0 --cpu-baseline-steps 0 --no-amdahl --no-graph-profile"; [ $c = iwae ] && extra="--cpu-baseline-steps 0"
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -o $c -- python bench.py --config $c $(steps_of $c) $extra > $out/stats_${c}_stdout.txt 2>&1
      echo "stats $c rc=$?"
    done
  elif [ $w = pmc ]; then
    for p in $PROBES; do
      timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_${p}_fetch -o p -- python tools/kernel_probe.py $p 3 > /dev/null 2>&1; a=$?
      timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_${p}_write -o p -- python tools/kernel_probe.py $p 3 > /dev/null 2>&1; b=$?
      timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc_${p}_sq -o p -- python tools/kernel_probe.py $p 3 > /dev/null 2>&1; c=$?
      timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_${p}_sq2 -o p -- python tools/kernel_probe.py $p 3 > /dev/null 2>&1; d=$?
      echo "pmc $p rc=$a $b $c $d"
    done
  elif [ $w = timeline ]; then
    for n in 200 3125 25000; do
      d=$out/tl_C$n; mkdir -p $d
      timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python bench.py --exemplars $n --steps 60 --warmup 20 --iwae-images 0 --cpu-baseline-steps 0 --probe-steps 0 --probe-warmup 0 > $d/stdout.txt 2>&1
      f=$(find $d -name "*kernel_trace.csv" | head -1)
      python tools/step_timeline.py $f 40 > $out/timeline_C$n.txt; echo "timeline C=$n rc=$? $(wc -l < $out/timeline_C$n.txt) launches"
    done
  fi
done
git -C $root rev-parse HEAD > $out/HEAD 2>/dev/null || cp $root/.gpurun_head $out/HEAD 2>/dev/null || true   # (.gpurun_head: written by the caller before gpurun -- `git rev-parse HEAD > .gpurun_head` -- the box has no .git)
