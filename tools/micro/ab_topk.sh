cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -k "topk or knn or nearest" -x 2>&1 | tail -5
for v in 1 0; do for p in topk_c5 topk_c2; do
  EVAE_TOPK_TWO_LAUNCH=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_topk_${p}_$v -o p -- python tools/kernel_probe.py $p 20 > /dev/null 2>&1
  f=$(find gpurun_out/ab_topk_${p}_$v -name "*kernel_stats.csv" | head -1); echo "TWO=$v $p"; head -8 $f | cut -d, -f1-4 | cut -c1-60,200-
  rm -f $(find gpurun_out/ab_topk_${p}_$v -name "*kernel_trace.csv")
done; done
