# step time against the exemplar count (the replicated part of a step is what is left at C = 200)
cd $GRAFT_REPO_ROOT
for c in ${CHAIN_C:-200 3125 25000}; do
  python bench.py --exemplars $c --steps 300 --warmup 40 --iwae-images 0 --cpu-baseline-steps 0 --probe-steps 0 --probe-warmup 0 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('exemplars', d['config']['exemplars'], 'ms_per_step', d['ms_per_step'], 'img/s', d['value'])"
done
