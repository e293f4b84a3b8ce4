# un-profiled A/B of the merged element-wise launches (EVAE_NODE_MERGE bits: 1 broadcast in the heads' launch, 2 one Bernoulli
# launch, 4 / 8 the ELBO's assembly / the log-variance gradient's sum in the reparameterisation's launch; 0 = the r05 launch list;
# "default" = 15 for host-bound steps, 11 otherwise): mean ms/step, p50
run() { python bench.py $1 --steps 400 --warmup 20 --no-amdahl --cpu-baseline-steps 0 --iwae-images 0 --probe-steps 0 --no-graph-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['p50'])"; }
for c in ${CFGS:-"--exemplars=200" "--config=c1" "--config=c2a" "--config=c2"}; do for r in 1 2; do for m in ${MASKS:-0 default}; do
  echo -n "$c merge=$m "; if [ $m = default ]; then run "$c"; else EVAE_NODE_MERGE=$m run "$c"; fi; done; done; done
