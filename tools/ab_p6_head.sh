# un-profiled A/B at c2: layer 2's weight images built by the step's head launch (EVAE_P6_HEAD=0: two launches on the side stream + a join)
run() { timeout 200 python bench.py --steps ${STEPS:-500} --warmup 20 --no-amdahl --cpu-baseline-steps 0 --iwae-images 0 --probe-steps 0 --no-graph-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['p50'])"; }
for r in 1 2 3; do for h in 0 1; do echo -n "p6_head=$h "; EVAE_P6_HEAD=$h run; done; done
