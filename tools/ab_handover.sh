# un-profiled A/B of the control-block hand-over (EVAE_CTL_HANDOVER=0: device-to-device copy node in front of the graph)
run() { python bench.py --steps ${STEPS:-300} --warmup 20 --no-amdahl --cpu-baseline-steps 0 --iwae-images 0 --probe-steps 0 --no-graph-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['p50'], d['step_ms'].get('max'))"; }
for r in 1 2 3 4; do for h in 0 1; do echo -n "handover=$h "; EVAE_CTL_HANDOVER=$h run; done; done
