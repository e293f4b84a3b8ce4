run() { python bench.py --steps 200 --warmup 20 --no-amdahl --cpu-baseline-steps 0 --iwae-images 0 --probe-steps 0 --no-graph-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['p50'])"; }
for s in 0 128 64 256 384 192; do echo -n "sched=$s "; EVAE_SCHED=$s run; done
echo -n "elbo_split "; EVAE_ELBO_SPLIT=1 run
echo -n "finish_group "; EVAE_FINISH_GROUP=1 run
echo -n "side_prio0 "; EVAE_SIDE_PRIORITY=0 run
echo -n "headw1 "; EVAE_HEADW_EARLY=1 run
echo -n "sched=0 "; run
