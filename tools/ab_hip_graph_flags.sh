# un-profiled A/B of the HIP runtime's graph switches (read at start-up from the environment) on the host-bound (c1) and the
# GPU-bound (c2) captured step: mean ms/step, p50
run() { python bench.py --config $1 --steps 300 --warmup 20 --no-amdahl --cpu-baseline-steps 0 --iwae-images 0 --probe-steps 0 --no-graph-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['p50'])"; }
for c in c1 c2; do
  echo -n "$c default "; run $c
  for v in 0 1; do echo -n "$c DEBUG_CLR_GRAPH_PACKET_CAPTURE=$v "; DEBUG_CLR_GRAPH_PACKET_CAPTURE=$v run $c; done
  for v in 1 2 4 8; do echo -n "$c DEBUG_HIP_FORCE_GRAPH_QUEUES=$v "; DEBUG_HIP_FORCE_GRAPH_QUEUES=$v run $c; done
  for v in 1 8 64 256; do echo -n "$c DEBUG_HIP_GRAPH_BATCH_SIZE=$v "; DEBUG_HIP_GRAPH_BATCH_SIZE=$v run $c; done
  echo -n "$c default "; run $c
done
