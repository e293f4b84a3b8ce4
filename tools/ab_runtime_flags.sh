# un-profiled A/B of HIP / ROCclr runtime switches (environment, read at start-up) on the captured step: mean ms/step, p50
# (every run under its own timeout: ROC_SYSTEM_SCOPE_SIGNAL=0 HANGS the process -- the host never sees a completion signal)
run() { timeout 120 python bench.py --config $1 --steps 300 --warmup 20 --no-amdahl --cpu-baseline-steps 0 --iwae-images 0 --probe-steps 0 --no-graph-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['p50'])"; }
for c in ${CFGS:-c2 c1}; do
  echo -n "$c default "; run $c
  for kv in ${FLAGS:-HIP_FORCE_DEV_KERNARG=0 DEBUG_HIP_KERNARG_COPY_OPT=0 DEBUG_HIP_KERNARG_COPY_OPT=1 AMD_OPT_FLUSH=0 AMD_OPT_FLUSH=1 GPU_FLUSH_ON_EXECUTION=1 ROC_USE_FGS_KERNARG=0 ROC_USE_FGS_KERNARG=1 DEBUG_HIP_DYNAMIC_QUEUES=0 DEBUG_HIP_DYNAMIC_QUEUES=1 GPU_STREAMOPS_CP_WAIT=0 GPU_STREAMOPS_CP_WAIT=1 ROC_ACTIVE_WAIT_TIMEOUT=0 ROC_ACTIVE_WAIT_TIMEOUT=100000 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1}; do
    echo -n "$c $kv "; env $kv bash -c "$(declare -f run); run $c"
  done
  echo -n "$c default "; run $c
done
