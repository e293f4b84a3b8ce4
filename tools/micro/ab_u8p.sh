cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
EVAE_U8_PIPE=2 EVAE_U8_TALL=2 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_p6.py -q -m gpu -k "uint8 or u8" 2>&1 | tail -5
for v in "1 1" "1 0" "0 0"; do
  set -- $v
  EVAE_U8_PIPE=$1 EVAE_U8_TALL=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_u8p_$1$2 -o p -- python tools/kernel_probe.py u8fwd1_img 20 > /dev/null 2>&1
  f=$(find gpurun_out/ab_u8p_$1$2 -name "*kernel_stats.csv" | head -1); echo "PIPE=$1 TALL=$2"; grep gemm_kernel $f | sed 's/.*P6Sink)",//'
done
for v in "1 1" "1 0" "0 0" "1 1" "0 0"; do
  set -- $v
  EVAE_U8_PIPE=$1 EVAE_U8_TALL=$2 python bench.py --steps 100 --warmup 10 --cpu-baseline-steps 0 --iwae-images 0 --probe-steps 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d.get('step_ms'))"
done
