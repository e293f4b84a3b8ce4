cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
EVAE_U8_PIPE=2 EVAE_U8_TALL=2 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_p6.py -q -m gpu -k "uint8 or u8" 2>&1 | tail -3
EVAE_U8_PIPE=2 EVAE_U8_TALL=0 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_p6.py -q -m gpu -k "uint8 or u8" 2>&1 | tail -3
for p in u8fwd1 u8fwd1_img; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_u8p_$p -o p -- python tools/kernel_probe.py $p 20 > /dev/null 2>&1
  f=$(find gpurun_out/ab_u8p_$p -name "*kernel_stats.csv" | head -1); echo "$p"; grep gemm_kernel $f | sed 's/.*P6Sink)",//'
  rm -f $(find gpurun_out/ab_u8p_$p -name "*kernel_trace.csv")
done
