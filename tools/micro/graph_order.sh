# tools/micro/graph_order.sh: every case of graph_order.py un-profiled (time per replay) and under the kernel trace (dispatch order)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gorder
for c in mA mB sA sB mAx sAx; do
  python tools/micro/graph_order.py $c 2>/dev/null | tail -1
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gorder/$c -o t -- python tools/micro/graph_order.py $c > /dev/null 2>&1
  python tools/micro/graph_order.py $c $(find gpurun_out/gorder/$c -name "*kernel_trace.csv" | head -1)
done
