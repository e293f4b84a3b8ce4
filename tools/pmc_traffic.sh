#!/bin/bash
# HBM traffic of the GatedDense forward GEMM (FETCH_SIZE / WRITE_SIZE in separate passes, as the guide prescribes)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py "$@" > /dev/null 2>&1; echo fetch rc=$?
timeout 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py "$@" > /dev/null 2>&1; echo write rc=$?
ls -R $out | head
