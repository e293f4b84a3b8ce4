#!/bin/bash
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> <probe args...>   (run on the GPU box via gpurun)
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $out/sq2 -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py "$@" > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/fetch -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o p -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py "$@" > /dev/null 2>&1
ls -R $out | head -30
