# SQ counters of the split-bf16 GEMM kernel on the x6_bench launches: tools/x6_pmc.sh  (GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/x6pmc; rm -rf $out; mkdir -p $out
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/a -o p -- python tools/x6_bench.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC --output-format csv -d $out/b -o p -- python tools/x6_bench.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
for grp in "ab":
    f = glob.glob("gpurun_out/x6pmc/%s/**/*counter_collection.csv" % grp, recursive=True)
    if not f:
        print("no counter file for", grp); continue
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "gemm_x6" in r["Kernel_Name"] or "gemm_kernel<true, true, 1" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
