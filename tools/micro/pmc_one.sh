# PMC passes for one probe: tools/micro/pmc_one.sh <probe>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
p=$1; out=gpurun_out/pmc1
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/${p}_sq -o p -- python tools/kernel_probe.py $p 3 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $out/${p}_sq2 -o p -- python tools/kernel_probe.py $p 3 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $out/${p}_sq3 -o p -- python tools/kernel_probe.py $p 3 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for grp in ("sq","sq2","sq3"):
    fs = glob.glob("$out/${p}_%s/**/*counter_collection.csv" % grp, recursive=True)
    if not fs: print(grp, "no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        if "gemm" in k: print(grp, k, {a: round(b/3) for a, b in d.items()})
PY
