# PMC passes over one case of tools/micro/cw_bench: tools/micro/cw_pmc.sh <case> [images]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
c=$1; n=${2:-20224}; out=gpurun_out/cw_pmc; export CW_NOABL=1
mkdir -p $out
timeout 200 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/${c}_sq -o p -- tools/micro/cw_bench $n 1 $c > /dev/null 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $out/${c}_sq2 -o p -- tools/micro/cw_bench $n 1 $c > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $out/${c}_sq3 -o p -- tools/micro/cw_bench $n 1 $c > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for grp in ("sq","sq2","sq3"):
    fs = glob.glob("$out/${c}_%s/**/*counter_collection.csv" % grp, recursive=True)
    if not fs: print(grp, "no file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen=collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        if "conv_w" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
    for k, d in acc.items():
        nd = len(seen[k])
        print(grp, k, "dispatches", nd, {a: round(b/nd) for a, b in d.items()})
PY
