cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tk; mkdir -p gpurun_out/tk
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tk -o t -- python tools/kernel_probe.py topk_c5 10 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tk/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("%-64s calls %4s avg %8.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
