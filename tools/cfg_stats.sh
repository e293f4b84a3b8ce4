# per-kernel stats of one bench config: tools/cfg_stats.sh <cfg> [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
c=$1; shift
rm -rf gpurun_out/cs_$c; mkdir -p gpurun_out/cs_$c
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cs_$c -o t -- python bench.py --config $c "$@" --iwae-images 0 --cpu-baseline-steps 0 > gpurun_out/cs_$c/stdout.txt 2>&1
python - $c <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/cs_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:16]:
    print("%-84s calls %6s avg %9.1f us %5.1f%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
