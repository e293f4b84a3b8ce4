# kernel timeline of one replayed approximate-prior step at c2 sizes: tools/c2a_timeline.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2a
GRAPH=1 timeout 200 python tools/config_bench.py c2a 25000 200 2>&1 | tail -1
GRAPH=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c2a -o t -- python tools/config_bench.py c2a 25000 60 > gpurun_out/c2a/stdout.txt 2>&1
f=$(find gpurun_out/c2a -name "*kernel_trace.csv" | head -1)
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
idx = [i for i, e in enumerate(ev) if 'adam_step_kernel' in e[2]]
a, b = idx[-10], idx[-9]
step = ev[a + 1:b + 1]
t0 = ev[a][1]
print("step span %.1f us, %d kernels, busy %.1f us" % ((step[-1][1] - t0) / 1e3, len(step), sum(e[1] - e[0] for e in step) / 1e3))
prev = t0
for s, e, n in step:
    print("%8.1f +%6.1f gap %6.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n.replace('void evae::', '').replace('evae::', '')[:86]))
    prev = max(prev, e)
PY
