#!/usr/bin/env python3
"""Copy the judged evidence of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ (tracked)."""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, "r01_bench_line.json"))
shutil.copy(os.path.join(src, "stats", tag + "_kernel_stats.csv"), os.path.join(dst, "r01_bench_kernel_stats.csv"))
with open(os.path.join(src, "bench_under_rocprof_stdout.txt")) as f:
    lines = [l for l in f if l.startswith("{")]
open(os.path.join(dst, "r01_bench_under_rocprof_stdout.txt"), "w").write(lines[-1] if lines else "")
os.makedirs(os.path.join(dst, "r01_pmc"), exist_ok=True)
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", "SQ"), ("pmc_sq2", "SQ2")):
    rows = [r for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))) if "gemm_kernel" in r["Kernel_Name"]]
    with open(os.path.join(dst, "r01_pmc", "gated_fwd_L1_%s.csv" % name), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
# the dominant launch: the row-gathered gated forward GEMM of encoder layer 1 is a kernel symbol of its own
# (template parameter GATHER_A = true), so the stats CSV lists it directly; the trace gives min / max
DOM = "gemm_kernel<true, true, 1, true, 128, 8, 0, true>"
stats = [r for r in csv.DictReader(open(os.path.join(src, "stats", tag + "_kernel_stats.csv"))) if DOM in r["Name"]]
tr = [r for r in csv.DictReader(open(os.path.join(src, "stats", tag + "_kernel_trace.csv"))) if DOM in r["Kernel_Name"]]
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
summary = {"kernel": "evae::" + DOM + " -- GatedDense forward of encoder layer 1, one launch per step",
           "launches": len(durs), "avg_us": round(sum(durs) / len(durs), 2), "min_us": round(min(durs), 2), "max_us": round(max(durs), 2),
           "kernel_stats_csv_AverageNs": float(stats[0]["AverageNs"]) if stats else None,
           "source": "rocprofv3 --kernel-trace --stats run of tools/profile_round.sh (r01_bench_kernel_stats.csv lists this symbol "
                     "on its own; launches = replayed steps + the eager probe steps of bench.py)"}
json.dump(summary, open(os.path.join(dst, "r01_dominant_launch.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
print(open(os.path.join(dst, "r01_bench_line.json")).read()[:600])
