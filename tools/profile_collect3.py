#!/usr/bin/env python3
"""Copy the judged evidence of a tools/profile_round3.sh run from gpurun_out/r03/ into profiles/ (tracked):
  r03_bench_<cfg>.json            the un-profiled bench line of every --config
  r03_<cfg>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of `python bench.py --config <cfg>`
  r03_pmc/<probe>.json            per kernel family (tools/kernel_probe.py): FETCH_SIZE / WRITE_SIZE / SQ counters of its dominant
                                  kernel, HBM bytes per launch with the gfx950 correction (MI355X_MICROARCH.md, HBM section:
                                  FETCH_SIZE counts 128-B requests as 64 B -> read bytes = 2 x FETCH_SIZE KB), matrix-pipe busy
  r03_pmc/<probe>_rows.csv        the raw counter rows of that kernel"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r03")
dst = os.path.join(ROOT, "profiles")
os.makedirs(os.path.join(dst, "r03_pmc"), exist_ok=True)
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    cfg = os.path.basename(f)[len("bench_"):-len(".json")]
    if os.path.getsize(f) > 10:
        shutil.copy(f, os.path.join(dst, "r03_bench_%s.json" % cfg))
for d in sorted(glob.glob(os.path.join(src, "stats_*"))):
    if not os.path.isdir(d):
        continue
    cfg = os.path.basename(d)[len("stats_"):]
    fs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(dst, "r03_%s_kernel_stats.csv" % cfg))
# which kernel of a probe is "its" kernel
# which kernel of a probe is "its" kernel: the first alternative that appears in the counter rows (the split-bf16 kernel where
# the launch takes it, the fp32-MFMA kernel otherwise)
MAIN = {"u8fwd1": ["u8_gemm_kernel<true>"], "u8wgrad1": ["u8_gemm_kernel<false>"],
        "fwd1": ["gemm_x6_kernel<1, 0", "gemm_kernel<true, true, 1"], "fwd2": ["gemm_x6_kernel<1, 0", "gemm_kernel<true, true, 1"],
        "dgrad2": ["gemm_x6_kernel<2, 0", "gemm_kernel<true, false, 2"], "wgrad1": ["gemm_kernel<false, false, 3"],
        "wgrad2": ["gemm_kernel<false, false, 3"], "prior_iwae": ["prior_x6_lse_kernel", "prior_fwd_mfma_kernel"],
        "prior_c5": ["gemm_x6_kernel<7, 0", "gemm_kernel<true, true, 7"], "prior_train": ["prior_bwd_mfma_kernel"],
        "topk_c5": ["gemm_x6_kernel<5, 0", "gemm_kernel<true, true, 5"], "topk_c2": ["gemm_x6_kernel<5, 0", "gemm_kernel<true, true, 5"],
        "conv5_fwd": ["gemm_x6_kernel<1, 1", "gemm_kernel<true, true, 1"], "conv5_bwd": ["gemm_x6_kernel<0, 1", "gemm_kernel<true, false, 0"],
        "conv96_fwd": ["gemm_x6_kernel<0, 1", "gemm_kernel<true, true, 0"]}
summary = {}
def pick_kernel(probe, alts):
    for kern in alts:
        for grp in ("sq", "fetch", "write", "sq2"):
            fs = glob.glob(os.path.join(src, "pmc_%s_%s" % (probe, grp), "**", "*counter_collection.csv"), recursive=True)
            if fs and any(kern in r["Kernel_Name"] for r in csv.DictReader(open(fs[0]))):
                return kern
    return alts[0]


for probe, alts in MAIN.items():
    kern = pick_kernel(probe, alts)
    rows_all, ctr = [], collections.defaultdict(list)
    for grp in ("fetch", "write", "sq", "sq2"):
        fs = glob.glob(os.path.join(src, "pmc_%s_%s" % (probe, grp), "**", "*counter_collection.csv"), recursive=True)
        if not fs:
            continue
        for r in csv.DictReader(open(fs[0])):
            if kern in r["Kernel_Name"]:
                ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
                rows_all.append(r)
    if not ctr:
        continue
    c = {k: sum(v) / len(v) for k, v in ctr.items()}
    out = {"probe": "tools/kernel_probe.py %s" % probe, "kernel": kern, "launches_averaged": len(next(iter(ctr.values()))),
           "counters_per_launch": {k: round(v, 1) for k, v in sorted(c.items())}}
    if "FETCH_SIZE" in c:
        out["hbm_read_bytes_per_launch"] = round(2 * c["FETCH_SIZE"] * 1024)
        out["read_correction"] = "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> read bytes = 2 x FETCH_SIZE(KB) x 1024"
    if "WRITE_SIZE" in c:
        out["hbm_write_bytes_per_launch"] = round(c["WRITE_SIZE"] * 1024)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        act = c["GRBM_GUI_ACTIVE"] / 8.0                      # per XCD
        out["active_cycles"] = round(act); out["active_us_at_2.4GHz"] = round(act / 2400.0, 1)
        out["matrix_pipe_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / act, 4)      # 1024 SIMDs
        out["valu_per_mfma"] = round(c["SQ_INSTS_VALU"] / max(c["SQ_INSTS_MFMA"], 1.0), 2)
        out["lds_per_mfma"] = round(c["SQ_INSTS_LDS"] / max(c["SQ_INSTS_MFMA"], 1.0), 2)
    json.dump(out, open(os.path.join(dst, "r03_pmc", probe + ".json"), "w"), indent=1)
    if rows_all:
        with open(os.path.join(dst, "r03_pmc", probe + "_rows.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows_all[0].keys())); w.writeheader(); w.writerows(rows_all)
    summary[probe] = {k: out.get(k) for k in ("kernel", "active_us_at_2.4GHz", "matrix_pipe_busy", "hbm_bytes_per_launch")}
print(json.dumps(summary, indent=1))
# the dominant launch of the default bench: alias for bench.py's roofline.traffic
dom = os.path.join(dst, "r03_pmc", "dgrad2.json")
if os.path.exists(dom):
    shutil.copy(dom, os.path.join(dst, "r03_pmc", "c2_dominant.json"))
