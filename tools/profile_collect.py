#!/usr/bin/env python3
"""Copy the judged evidence of a `tools/profile_round.sh <tag>` run from gpurun_out/<tag>/ into profiles/ (tracked):
  <tag>_bench_<cfg>.json          the un-profiled bench line of every --config
  <tag>_<cfg>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `python bench.py --config <cfg>`
  <tag>_step_timeline_C<n>.txt    kernel timeline of one replayed step
  <tag>_pmc/<probe>.json          per kernel family (tools/kernel_probe.py): FETCH_SIZE / WRITE_SIZE / SQ counters of its dominant
                                  kernel, HBM bytes per launch with the gfx950 correction (MI355X_MICROARCH.md, HBM section:
                                  FETCH_SIZE counts 128-B requests as 64 B -> read bytes = 2 x FETCH_SIZE KB), matrix-pipe busy
  <tag>_pmc/<probe>_rows.csv      the raw counter rows of that kernel
  <tag>_STAMP.json                the commit (and whether the tree was clean) the files were produced at
usage: tools/profile_collect.py <tag>     (run in the repository, after the gpurun call that ran profile_round.sh)"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
pmc_dir = os.path.join(dst, tag + "_pmc")
os.makedirs(pmc_dir, exist_ok=True)


def git(*a):
    return subprocess.run(("git", "-C", ROOT) + a, capture_output=True, text=True).stdout.strip()


head = git("rev-parse", "HEAD")
SRC = ("exemplar-vae_amd", "bench.py", "tools", "include")
dirty = [l for l in git("status", "--porcelain", "--", *SRC).splitlines() if l]
# the commit the GPU box ran (tools/profile_round.sh leaves it in <tag>/HEAD): later commits that only add profiles / docs do not change it
box = os.path.join(src, "HEAD")
ran = open(box).read().strip() if os.path.exists(box) and os.path.getsize(box) > 0 else head
if ran != head:
    drift = [l for l in git("diff", "--name-only", ran, head, "--", *SRC).splitlines() if l and not l.startswith("tools/profile_")]
    dirty += ["changed since %s: %s" % (ran[:12], l) for l in drift]
    head = ran
stamp = {"commit": head, "source_tree_clean": not dirty, "uncommitted": dirty[:20]}
for f in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
    cfg = os.path.basename(f)[len("bench_"):-len(".json")]
    if os.path.getsize(f) > 10:
        d = json.load(open(f)); d["commit"] = head
        json.dump(d, open(os.path.join(dst, "%s_bench_%s.json" % (tag, cfg)), "w"))
for d in sorted(glob.glob(os.path.join(src, "stats_*"))):
    if not os.path.isdir(d):
        continue
    cfg = os.path.basename(d)[len("stats_"):]
    fs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if fs:
        shutil.copy(fs[0], os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, cfg)))
for f in sorted(glob.glob(os.path.join(src, "timeline_C*.txt"))):
    n = os.path.basename(f)[len("timeline_C"):-4]
    with open(os.path.join(dst, "%s_step_timeline_C%s.txt" % (tag, n)), "w") as o:
        o.write("# kernel timeline of one replayed step at C = %s exemplars (tools/profile_round.sh %s timeline; rocprofv3 --kernel-trace of\n"
                "# bench.py --exemplars %s; commit %s); columns: start us, end us, duration, gap to the latest end so far, hardware\n"
                "# queue, kernel; under the profiler a step is 5-10 %% longer than un-profiled\n" % (n, tag, n, head))
        o.write(open(f).read())
# which kernel of a probe is "its" kernel: the first alternative that appears in the counter rows
MAIN = {"u8fwd1": ["u8p_gemm_kernel", "u8_gemm_kernel<true>"], "u8fwd1_img": ["u8p_gemm_kernel", "u8_gemm_kernel<true>"], "u8wgrad1": ["u8_gemm_kernel<false>"],
        "fwd2_p6": ["gemm_p6_kernel<1, 128, true>"], "hdgrad2_img": ["gemm_x6_kernel<2, 0", "gemm_kernel<true, false, 2"],
        "dgrad2_p6": ["gemm_p6_kernel<9, 128, true>", "gemm_p6_kernel<9, 64, true>"], "wgrad2_p6": ["gemm_p6_kernel<3, 64, false>"],
        "hwgrad": ["narrow_wgrad_mfma_kernel", "narrow_wgrad_kernel"],
        "fwd1": ["gemm_x6_kernel<1, 0", "gemm_kernel<true, true, 1"], "fwd2": ["gemm_x6_kernel<1, 0", "gemm_kernel<true, true, 1"],
        "dgrad2": ["gemm_x6_kernel<2, 0", "gemm_kernel<true, false, 2"], "wgrad1": ["gemm_kernel<false, false, 3"],
        "wgrad2": ["gemm_kernel<false, false, 3"], "prior_iwae": ["prior_x6_lse_kernel", "prior_fwd_mfma_kernel"],
        "prior_c5": ["gemm_x6_kernel<7, 0", "gemm_kernel<true, true, 7"], "prior_train": ["prior_bwd_mfma_kernel"], "prior_train1": ["prior_train_kernel"],
        "topk_c5": ["gemm_x6_kernel<5, 0", "gemm_kernel<true, true, 5"], "topk_c2": ["gemm_x6_kernel<5, 0", "gemm_kernel<true, true, 5"],
        "conv5_fwd": ["gemm_x6_kernel<1, 1", "gemm_kernel<true, true, 1"], "conv5_bwd": ["gemm_x6_kernel<0, 1", "gemm_kernel<true, false, 0"],
        "conv96_fwd": ["gemm_x6_kernel<0, 1", "gemm_kernel<true, true, 0"],
        "cw5_fwd": ["conv_win_kernel<0, 2, 2, 320>"], "cw5_bwd": ["conv_win_kernel<1, 4, 1, 576>"], "cw5_wgrad": ["conv_wgrad_win_kernel<13, 192, 8, 1, false>"],
        "cw2_bwd": ["conv_win_kernel<1, 4, 1, 576>"], "res96_fwd": ["conv_win_kernel<3, 4, 2, 576>"], "res96_bwd": ["conv_win_kernel<4, 4, 2, 576>"],
        "res96_wgrad": ["conv_wgrad_win_kernel<9, 224, 8, 2, false>"], "cw1_fwd": ["conv_first_kernel"], "cw1_wgrad": ["conv_first_wgrad_kernel"]}
summary = {}


def rows_of(probe, grp):
    fs = glob.glob(os.path.join(src, "pmc_%s_%s" % (probe, grp), "**", "*counter_collection.csv"), recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []


def pick_kernel(probe, alts):
    for kern in alts:
        for grp in ("sq", "fetch", "write", "sq2"):
            if any(kern in r["Kernel_Name"] for r in rows_of(probe, grp)):
                return kern
    return None


for probe, alts in MAIN.items():
    kern = pick_kernel(probe, alts)
    if kern is None:
        continue
    rows_all, ctr = [], collections.defaultdict(list)
    for grp in ("fetch", "write", "sq", "sq2"):
        for r in rows_of(probe, grp):
            if kern in r["Kernel_Name"]:
                ctr[r["Counter_Name"]].append(float(r["Counter_Value"]))
                rows_all.append(r)
    if not ctr:
        continue
    c = {k: sum(v) / len(v) for k, v in ctr.items()}
    out = {"probe": "tools/kernel_probe.py %s" % probe, "rows": int(os.environ.get("EVAE_PROBE_ROWS", "19968")), "kernel": kern, "kernel_symbol": rows_all[0]["Kernel_Name"][:160], "commit": head,
           "launches_averaged": len(next(iter(ctr.values()))), "counters_per_launch": {k: round(v, 1) for k, v in sorted(c.items())}}
    if "FETCH_SIZE" in c:
        out["hbm_read_bytes_per_launch"] = round(2 * c["FETCH_SIZE"] * 1024)
        out["read_correction"] = "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> read bytes = 2 x FETCH_SIZE(KB) x 1024"
    if "WRITE_SIZE" in c:
        out["hbm_write_bytes_per_launch"] = round(c["WRITE_SIZE"] * 1024)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
    if "GRBM_GUI_ACTIVE" in c:
        act = c["GRBM_GUI_ACTIVE"] / 8.0                      # per XCD
        out["active_cycles"] = round(act); out["active_us_at_2.4GHz"] = round(act / 2400.0, 1)
        if c.get("SQ_INSTS_MFMA"):
            out["matrix_pipe_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / act, 4)      # 1024 SIMDs
            out["valu_per_mfma"] = round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], 2)
            out["lds_per_mfma"] = round(c["SQ_INSTS_LDS"] / c["SQ_INSTS_MFMA"], 2)
        if "hbm_bytes_per_launch" in out:
            out["hbm_tb_per_s_at_2.4GHz"] = round(out["hbm_bytes_per_launch"] / (act / 2400.0) / 1e6, 2)
    json.dump(out, open(os.path.join(pmc_dir, probe + ".json"), "w"), indent=1)
    if rows_all:
        with open(os.path.join(pmc_dir, probe + "_rows.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows_all[0].keys())); w.writeheader(); w.writerows(rows_all)
    summary[probe] = {k: out.get(k) for k in ("kernel", "active_us_at_2.4GHz", "matrix_pipe_busy", "valu_per_mfma", "hbm_bytes_per_launch")}
json.dump(stamp, open(os.path.join(dst, tag + "_STAMP.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
print(json.dumps(stamp))
