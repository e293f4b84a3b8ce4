#!/bin/bash
# A/B of the build's environment knobs on ONE box, two alternating runs each, one JSON line per run -> gpurun_out/r03/ab_knobs.jsonl
# (copied to profiles/r03_ab/knobs.jsonl).  tools/ab_knobs.sh [quick]
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03/ab_knobs.jsonl
mkdir -p gpurun_out/r03; : > $out
common="--iwae-images 0 --cpu-baseline-steps 0 --probe-steps 0 --probe-warmup 0"
run() {   # run <label> <bench args> -- <env assignments...>
  local label=$1; shift; local args=$1; shift
  for rep in 1 2; do
    for envs in "$@"; do
      line=$(env $envs python bench.py $args $common 2>/dev/null | grep '^{' | tail -1)
      python - "$label" "$envs" "$rep" "$line" >> $out <<'PY'
import json, sys
label, envs, rep, line = sys.argv[1:5]
d = json.loads(line) if line.startswith("{") else {}
print(json.dumps({"ab": label, "env": envs, "rep": int(rep), "ms_per_step": d.get("ms_per_step"), "value": d.get("value"),
                  "host_issue_ms_per_step": d.get("host_issue_ms_per_step"), "mean_loss": d.get("mean_loss"),
                  "untimed_steps": d.get("untimed_steps"), "launch": (d.get("config") or {}).get("launch")}))
PY
    done
  done
}
run "c4: heads+sample+density / loss assembly as single Functions" "--config c4" "EVAE_HVAE_FUSED_HEADS=1" "EVAE_HVAE_FUSED_HEADS=0"
run "c4: two streams" "--config c4" "EVAE_HVAE_TWO_STREAM=1" "EVAE_HVAE_TWO_STREAM=0"
run "c4: thin weight gradients as nodes on a third stream" "--config c4" "EVAE_HVAE_LEAF_STREAM=0" "EVAE_HVAE_LEAF_STREAM=1"
run "c2: narrow (heads) weight gradient as a streaming reduction" "--config c2" "EVAE_WGRAD_NARROW=1" "EVAE_WGRAD_NARROW=0"
run "c4: narrow (heads) weight gradient as a streaming reduction" "--config c4" "EVAE_WGRAD_NARROW=1" "EVAE_WGRAD_NARROW=0"
run "C=200: control block uploaded directly on the step's stream" "--exemplars 200" "EVAE_CTL_DIRECT=1" "EVAE_CTL_DIRECT=0"
run "c1: control block uploaded directly on the step's stream" "--config c1" "EVAE_CTL_DIRECT=1" "EVAE_CTL_DIRECT=0"
run "c2a: control block uploaded directly (default: staged, the block carries the 25 000-candidate draw)" "--config c2a" "EVAE_CTL_DIRECT=0" "EVAE_CTL_DIRECT=1"
run "C=200: whole step on one stream" "--exemplars 200" "EVAE_ONE_STREAM=0" "EVAE_ONE_STREAM=1"
run "c2: whole step on one stream" "--config c2" "EVAE_ONE_STREAM=0" "EVAE_ONE_STREAM=1"
run "C=200: planner's price of a finish launch" "--exemplars 200" "EVAE_PLAN_FINISH=3" "EVAE_PLAN_FINISH=6" "EVAE_PLAN_FINISH=10"
run "c2a: top-K by screening vs exact scan" "--config c2a" "EVAE_TOPK_EXACT_SCAN=0" "EVAE_TOPK_EXACT_SCAN=1"
run "c2: four leaf weight gradients grouped" "--config c2" "EVAE_GROUP_LEAVES=1" "EVAE_GROUP_LEAVES=0"
run "c2: layer-2 data gradient writes the byte layer's dy images" "--config c2" "EVAE_IMG_DGRAD=1" "EVAE_IMG_DGRAD=0"
if [ "$1" != quick ]; then
  run "c5: host threads bounded (evae/hostcpu.py) vs the 256-thread pool" "--config c5 --steps 20 --warmup 4" "EVAE_HOST_THREADS=8" "EVAE_HOST_THREADS=256"
  run "c5: eager (unique leaves ~100 exemplars) vs captured with 1000 static slots" "--config c5 --steps 20 --warmup 4" "EVAE_C5_GRAPH=0" "EVAE_C5_GRAPH=1"
fi
wc -l $out
