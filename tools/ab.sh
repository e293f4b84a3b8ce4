#!/bin/bash
# A/B of environment knobs on ONE box, alternating runs, one JSON line per run appended to gpurun_out/<tag>/ab.jsonl:
#   tools/ab.sh <tag> <label> "<bench args>" <reps> "ENV=.. ENV2=.." "ENV=.." ...     (an empty assignment list: "X=" )
cd $GRAFT_REPO_ROOT
tag=$1; label=$2; args=$3; reps=$4; shift 4
out=gpurun_out/$tag/ab.jsonl
mkdir -p gpurun_out/$tag
common="--iwae-images 0 --cpu-baseline-steps 0 --probe-steps 0 --probe-warmup 0"
for rep in $(seq 1 $reps); do
  for envs in "$@"; do
    line=$(env $envs python bench.py $args $common 2>gpurun_out/$tag/ab_last_stderr.txt | grep '^{' | tail -1)
    python - "$label" "$envs" "$rep" "$line" <<'PY' | tee -a $out
import json, sys
label, envs, rep, line = sys.argv[1:5]
d = json.loads(line) if line.startswith("{") else {}
print(json.dumps({"ab": label, "env": envs, "rep": int(rep), "ms_per_step": d.get("ms_per_step"), "p50": (d.get("step_ms") or {}).get("p50"),
                  "mean_loss": d.get("mean_loss"), "launch": (d.get("config") or {}).get("launch")}))
PY
  done
done
