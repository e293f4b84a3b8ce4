#!/usr/bin/env python3
"""Which replayed steps are slow?  Per-step device times (an event behind every step) of N captured c2 steps; prints the indices
and durations of steps over 3 x the median -- r06: one ~5 ms step every ~400 steps (~16 400 kernel dispatches)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
os.environ["EVAE_BENCH_DUMP_STEPS"] = "1"
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(n), "--warmup", "20", "--no-amdahl", "--cpu-baseline-steps", "0",
                    "--iwae-images", "0", "--probe-steps", "0", "--no-graph-profile"] + sys.argv[2:], capture_output=True, text=True)
line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
d = json.loads(line)
ts = d.get("per_step_ms")
med = sorted(ts)[len(ts) // 2]
out = [(i, round(t, 3)) for i, t in enumerate(ts) if t > 3 * med]
print("steps %d median %.4f ms  mean %.4f  outliers (index, ms): %s" % (len(ts), med, sum(ts) / len(ts), out))
print("gaps between outliers:", [b[0] - a[0] for a, b in zip(out, out[1:])])
