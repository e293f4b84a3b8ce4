"""In what ORDER does the host enqueue the branches of a replayed two-branch hipGraph, and do they overlap?
python tools/micro/graph_order.py <case> [trace.csv]   -- run under rocprofv3 --kernel-trace, then again with the trace to print it.
A fork into chain A (6 long spins, 30..35 us: 'GEMMs') and chain B (20 spins of 3 us: 'thin launches'), a join, a 50-us tail.
  case = <stream of A: m|s><captured first: A|B>, e.g. mA: A on the capturing (main) stream, captured before B (on the side stream)
  a trailing 'x': chain B is captured in two halves around A (B1 A B2)."""
import sys, time
if len(sys.argv) > 2:
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows = [r for r in rows if "sleep" in r["Kernel_Name"].lower() or "spin" in r["Kernel_Name"].lower() or "delay" in r["Kernel_Name"].lower()] or rows
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tails = [i for i, r in enumerate(rows) if 40 < dur(r) < 60]
    a, b = tails[-3], tails[-2]
    step = rows[a + 1:b + 1]
    t0 = int(step[0]["Start_Timestamp"])
    step.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = []
    for r in step:
        d = dur(r)
        lab = "A" if 22 < d < 40 else ("B" if d < 8 else ("root" if d < 14 else "tail"))
        out.append("%s q%s %.0f+%.0f" % (lab, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, d))
    print(sys.argv[1], "span %.0f us | " % ((int(max(step, key=lambda r: int(r["End_Timestamp"]))["End_Timestamp"]) - t0) / 1e3) + "  ".join(out))
    sys.exit(0)
import torch
case = sys.argv[1]
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
US = 2100
spin = lambda us: torch.cuda._sleep(int(us * US))
main = torch.cuda.Stream(); side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
a_stream = main if case[0] == "m" else side
b_stream = side if case[0] == "m" else main


def chain_a():
    with torch.cuda.stream(a_stream):
        for i in range(6):
            spin(30 + i)


def chain_b(n=20):
    with torch.cuda.stream(b_stream):
        for i in range(n):
            spin(3)


torch.cuda.synchronize()
with torch.cuda.stream(main):
    spin(5); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=main):
        spin(10)                                  # root
        side.wait_stream(main)
        if case.endswith("x"):
            chain_b(10); chain_a(); chain_b(10)
        elif case[1] == "A":
            chain_a(); chain_b()
        else:
            chain_b(); chain_a()
        main.wait_stream(side)
        spin(50)                                  # tail
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200):
    g.replay()
torch.cuda.synchronize()
print("case %s: %.1f us per replay (serial sum = 10 + 195 + 60 + 50 = 315 us, perfect overlap = 255)" % (case, (time.perf_counter() - t) / 200 * 1e6))
