"""Does launching a graph exec again while its previous launch is still running block the host?  And do two execs of the same
capture, alternated, let the host run ahead?   python tools/micro/graph_relaunch.py"""
import time
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
spin = lambda us: torch.cuda._sleep(int(us * 2100))
st = torch.cuda.Stream()


def capture(n, us):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        spin(5); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                spin(us)
    return g


for n, us in ((50, 10), (10, 50), (50, 2)):
    g1, g2 = capture(n, us), capture(n, us)
    for mode in ("one exec", "two execs alternated"):
        with torch.cuda.stream(st):
            for _ in range(5):
                g1.replay(); g2.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40):
                (g1 if (mode == "one exec" or i % 2 == 0) else g2).replay()
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
        print("%2d kernels x %2d us, %-22s: issue %.1f us / replay, total %.1f us / replay" % (n, us, mode, 1e6 * t_issue / 40, 1e6 * t_all / 40))
