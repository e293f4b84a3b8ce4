"""How does a replayed hipGraph resolve a cross-stream dependency?  Two-stream captures of spin kernels (torch.cuda._sleep);
run under rocprofv3 --kernel-trace and read the start times:  python tools/micro/graph_deps.py <case>
  A: side = 10 x spin, event after the 3rd; main = spin, wait(event), spin(marker: longer), join
  B: as A, and the side stream waits on main's first spin in front of its 6th
  C: as A with the waiting launch captured BEFORE the side stream's 4th..10th launches (capture order)
Every kernel is a spin of a distinct length so that the trace identifies it by duration."""
import sys
import torch

case = sys.argv[1] if len(sys.argv) > 1 else "A"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
US = 2100          # cycles per microsecond, roughly (100 MHz timer x ... calibrated below by the trace itself)
spin = lambda us: torch.cuda._sleep(int(us * US))
main = torch.cuda.Stream(); side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
x = torch.zeros(1, device=dev)
torch.cuda.synchronize()
with torch.cuda.stream(main):
    spin(5); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=main):
        spin(20)                                   # main #1
        m1 = torch.cuda.Event(); m1.record()
        side.wait_stream(main)
        if case in ("A", "B"):
            with torch.cuda.stream(side):
                for i in range(10):
                    if case == "B" and i == 5:
                        side.wait_event(m1)
                    spin(10 + i)                   # side #i: 10..19 us
                    if i == 2:
                        e3 = torch.cuda.Event(); e3.record()
            main.wait_event(e3)
            spin(40)                               # the waiter
        elif case == "C":
            with torch.cuda.stream(side):
                for i in range(3):
                    spin(10 + i)
                e3 = torch.cuda.Event(); e3.record()
            main.wait_event(e3)
            spin(40)
            with torch.cuda.stream(side):
                for i in range(3, 10):
                    spin(10 + i)
        elif case == "D":                          # the waiter depends on the side stream's LAST launch before the side waits on main
            with torch.cuda.stream(side):
                for i in range(5):
                    spin(10 + i)
                e3 = torch.cuda.Event(); e3.record()
            main.wait_event(e3)
            spin(40)
            m2 = torch.cuda.Event(); m2.record()
            with torch.cuda.stream(side):
                side.wait_event(m2)
                for i in range(5, 10):
                    spin(10 + i)
        elif case == "E":                          # 50 x 1 us, one stream
            for i in range(50):
                spin(1)
        elif case == "F":                          # 2 x 25 x 1 us, two streams, no dependency between them
            with torch.cuda.stream(side):
                for i in range(25):
                    spin(1)
            for i in range(25):
                spin(1)
        elif case == "G":                          # the step's shape: thin side chain, main waits on its 3rd, side waits on main's waiter after its 12th
            with torch.cuda.stream(side):
                for i in range(12):
                    spin(5)
                    if i == 2:
                        e3 = torch.cuda.Event(); e3.record()
            spin(10)
            main.wait_event(e3)
            spin(40)
            m2 = torch.cuda.Event(); m2.record()
            spin(12)
            with torch.cuda.stream(side):
                side.wait_event(m2)
                for i in range(8):
                    spin(5)
        main.wait_stream(side)
        spin(30)                                   # tail
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print("done", case)
