import torch, ctypes
x = torch.randn(1000, device="cuda")
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        y = x * 2
        x.cpu()          # illegal during capture
except Exception as e:
    print("capture failed:", type(e).__name__, str(e).splitlines()[0][:80])
for i in range(4):
    try:
        torch.cuda.synchronize()
        z = (x * 3).sum().item()
        print("attempt", i, "ok", z)
        break
    except Exception as e:
        print("attempt", i, "still failing:", str(e).splitlines()[0][:80])
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            print("  hipGetLastError ->", hip.hipGetLastError(), hip.hipGetLastError())
        except Exception as e2:
            print("  ctypes failed", e2)
