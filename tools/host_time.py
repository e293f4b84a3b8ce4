"""Where does the host spend a replayed step?  Wraps GraphedTrainStep._refresh and CUDAGraph.replay with timers and runs bench.py's
main():  python tools/host_time.py --exemplars 200 --iwae-images 0 --cpu-baseline-steps 0 --probe-steps 0"""
import atexit, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "exemplar-vae_amd"))
import torch
from evae import graph as G
acc = {"refresh": [0.0, 0], "replay": [0.0, 0]}
_r, _p = G.GraphedTrainStep._refresh, torch.cuda.CUDAGraph.replay


def refresh(self, *a, **k):
    t = time.perf_counter(); out = _r(self, *a, **k); acc["refresh"][0] += time.perf_counter() - t; acc["refresh"][1] += 1
    return out


def replay(self):
    t = time.perf_counter(); out = _p(self); acc["replay"][0] += time.perf_counter() - t; acc["replay"][1] += 1
    return out


G.GraphedTrainStep._refresh = refresh
torch.cuda.CUDAGraph.replay = replay
atexit.register(lambda: print("host per call: " + ", ".join("%s %.1f us x %d" % (k, 1e6 * v[0] / max(v[1], 1), v[1]) for k, v in acc.items()),
                              file=sys.stderr))
import bench
bench.main()
