"""Host-thread budget of a rank.  One process per GPU on a 256-core node: PyTorch sizes its intra-op pool (OpenMP) to every core
it can see, and the per-step host operations of the eager paths (the candidate draw of the approximate prior -- reference
models/BaseModel.py:245,257 draws it on the host --, index arithmetic, `unique`) then wake 256 threads that spin at the region's
barrier.  Under a container CPU quota that is not just waste: r03 measurement on the bench box (cgroup cpu.max = 16 CPUs, 256
visible) -- the pool burns the 100-ms period's quota in a few milliseconds, the kernel freezes the whole cgroup for the rest of
the period, and every second or third c5 step shows a 25-80 ms stall with the GPU idle (cpu.stat: 15.4 s throttled in a 4.4 s
run).  c5 eager: 35.0 -> see DESIGN section 7 with the pool bounded.

limit_host_threads() bounds the pool to min(8, this rank's share): cores in the affinity mask, the cgroup quota (v2 cpu.max, v1
cpu.cfs_quota_us), the node's cores / ranks on the node.  An explicit OMP_NUM_THREADS or EVAE_HOST_THREADS wins."""
import math
import os

import torch

_DONE = [False]


def _cgroup_quota():
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, math.ceil(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return max(1, math.ceil(q / p))
    except (OSError, ValueError):
        pass
    return None


def cpu_budget():
    """CPUs this rank may keep busy"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = _cgroup_quota()
    if q is not None:
        n = min(n, q)
    local = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)
    return max(1, n // max(local, 1))


def limit_host_threads(cap=8):
    """Bound PyTorch's intra-op pool once per process; returns the thread count in force."""
    if _DONE[0]:
        return torch.get_num_threads()
    _DONE[0] = True
    want = os.environ.get("EVAE_HOST_THREADS")
    if want is None and os.environ.get("OMP_NUM_THREADS"):
        return torch.get_num_threads()             # the user's own setting stands
    n = int(want) if want else min(cap, cpu_budget())
    if n >= 1 and n < torch.get_num_threads() or want:
        torch.set_num_threads(max(1, n))
    return torch.get_num_threads()
