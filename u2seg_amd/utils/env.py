"""Host-side process set-up (the role of detectron2/utils/env.py:setup_environment and of the OMP_NUM_THREADS default that
torch.distributed.run gives every rank): one process per GPU whose host work is launch submission - the intra-op thread pool
must not compete with it.

Why it matters here (measured): PyTorch sizes its OpenMP pool from the HOST's cores (128 on the MI355X
boxes) while the container's CPU controller grants 16 CPUs per 100 ms period.  A parallel region of any size - ATen's CPU
`repeat_interleave` uses a grain of 1 - wakes the whole pool, and the woken threads spin for milliseconds before they sleep:
10 CPU-seconds per wall second in the training loop.  When the container's quota runs out the kernel stops EVERY thread of the
container for the rest of the period, the launching thread included: the 25-35 ms holes between two launches."""
import os

import torch


def cgroup_cpu_quota():
    """CPUs the container may use per period (cgroup v2 cpu.max, v1 cfs quota), or None when unlimited / not readable."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            quota, period = int(fq.read()), int(fp.read())
        return quota / period if quota > 0 else None
    except (OSError, ValueError):
        return None


def configure_host_threads(max_threads=8):
    """Caps (never raises) torch's intra-op thread count at min(max_threads, half the CPU quota, the affinity mask): the other
    half stays with the launching thread, the autograd thread and the HIP runtime's own threads.  An explicit OMP_NUM_THREADS
    is respected as it is.  Returns the thread count in effect."""
    if os.environ.get("OMP_NUM_THREADS"):
        return torch.get_num_threads()
    cap = max_threads
    quota = cgroup_cpu_quota()
    if quota is not None:
        cap = min(cap, max(1, int(quota // 2)))
    try:
        cap = min(cap, max(1, len(os.sched_getaffinity(0))))
    except (AttributeError, OSError):
        pass
    if torch.get_num_threads() > cap:
        torch.set_num_threads(cap)
    return torch.get_num_threads()
