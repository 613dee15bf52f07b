"""Host-side environment guard for measurements.

On a container whose CPU quota (cgroup `cpu.max`) is far below the number of visible cores, the
OpenMP pool behind PyTorch's CPU ops starts one thread per VISIBLE core; after every parallel
region they spin, the cgroup runs out of quota and the whole process — including the thread that
launches the BA kernels — is frozen for the rest of the 100 ms period.  Measured on the MI355X
box (256 cores visible, quota 16): 60–90 ms stalls every few frames of the replayed sequence,
update() 9 ms instead of 1.7 ms.  `limit_host_threads()` caps the pool at the quota.
"""
import os


def cpu_quota():
    """CPUs this process may use: min(visible cores, cgroup v2/v1 quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def limit_host_threads(cap=None):
    """Cap PyTorch's intra-op CPU pool at the CPU quota (or `cap`); returns the value set."""
    import torch
    n = cpu_quota() if cap is None else max(1, min(int(cap), cpu_quota()))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()
