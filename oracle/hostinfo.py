"""Host core count that respects CPU affinity and cgroup quotas (TEST/BENCH INFRASTRUCTURE ONLY).

os.cpu_count() reports the machine's logical CPUs; inside a container with a CPU quota, sizing a torch
thread pool from it oversubscribes the quota by an order of magnitude.  cpu_baseline must state the cores
it really used."""
import os


def host_cores(cap: int = 64) -> int:
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // p))
    except Exception:
        pass
    return max(1, min(n, cap))
