"""Host-side environment facts the Python layer needs before numpy is imported.

The GPU boxes of this project show 256 cores and grant the container 16 cores' worth of CPU time per 100 ms (cgroup cpu.max).
OpenBLAS / OpenMP size their worker pools by the core count; 256 spinning workers under a 16-core quota get the WHOLE process
frozen for the rest of the period - 20-45 ms freezes of the host thread that enqueues the training steps (one step in ten of a
5 ms step took 45 ms: profiles/r03_host_stalls.txt).  Importing the package therefore caps the pools at the quota unless the
caller has chosen a size; it only takes effect if numpy has not started its pools yet.
"""
import os


def effective_cores():
    """Host cores this process may really use: affinity mask and cgroup (v2 or v1) CPU quota taken into account."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return n


def local_world_size():
    """Rank processes that share this host (torchrun's LOCAL_WORLD_SIZE, else WORLD_SIZE on a single node, else 1)."""
    for v in ("LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        try:
            n = int(os.environ.get(v, ""))
            if n >= 1:
                return n
        except ValueError:
            pass
    return 1


def bound_thread_pools():
    n = max(1, effective_cores() // local_world_size())     # the ranks of a node share its quota
    for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(v, str(n))
