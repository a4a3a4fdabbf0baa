"""Host-side environment facts the Python layer needs before numpy is imported.

The GPU boxes of this project show 256 cores and grant the container 16 cores' worth of CPU time per 100 ms (cgroup cpu.max).
OpenBLAS / OpenMP size their worker pools by the core count; 256 spinning workers under a 16-core quota get the WHOLE process
frozen for the rest of the period - 20-45 ms freezes of the host thread that enqueues the training steps (one step in ten of a
5 ms step took 45 ms: profiles/r03_host_stalls.txt).  Importing the package therefore caps the pools at the quota unless the
caller has chosen a size; it only takes effect if numpy has not started its pools yet.

Side effect, stated: importing the package sets OPENBLAS_NUM_THREADS / OMP_NUM_THREADS / MKL_NUM_THREADS in os.environ when they
are unset (other libraries of the process and child processes inherit them).  MGR_NO_THREAD_CAP=1 turns that off.
"""
import os


def effective_cores():
    """Host cores this process may really use: affinity mask and cgroup (v2 or v1) CPU quota taken into account."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    # cgroup v2: the quota of the process's own (possibly nested) group and of every ancestor up to the mount root
    paths = ["/sys/fs/cgroup/cpu.max"]
    try:
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2].strip("/")
                while rel:
                    paths.append("/sys/fs/cgroup/%s/cpu.max" % rel)
                    rel = rel.rpartition("/")[0]
    except OSError:
        pass
    for path in paths:
        try:
            quota, period = open(path).read().split()[:2]
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
    """Rank processes that share this host: the launcher's LOCAL_WORLD_SIZE (torchrun sets it); WORLD_SIZE only where the job
    is known to be a single node (NNODES / GROUP_WORLD_SIZE = 1, or bench.py's own launcher, which sets LOCAL_WORLD_SIZE); else
    1 - a multi-node WORLD_SIZE (2 x 8 ranks) must not divide one node's quota by 16."""
    def num(v):
        try:
            return int(os.environ.get(v, ""))
        except ValueError:
            return 0
    n = num("LOCAL_WORLD_SIZE")
    if n >= 1:
        return n
    if num("WORLD_SIZE") >= 1 and 1 in (num("NNODES"), num("GROUP_WORLD_SIZE")):
        return num("WORLD_SIZE")
    return 1


def bound_thread_pools():
    if os.environ.get("MGR_NO_THREAD_CAP", "") not in ("", "0"):
        return
    n = max(1, effective_cores() // local_world_size())     # the ranks of a node share its quota
    for v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(v, str(n))
