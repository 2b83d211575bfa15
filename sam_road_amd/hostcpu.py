"""How many CPUs the host stages may use.  os.cpu_count() reports the machine (256 logical CPUs on the MI355X hosts), but a
container usually runs under a cgroup CPU quota (16 CPUs on the measured boxes): a thread pool sized for the machine burns the
quota in a few milliseconds and the kernel then throttles EVERY thread of the container for the rest of each 100 ms period.
Measured on the GPU box (profiles/r02_scene_pipeline.txt): torch's default 128 OpenMP threads got 69 s of thread time throttled
during model start-up alone; with the thread count capped to the quota the CPU-oracle legs run 1.4x faster (bench.py cpu_baseline
1.39 -> 1.92 tiles/s) and the GPU test suite, which is mostly oracle time, takes 97 s instead of 245 s."""
import os


def usable_cpus():
    """min(CPU affinity mask, cgroup v2 / v1 CPU quota), at least 1."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                     # cgroup v2: "<quota|max> <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # v1
                q, p = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def ranks_on_this_host():
    """How many processes of the job share this host's CPUs (torchrun exports LOCAL_WORLD_SIZE): the host stages of every rank run
    at the same time in the tile-sharded mode, so each rank sizes its worker pools for its share of the usable CPUs."""
    try:
        return max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        return 1


def worker_threads(cap=8):
    """Thread count for the library's short host-side worker pools (mask scans, vote accumulation): half of this rank's share of
    the usable CPUs, at most `cap`."""
    return max(1, min(cap, usable_cpus() // ranks_on_this_host() // 2))


def fill_threads(cap=16):
    """Thread count for srh_pass2_fill, the one host stage with tens of milliseconds of CPU work per scene (kNN of ~50k query rows):
    every usable CPU of this rank's share, at most `cap` (measured on a 16-CPU quota: 23.4 ms on one thread, 4.3 on 8, 2.8 on 16)."""
    return max(1, min(cap, usable_cpus() // ranks_on_this_host()))
