"""Host-side CPU budget.  torch sizes its intra-op (OpenMP) pool by the cores it can SEE — 128 threads on the MI355X box — while
the container's cgroup grants 16 CPUs of quota per 100 ms period.  Every multi-threaded host op of a clip (the fp16 -> fp32
up-cast of the frames, dtype conversions of the latents) then wakes 128 threads that spin behind the parallel region, the
cgroup runs out of quota and the WHOLE process is frozen until the next period: round 6 measured it as 50-90 ms stalls in
arbitrary host-bound stages of roughly every second clip, with the GPU idle behind them (per-clip wall time bimodal 1 237 /
1 300 ms; `cpu.stat` nr_throttled +130 over six clips at the default pool size, +1 with 4 threads; profiles/r06/e_*).
`bound_host_threads()` caps the pool at what the quota can actually run."""
import os

import torch


def cpu_quota(root="/sys/fs/cgroup"):
    """CPUs the cgroup grants this process (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us / cpu.cfs_period_us`), or None if
    unlimited / unknown"""
    try:
        with open(os.path.join(root, "cpu.max")) as f:
            q, p = f.read().split()[:2]
        if q != "max" and float(p) > 0:
            return max(1.0, float(q) / float(p))
        return None
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(root, "cpu", "cpu.cfs_quota_us")) as f:
            q = float(f.read())
        with open(os.path.join(root, "cpu", "cpu.cfs_period_us")) as f:
            p = float(f.read())
        return max(1.0, q / p) if q > 0 and p > 0 else None
    except (OSError, ValueError):
        return None


def usable_cpus():
    """min(schedulable cores, cgroup quota), at least 1"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = cpu_quota()
    return max(1, int(min(n, q))) if q is not None else max(1, n)


_DONE = False


def bound_host_threads(limit=8, force=False):
    """Cap torch's intra-op pool at min(limit, usable_cpus() // 2) threads when it exceeds the CPUs the cgroup can run (`force`:
    whenever it exceeds the cap) — once per process, only ever downwards, and never if the application has chosen (OMP_NUM_THREADS set, or ANIP_HOST_THREADS=0; ANIP_HOST_THREADS=n picks n).  The per-clip host
    work of the pipeline is a few small conversions; half the quota leaves room for the threads' spin-wait."""
    global _DONE
    if _DONE and not force:
        return torch.get_num_threads()
    _DONE = True
    env = os.environ.get("ANIP_HOST_THREADS")
    if env is not None:
        if int(env) > 0:
            torch.set_num_threads(int(env))
        return torch.get_num_threads()
    if os.environ.get("OMP_NUM_THREADS"):
        return torch.get_num_threads()
    usable = usable_cpus()
    if torch.get_num_threads() > usable or force:      # the pool is larger than what the cgroup can run (or the application asks)
        want = max(1, min(int(limit), max(1, usable // 2)))
        if torch.get_num_threads() > want:
            torch.set_num_threads(want)
    return torch.get_num_threads()
