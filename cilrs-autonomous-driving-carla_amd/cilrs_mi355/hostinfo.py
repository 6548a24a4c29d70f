"""How many host cores this process may actually keep busy.

`os.sched_getaffinity` lists every core of the machine on a shared GPU box, but the box hands a
job a CPU *share* (16 cores per GPU on this pool): torch / OpenMP pools sized by the affinity mask
oversubscribe it by an order of magnitude and every CPU-side step (the oracle in the parity tests,
bench.py's cpu_baseline, the JPEG decode pool) crawls -- the first round-2 GPU call lost its whole
20-minute budget that way.  `usable_cores()` = min(affinity, cgroup CPU quota), overridable with
CILRS_HOST_CORES (no knowledge of a particular pool: the quota is what the job was given; where
there is no quota the affinity mask is all there is to go by)."""
from __future__ import annotations

import os


def _cgroup_quota():
    try:                                              # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    try:                                              # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return max(1, q // p)
    except (OSError, ValueError):
        pass
    return None


def usable_cores(n_gpus: int | None = None) -> int:
    env = os.environ.get("CILRS_HOST_CORES")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = _cgroup_quota()
    if q is not None:
        n = min(n, q)
    return max(1, n)


def describe() -> str:
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    return f"affinity {aff} cores, cgroup quota {_cgroup_quota()} -> {usable_cores()} used"
