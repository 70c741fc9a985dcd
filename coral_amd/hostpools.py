"""Host BLAS thread pools and the graph build.

The host side of the build is a single thread plus the few native search workers of coral_search_*; its only dense linear
algebra is the CN assignment, systems of ~100 unknowns.  numpy brings an OpenBLAS pool with one thread per core
(64 on the MI355X hosts; the build itself no longer imports scipy), and every multi-threaded BLAS call — or every change of the pool size — leaves those threads spinning
for a while.  That costs far more CPU time than the build itself, and under a container CPU quota (cgroup ``cpu.max``) it gets
the whole process throttled for tens of milliseconds at a time (measured: 13 throttling events in 10 builds).  So the first
build of a process sets the BLAS pools to ONE thread, for the rest of the process (flipping the size per build is exactly what
wakes the spinners).  ``CORAL_HOST_THREADS=0`` leaves the pools alone; any other value is the BLAS thread count to set.
Exporting ``OPENBLAS_NUM_THREADS`` / ``OMP_NUM_THREADS`` before starting Python (bench.py and the CLI do) avoids creating the
threads in the first place.
"""
from __future__ import annotations

import os

_applied = None          # keeps the threadpoolctl limiter alive: its destructor would restore the old sizes


def apply_once():
    global _applied
    if _applied is not None:
        return
    n = int(os.environ.get("CORAL_HOST_THREADS", "1"))
    if n <= 0 or os.environ.get("OPENBLAS_NUM_THREADS") == str(n):
        _applied = False          # left alone on request, or the pools were created at that size already (bench.py, the CLI)
        return
    from threadpoolctl import threadpool_limits
    _applied = threadpool_limits(limits=n, user_api="blas")
