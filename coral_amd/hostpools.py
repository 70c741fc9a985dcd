"""Host thread pools (OpenMP, BLAS) during a graph build.

The host side of the build is a single thread plus the few native search workers of coral_search_*; its numpy / scipy / torch
calls work on arrays of a few hundred kilobytes at most.  On a many-core host the default OpenMP / OpenBLAS pools (one thread
per core, spinning between parallel regions) burn far more CPU time than the build itself — under a container CPU quota
(cgroup ``cpu.max``) that gets the whole process throttled for tens of milliseconds at a time.  ``limited()`` caps the pools for
the duration of a build and restores them afterwards; ``CORAL_HOST_THREADS`` (default 4) sets the cap, ``0`` leaves the pools alone.
"""
from __future__ import annotations

import contextlib
import os

_controller = None


@contextlib.contextmanager
def limited():
    n = int(os.environ.get("CORAL_HOST_THREADS", "4"))
    if n <= 0:
        yield
        return
    global _controller
    import torch
    if _controller is None:
        from threadpoolctl import ThreadpoolController
        _controller = ThreadpoolController()
    before = torch.get_num_threads()
    if before > n:
        torch.set_num_threads(n)
    try:
        with _controller.limit(limits=n):
            yield
    finally:
        if before > n:
            torch.set_num_threads(before)
