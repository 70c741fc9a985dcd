"""Breakpoint clustering and exact-breakpoint calling on candidate arrays.

Mirrors ``cluster_bp_list`` (/root/reference/src/breakpoint_utilities.py:252-286), ``bpc2bp`` (:299-388) and
``bp_match`` (:391-416).  Sums that the reference keeps in Python integers are kept exact here as well, and
every float operation is performed in the reference's order, so positions, supports and the printed
statistics are identical.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Tuple

import numpy as np

from . import _lib
from .chimeric import Candidates


def cluster_bp_list(c: Candidates, min_cluster_size, bp_distance_cutoff: int) -> List[np.ndarray]:
    """Index arrays (into ``c``) of the clusters, in the reference's order."""
    n = len(c)
    if n == 0:
        return []
    key = (c.c1 << 33) | (c.c2 << 2) | (c.o1 << 1) | c.o2          # any BAM refID (int32) — no limit on the header's size
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    group_rank = np.empty(len(first), dtype=np.int64)
    group_rank[np.argsort(first, kind="stable")] = np.arange(len(first))
    g = group_rank[inv]                                   # group id in first-seen order (bu:257-262)
    order = np.argsort(g, kind="stable")
    bounds = np.searchsorted(g[order], np.arange(len(first) + 1))
    L = _lib.lib()
    out: List[np.ndarray] = []
    for k in range(len(first)):
        idx = order[bounds[k]:bounds[k + 1]]
        if len(idx) < min_cluster_size:                   # small groups pass through as one cluster (bu:285)
            out.append(idx)
            continue
        p1 = np.ascontiguousarray(c.p1[idx])
        p2 = np.ascontiguousarray(c.p2[idx])
        cl = np.empty(len(idx), dtype=np.int32)
        ncl = C.c_int32(0)
        _lib.check(L.coral_cluster_first_fit(len(idx), p1.ctypes.data, p2.ctypes.data, int(bp_distance_cutoff),
                                             cl.ctypes.data, C.byref(ncl)), "coral_cluster_first_fit")
        o2 = np.argsort(cl, kind="stable")
        b2 = np.searchsorted(cl[o2], np.arange(ncl.value + 1))
        for j in range(ncl.value):
            out.append(idx[o2[b2[j]:b2[j + 1]]])
    return out


def _exact_sums(p: np.ndarray) -> Tuple[int, int]:
    """(Σp, Σp²) as Python integers (the reference accumulates Python ints, bu:310-314)."""
    n = len(p)
    c = int(p[0])
    d = p - c
    md = int(np.max(np.abs(d))) if n else 0
    if md < (1 << 31) and n * md * md < (1 << 62):
        s2d = int(np.dot(d, d))
    else:
        s2d = sum(int(v) * int(v) for v in d)
    sd = int(d.sum())
    return n * c + sd, s2d + 2 * c * sd + n * c * c


def _sigma(sq_mean: float, mean: float, floor=None):
    try:
        s = math.sqrt(sq_mean - mean * mean)
    except ValueError:
        return floor if floor is not None else 0
    return max(floor, s) if floor is not None else s


def _consensus(values: np.ndarray, last_is_plus: bool) -> int:
    """Unique mode, else the median rounded toward the junction side of the LAST member (bu:336-357, Q6)."""
    u, cnt = np.unique(values, return_counts=True)
    top = cnt.max()
    if len(u) == 1 or int((cnt == top).sum()) == 1:
        return int(u[np.argmax(cnt)])
    med = np.median(values)
    if len(values) % 2 == 1:
        return int(med)
    return int(math.ceil(med)) if last_is_plus else int(math.floor(med))


def bp_match_many(p1, p2, o1: int, o2: int, bp1: int, bp2: int, rgap: np.ndarray, cut: int) -> np.ndarray:
    """bp_match (bu:391-416) of many candidates that share chromosomes/orientations with the breakpoint."""
    d1 = np.abs(p1 - bp1) < cut
    d2 = np.abs(p2 - bp2) < cut
    left = rgap.astype(np.float64).copy()
    if o1 == 0:
        u1 = p1 <= bp1 - cut
        left = np.where(u1, left - (bp1 - cut - p1 + 1), left)
    else:
        u1 = p1 >= bp1 + cut
        left = np.where(u1, left - (p1 - bp1 - cut + 1), left)
    if o2 == 0:
        u2 = p2 <= bp2 - cut
        left = np.where(u2, left - (bp2 - cut - p2 + 1), left)
    else:
        u2 = p2 >= bp2 + cut
        left = np.where(u2, left - (p2 - bp2 - cut + 1), left)
    with_gap = ((u1 & (left >= 0)) | d1) & ((u2 & (left >= 0)) | d2)
    return np.where(rgap <= 0, d1 & d2, with_gap)


def bpc2bp(c: Candidates, idx: np.ndarray, cutoff: int):
    """Call the exact breakpoint of cluster ``idx``.

    Returns (p1, p2, support index array, stats list of 6 python floats, remaining index array).
    """
    p1, p2 = c.p1[idx], c.p2[idx]
    o1, o2 = int(c.o1[idx[0]]), int(c.o2[idx[0]])
    n = float(len(idx))
    s1, s11 = _exact_sums(p1)
    s2, s22 = _exact_sums(p2)
    mu1, mu2 = s1 / n, s2 / n
    sd1 = _sigma(s11 / n, mu1, cutoff / 2.99)
    sd2 = _sigma(s22 / n, mu2, cutoff / 2.99)
    keep = (p1 <= mu1 + 3 * sd1) & (p1 >= mu1 - 3 * sd1) & (p2 <= mu2 + 3 * sd2) & (p2 >= mu2 - 3 * sd2)
    bp1 = 0 if o1 == 0 else 1000000000
    bp2 = 0 if o2 == 0 else 1000000000
    if keep.any():
        last = idx[-1]
        bp1 = _consensus(p1[keep], int(c.o1[last]) == 0)
        bp2 = _consensus(p2[keep], int(c.o2[last]) == 0)
    ok = bp_match_many(p1, p2, o1, o2, bp1, bp2, c.gap[idx] * 1.2, cutoff)
    sup = idx[ok]
    if len(sup) == 0:
        return bp1, bp2, sup, [0, 0, 0, 0, 0, 0], idx[:0]
    k = float(len(sup))
    a1, a11 = _exact_sums(c.p1[sup])
    a2, a22 = _exact_sums(c.p2[sup])
    sw = c.swapped[sup] != 0
    m4 = int(np.where(sw, c.mqb[sup], c.mqa[sup]).sum())
    m5 = int(np.where(sw, c.mqa[sup], c.mqb[sup]).sum())
    st = [a1 / k, a2 / k, a11 / k, a22 / k, m4 / k, m5 / k]
    st[2] = _sigma(st[2], st[0])
    st[3] = _sigma(st[3], st[1])
    return bp1, bp2, sup, st, idx[~ok]


def call_breakpoints(c: Candidates, min_cluster_cutoff: float, bp_distance_cutoff: int, match_cutoff: int, accept_floor: float,
                     advance_subcluster: bool):
    """cluster_bp_list + the sub-cluster loop around bpc2bp in one native call (coral_call_breakpoints).

    Returns (cluster sizes, calls) with calls = [(head index, p1, p2, support index array, stats list)], in the reference's
    order; the functions above are the same algorithm step by step (kept for the unit vectors and as documentation).
    """
    n = len(c)
    if n == 0:
        return [], []
    cols = [getattr(c, k) for k in Candidates.FIELDS]
    ptrs = (C.c_void_p * 13)(*[a.ctypes.data for a in cols])
    strides = (C.c_int64 * 13)(*[a.strides[0] // 8 for a in cols])
    n_cl, n_calls = C.c_int32(0), C.c_int32(0)
    cluster_size = np.empty(n, dtype=np.int32)
    head, p1, p2 = (np.empty(n, dtype=np.int64) for _ in range(3))
    stats = np.empty(6 * n, dtype=np.float64)
    flags = np.empty(n, dtype=np.int32)
    sup_off = np.empty(n + 1, dtype=np.int64)
    sup_idx = np.empty(n, dtype=np.int64)
    _lib.check(_lib.lib().coral_call_breakpoints(
        n, ptrs, strides, float(min_cluster_cutoff), int(bp_distance_cutoff), int(match_cutoff), float(accept_floor),
        1 if advance_subcluster else 0, C.byref(n_cl), cluster_size.ctypes.data, C.byref(n_calls), head.ctypes.data,
        p1.ctypes.data, p2.ctypes.data, stats.ctypes.data, flags.ctypes.data, sup_off.ctypes.data, sup_idx.ctypes.data),
        "coral_call_breakpoints")
    calls = []
    for k in range(n_calls.value):
        st = stats[6 * k:6 * k + 6].tolist()
        if flags[k] & 1:
            st[2] = 0                                     # the reference's ValueError branch stores the integer 0
        if flags[k] & 2:
            st[3] = 0
        calls.append((int(head[k]), int(p1[k]), int(p2[k]), sup_idx[sup_off[k]:sup_off[k + 1]], st))
    return cluster_size[:n_cl.value].tolist(), calls
