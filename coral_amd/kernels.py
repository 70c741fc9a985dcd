"""Thin Python wrappers over the C ABI (include/coral_hip.h): allocate outputs with torch, launch on the
current HIP stream, give results the deterministic order the host logic expects."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .records import DeviceRecords


class ScanResult:
    """Per-record CIGAR summaries (device tensors) and the large-gap rows (host, reference order)."""

    def __init__(self, mbases, qinfer, blk_first, blk_last, gaps):
        self.mbases, self.qinfer, self.blk_first, self.blk_last = mbases, qinfer, blk_first, blk_last
        self.gaps = gaps      # int32 [K, 4]: record, op index, previous block end, next block start


def cigar_scan(dr: DeviceRecords, min_gap: int = 600, min_mapq: int = 20, gap_cap: int = 1 << 16) -> ScanResult:
    L = _lib.lib()
    dev = dr.device
    n = dr.n
    mb = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    qi = torch.empty_like(mb)
    b0 = torch.empty_like(mb)
    b1 = torch.empty_like(mb)
    rs = dr.c_struct()
    while True:
        gaps = torch.empty((gap_cap, 4), dtype=torch.int32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.coral_cigar_scan(C.byref(rs), min_gap, min_mapq, mb.data_ptr(), qi.data_ptr(), b0.data_ptr(),
                                      b1.data_ptr(), gaps.data_ptr(), cnt.data_ptr(), gap_cap, dr.stream()),
                   "coral_cigar_scan")
        k = int(cnt.item()) & 0xFFFFFFFF
        if k <= gap_cap:
            break
        gap_cap = 1 << int(np.ceil(np.log2(k + 1)))
    g = gaps[:k].cpu().numpy()
    if k:
        order = np.lexsort((g[:, 1], g[:, 0]))       # (record ordinal, op index): the reference's iteration order
        g = g[order]
    return ScanResult(mb[:n], qi[:n], b0[:n], b1[:n], g)


def _disjoint_batches(segs: np.ndarray) -> List[np.ndarray]:
    """Split segment indices into batches that are sorted by (tid, start) and pairwise disjoint."""
    order = np.lexsort((segs[:, 1], segs[:, 0]))
    batches: List[List[int]] = []
    last_end: List[Tuple[int, int]] = []
    for i in order:
        t, s, e = segs[i]
        for b, (lt, le) in enumerate(last_end):
            if lt != t or le <= s:
                batches[b].append(i)
                last_end[b] = (t, e)
                break
        else:
            batches.append([i])
            last_end.append((t, e))
    return [np.array(b, dtype=np.int64) for b in batches]


def segment_coverage(dr: DeviceRecords, scan: ScanResult, segs: Sequence[Tuple[int, int, int]]):
    """(n_reads, n_bases) int64 arrays for half-open segments (tid, start, end); any order, may overlap.

    n_bases already excludes aligned non-ACGT bases (what pysam count_coverage leaves out of its four arrays).
    """
    L = _lib.lib()
    S = len(segs)
    n_reads = np.zeros(S, dtype=np.int64)
    n_bases = np.zeros(S, dtype=np.int64)
    if S == 0 or dr.n == 0:
        return n_reads, n_bases
    sg = np.asarray(segs, dtype=np.int64).reshape(S, 3)
    dev = dr.device
    rs = dr.c_struct()
    strad = torch.empty(dr.n, dtype=torch.int32, device=dev)
    for batch in _disjoint_batches(sg):
        b = sg[batch]
        t = torch.tensor(b[:, 0], dtype=torch.int32, device=dev)
        s = torch.tensor(b[:, 1], dtype=torch.int32, device=dev)
        e = torch.tensor(b[:, 2], dtype=torch.int32, device=dev)
        out = torch.zeros((2, len(batch)), dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.coral_segment_coverage(C.byref(rs), scan.mbases.data_ptr(), scan.qinfer.data_ptr(), len(batch),
                                            t.data_ptr(), s.data_ptr(), e.data_ptr(), out[0].data_ptr(),
                                            out[1].data_ptr(), strad.data_ptr(), cnt.data_ptr(), dr.stream()),
                   "coral_segment_coverage")
        o = out.cpu().numpy()
        n_reads[batch] = o[0]
        n_bases[batch] = o[1]
    if len(dr.h_nonacgt_rec):
        nt = dr.h_tid[dr.h_nonacgt_rec]
        npos = dr.h_nonacgt_pos
        for j in range(S):
            n_bases[j] -= int(((nt == sg[j, 0]) & (npos >= sg[j, 1]) & (npos < sg[j, 2])).sum())
    return n_reads, n_bases


def point_cover(dr: DeviceRecords, points: Sequence[Tuple[int, int]], pair_cap: int = 1 << 20) -> List[np.ndarray]:
    """For every (tid, pos) the ordinals (file order) of the records with pos <= p < end."""
    L = _lib.lib()
    P = len(points)
    if P == 0:
        return []
    pts = np.asarray(points, dtype=np.int64).reshape(P, 2)
    uniq, inverse = np.unique(pts, axis=0, return_inverse=True)     # sorted by (tid, pos)
    dev = dr.device
    t = torch.tensor(uniq[:, 0], dtype=torch.int32, device=dev)
    p = torch.tensor(uniq[:, 1], dtype=torch.int32, device=dev)
    rs = dr.c_struct()
    while True:
        pairs = torch.empty(pair_cap, dtype=torch.int64, device=dev)
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.coral_point_cover(C.byref(rs), len(uniq), t.data_ptr(), p.data_ptr(), pairs.data_ptr(),
                                       cnt.data_ptr(), pair_cap, dr.stream()), "coral_point_cover")
        k = int(cnt.item()) & 0xFFFFFFFF
        if k <= pair_cap:
            break
        pair_cap = 1 << int(np.ceil(np.log2(k + 1)))
    keys = torch.sort(pairs[:k]).values.cpu().numpy()       # (point, record ordinal) order == fetch order per point
    pt = keys >> 32
    rec = (keys & 0xFFFFFFFF).astype(np.int64)
    bounds = np.searchsorted(pt, np.arange(len(uniq) + 1))
    per_uniq = [rec[bounds[j]:bounds[j + 1]] for j in range(len(uniq))]
    return [per_uniq[j] for j in inverse.reshape(-1)]
