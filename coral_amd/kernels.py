"""Python wrappers over the C ABI (include/coral_hip.h).

Each public function = the local launch on this process's shard of records (``_*_local``: allocate outputs with
torch, launch on the current HIP stream) + the exchange step when records are sharded over several GPUs
(coral_amd.sharding: all-gather-v of candidate rows, all-reduce of the integer per-segment sums) + the
deterministic ordering the host logic relies on.  Integer sums and sorted rows make the results identical for
any number of shards.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

PROFILE = {}      # bench.py: PROFILE["scan_ms"] = [] collects (start, end) HIP events around every cigar_scan launch


class ScanResult:
    """Per-record CIGAR summaries of the local shard (device tensors) and the large-gap rows of ALL shards.

    gaps: int64 [K, 6] = record ordinal (global), op index of the next block, previous block end, next block start,
    first block start, last block end of that record; sorted by (record, op index) — the reference's order.
    """

    def __init__(self, mbases, qinfer, blk_first, blk_last, gaps):
        self.mbases, self.qinfer, self.blk_first, self.blk_last = mbases, qinfer, blk_first, blk_last
        self.gaps = gaps


def _scan_local(dr, min_gap: int, min_mapq: int, gap_cap: int):
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
        prof = PROFILE.get("scan_ms")
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        _lib.check(L.coral_cigar_scan(C.byref(rs), min_gap, min_mapq, mb.data_ptr(), qi.data_ptr(), b0.data_ptr(),
                                      b1.data_ptr(), gaps.data_ptr(), cnt.data_ptr(), gap_cap, dr.stream()),
                   "coral_cigar_scan")
        if prof is not None:
            e1.record(torch.cuda.current_stream(dev))
            prof.append((e0, e1))
        k = int(cnt.item()) & 0xFFFFFFFF
        if k <= gap_cap:
            break
        gap_cap = 1 << int(np.ceil(np.log2(k + 1)))
    g = gaps[:k].to(torch.int64)
    rows = torch.cat([g, b0[:n][g[:, 0]].to(torch.int64)[:, None], b1[:n][g[:, 0]].to(torch.int64)[:, None]], dim=1) \
        if k else torch.zeros((0, 6), dtype=torch.int64, device=dev)
    return mb[:n], qi[:n], b0[:n], b1[:n], rows


def cigar_scan(dr, min_gap: int = 600, min_mapq: int = 20, gap_cap: int = 1 << 16, _worker=False) -> ScanResult:
    from . import sharding
    if dr.world > 1 and not _worker:
        sharding.command(dr, ("scan", min_gap, min_mapq, gap_cap))
    mb, qi, b0, b1, rows = _scan_local(dr, min_gap, min_mapq, gap_cap)
    rows[:, 0] += dr.lo
    if dr.world > 1:
        rows = sharding.allgather_rows(dr, rows)
    g = rows.cpu().numpy()
    if len(g):
        g = g[np.lexsort((g[:, 1], g[:, 0]))]      # (record ordinal, op index): the reference's iteration order
    return ScanResult(mb, qi, b0, b1, g)


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


def _coverage_local(dr, scan: ScanResult, sg: np.ndarray) -> torch.Tensor:
    """int64 [2, S] (n_reads, n_bases) of the local shard for sorted-or-not, possibly overlapping segments."""
    L = _lib.lib()
    S = len(sg)
    dev = dr.device
    out_all = torch.zeros((2, S), dtype=torch.int64, device=dev)
    if dr.n == 0:
        return out_all
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
        out_all[:, torch.tensor(batch, device=dev)] = out
    return out_all


def segment_coverage(dr, scan: ScanResult, segs: Sequence[Tuple[int, int, int]], _worker=False):
    """(n_reads, n_bases) int64 arrays for half-open segments (tid, start, end); any order, may overlap.

    n_bases already excludes aligned non-ACGT bases (what pysam count_coverage leaves out of its four arrays).
    """
    from . import sharding
    S = len(segs)
    if S == 0:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
    sg = np.asarray(segs, dtype=np.int64).reshape(S, 3)
    if dr.world > 1 and not _worker:
        sharding.command(dr, ("coverage", sg))
    out = _coverage_local(dr, scan, sg)
    if dr.world > 1:
        out = sharding.allreduce_sum(dr, out)
    o = out.cpu().numpy()
    n_reads, n_bases = o[0].copy(), o[1].copy()
    if len(dr.h_nonacgt_rec):
        nt = dr.h_tid[dr.h_nonacgt_rec]
        npos = dr.h_nonacgt_pos
        for j in range(S):
            n_bases[j] -= int(((nt == sg[j, 0]) & (npos >= sg[j, 1]) & (npos < sg[j, 2])).sum())
    return n_reads, n_bases


def _points_local(dr, uniq: np.ndarray, pair_cap: int) -> torch.Tensor:
    """Packed (point index << 32 | LOCAL record ordinal) pairs of the local shard, unsorted."""
    L = _lib.lib()
    dev = dr.device
    if dr.n == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
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
            return pairs[:k]
        pair_cap = 1 << int(np.ceil(np.log2(k + 1)))


def point_cover(dr, points: Sequence[Tuple[int, int]], pair_cap: int = 1 << 20, _worker=False) -> List[np.ndarray]:
    """For every (tid, pos) the ordinals (file order) of the records with pos <= p < end."""
    from . import sharding
    P = len(points)
    if P == 0:
        return []
    pts = np.asarray(points, dtype=np.int64).reshape(P, 2)
    if dr.world > 1 and not _worker:
        sharding.command(dr, ("points", pts, pair_cap))
    uniq, inverse = np.unique(pts, axis=0, return_inverse=True)     # sorted by (tid, pos)
    pairs = _points_local(dr, uniq, pair_cap) + dr.lo                # record ordinal -> global
    if dr.world > 1:
        pairs = sharding.allgather_rows(dr, pairs[:, None])[:, 0]
    keys = torch.sort(pairs).values.cpu().numpy()           # (point, record ordinal) order == fetch order per point
    pt = keys >> 32
    rec = (keys & 0xFFFFFFFF).astype(np.int64)
    bounds = np.searchsorted(pt, np.arange(len(uniq) + 1))
    per_uniq = [rec[bounds[j]:bounds[j + 1]] for j in range(len(uniq))]
    return [per_uniq[j] for j in inverse.reshape(-1)]


def _bp_candidates_local(dr, T, sel, mode: int, intervals, chr_rank, cutoff: int, min_mapq: int, gap_: int, gap_mapq: int,
                         groups=None):
    """int32 [K, 13] candidate rows from coral_bp_candidates (runs on this process's GPU; the chimeric table is small and
    lives with the host logic, so this step is not sharded).  Mode 2 (``groups`` = interval index per selected read)
    additionally returns the exclusive prefix of the per-read candidate counts (int32 [n_sel + 1])."""
    from .chimeric import ChimericTable  # noqa: F401
    L = _lib.lib()
    dev = dr.device
    n_sel = T.n_reads if sel is None else len(sel)
    if n_sel == 0 or T.n_rows == 0:
        empty = np.zeros((0, 13), dtype=np.int32)
        return (empty, np.zeros(n_sel + 1, dtype=np.int32)) if mode == 2 else empty
    off, qs, qe, tid, ra, rb, strand, mapq = T.device_arrays(dev)
    ct = _lib.coral_chimeric_t(T.n_reads, off.data_ptr(), qs.data_ptr(), qe.data_ptr(), tid.data_ptr(), ra.data_ptr(),
                               rb.data_ptr(), strand.data_ptr(), mapq.data_ptr())
    # one upload for all small inputs: [selection (| interval index per read) | interval tid | start | end | chromosome ranks]
    iv = np.asarray(intervals, dtype=np.int32).reshape(-1, 3)
    parts = [] if sel is None else [np.asarray(sel, dtype=np.int32)]
    if mode == 2:
        parts.append(np.asarray(groups, dtype=np.int32))
    parts += [iv[:, 0], iv[:, 1], iv[:, 2], np.asarray(chr_rank, dtype=np.int32)]
    packed = torch.from_numpy(np.concatenate(parts)).to(dev)
    at = packed.data_ptr()
    sel_ptr = None
    if sel is not None:
        sel_ptr = at
        at += 4 * n_sel * (2 if mode == 2 else 1)
    it_ptr, is_ptr, ie_ptr, cr_ptr = at, at + 4 * len(iv), at + 8 * len(iv), at + 12 * len(iv)
    counts = torch.empty(n_sel + 2, dtype=torch.int32, device=dev)
    cap = max(1024, 2 * n_sel)
    while True:
        cand = torch.empty((cap, 13), dtype=torch.int32, device=dev)
        n_out = C.c_int32(0)
        rc = L.coral_bp_candidates(C.byref(ct), n_sel, sel_ptr, mode, len(iv), it_ptr, is_ptr, ie_ptr, cr_ptr, len(chr_rank), cutoff,
                                   min_mapq, gap_, gap_mapq, counts.data_ptr(), cand.data_ptr(), cap, C.byref(n_out),
                                   dr.stream())
        if rc == -3:                      # CORAL_ERR_CAPACITY
            cap = int(n_out.value)
            continue
        if rc == -4:                      # CORAL_ERR_FORMAT: contig outside chr1..22,X,Y,M
            raise KeyError("contig name outside chr1..22,X,Y,M")      # gn:13-18 lookup at bu:293
        _lib.check(rc, "coral_bp_candidates")
        rows = cand[:n_out.value].cpu().numpy()
        if mode == 2:
            return rows, counts[:n_sel + 1].cpu().numpy()
        return rows


def bp_candidates(dr, T, sel, mode: int, intervals, chr_rank, cutoff=100, min_mapq=20, gap_=100, gap_mapq=10):
    """Breakpoint candidates (coral_amd.chimeric.Candidates) of the chimeric reads ``sel`` (None = all, dict order)."""
    from .chimeric import Candidates
    rows = _bp_candidates_local(dr, T, sel, mode, intervals, chr_rank, cutoff, min_mapq, gap_, gap_mapq).astype(np.int64)
    return Candidates(**{k: rows[:, j] for j, k in enumerate(Candidates.FIELDS)})


def bp_candidates_grouped(dr, T, reads_per_query, first_intervals, second_interval, chr_rank, cutoff=100, min_mapq=20, gap_mapq=10):
    """Several alignment2bp queries (bu:70-96) that share their second interval, in ONE coral_bp_candidates launch (mode 2).

    ``reads_per_query[g]``: reads (indices into T, iteration order) of query g; ``first_intervals[g]`` its (tid, start, end).
    Returns one ``Candidates`` per query, rows in the reference's order."""
    from .chimeric import Candidates
    if not reads_per_query:
        return []
    sizes = [len(r) for r in reads_per_query]
    sel = np.concatenate([np.asarray(r, dtype=np.int32) for r in reads_per_query]) if sum(sizes) else np.zeros(0, dtype=np.int32)
    groups = np.repeat(np.arange(len(sizes), dtype=np.int32), sizes)
    rows, prefix = _bp_candidates_local(dr, T, sel, 2, list(first_intervals) + [second_interval], chr_rank, cutoff, min_mapq, 100,
                                        gap_mapq, groups=groups)
    rows = rows.astype(np.int64)
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    out = []
    for g in range(len(sizes)):
        a = int(prefix[bounds[g]]) if bounds[g] < len(prefix) else len(rows)
        b = int(prefix[bounds[g + 1]]) if bounds[g + 1] < len(prefix) else len(rows)
        part = rows[a:b]
        out.append(Candidates(**{k: part[:, j] for j, k in enumerate(Candidates.FIELDS)}))
    return out


def _sa_table_local(dr):
    """coral_sa_table on this process's GPU over ALL records' SA rows (they are tiny next to the CIGARs; the table is
    consumed by the host logic, so it is built where that runs).  Returns numpy arrays
    (rows int32[n_rows, 8], off int64[n_reads + 1], name_id, failed, read_length)."""
    L = _lib.lib()
    dev = dr.device
    d = dr.sa_device_arrays()
    n_sa = dr.n_sa
    ws_bytes = max(1 << 20, 96 * max(n_sa, 1) + 8 * dr.n_names + (8 << 20))
    out_rows = torch.empty((max(n_sa, 1), 8), dtype=torch.int32, device=dev)
    out_off = torch.empty(max(n_sa, 1) + 1, dtype=torch.int32, device=dev)
    out_name = torch.empty(max(n_sa, 1), dtype=torch.int32, device=dev)
    out_failed = torch.empty(max(n_sa, 1), dtype=torch.int32, device=dev)
    out_rl = torch.empty(max(dr.n_names, 1), dtype=torch.int32, device=dev)
    counts = (C.c_int32 * 2)()
    while True:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        rc = L.coral_sa_table(dr.n_total, d["tid"].data_ptr(), d["flagmq"].data_ptr(), d["qlen"].data_ptr(), d["name"].data_ptr(),
                              dr.n_names, n_sa, d["sa"].data_ptr(), d["sa_nm"].data_ptr(), d["sa_rec"].data_ptr(), ws.data_ptr(),
                              ws_bytes, out_rows.data_ptr(), out_off.data_ptr(), out_name.data_ptr(), out_failed.data_ptr(),
                              out_rl.data_ptr(), counts, dr.stream())
        if rc == -3:
            ws_bytes = (int(counts[0]) + 1) << 20
            continue
        break
    if rc == -4:
        raise KeyError("SA CIGAR shape outside SM/MS/SMS/SMD/MDS/SMDS/SMI/MIS/SMIS")        # cp:255
    if rc == -5:
        raise ZeroDivisionError("float division by zero")                                # cp:268
    if rc != 0:
        raise _lib.CoralHipError("coral_sa_table failed (%d): %s" % (rc, L.coral_sa_last_error().decode()))
    n_reads, n_rows = int(counts[0]), int(counts[1])
    # the rows leave the GPU column by column (transposed there): the host wants seven contiguous int64 columns, and cutting
    # them out of a row-major [n, 8] array costs more than the whole kernel
    cols = out_rows[:n_rows].t().contiguous().to(torch.int64).cpu().numpy()
    return (cols, out_off[:n_reads + 1].cpu().numpy().astype(np.int64),
            out_name[:n_reads].cpu().numpy().astype(np.int64), out_failed[:n_reads].cpu().numpy().astype(bool),
            out_rl[:dr.n_names].cpu().numpy().astype(np.int64))


def sa_table(dr):
    """(columns int64 [8, n_rows] = qs, qe, tid, ra, rb, strand, mapq, nm; row offsets per read; name id per read; failed flag
    per read; read length per name id) — coral_sa_table over all SA rows."""
    return _sa_table_local(dr)
