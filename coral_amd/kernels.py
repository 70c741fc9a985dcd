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
    """Per-record CIGAR summaries of the local shard (device tensor ``summary`` int32 [n, 4] = aligned bases, inferred read
    length, first block start, last block end; the four columns are also available by name) and the large-gap rows of ALL
    shards.

    gaps: int64 [K, 6] = record ordinal (global), op index of the next block, previous block end, next block start,
    first block start, last block end of that record; sorted by (record, op index) — the reference's order.  On one GPU the
    rows are fetched from the device the first time they are asked for, so the launch itself never waits for the kernel.
    """

    def __init__(self, summary, gaps=None, pending=None):
        self.summary = summary
        self.mbases, self.qinfer, self.blk_first, self.blk_last = (summary[:, k] for k in range(4))
        self._gaps, self._pending = gaps, pending

    @property
    def gaps(self):
        if self._gaps is None:
            self._gaps = self._pending()
            self._pending = None
        return self._gaps


def _scan_local(dr, min_gap: int, min_mapq: int, gap_cap: int):
    """Launch coral_cigar_scan on the local shard; returns the summary tensor (int32 [n, 4]) and a callable that yields the gap
    rows (int64 [K, 6] device tensor, LOCAL record ordinals) — calling it is the first point that waits for the kernel."""
    L = _lib.lib()
    dev = dr.device
    n = dr.n
    summary = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
    rs = dr.c_struct()

    def launch(cap):
        gaps = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        cnt = torch.zeros(2, dtype=torch.int32, device=dev)      # [0] gap rows found, [1] the kernel's work cursor
        prof = PROFILE.get("scan_ms")
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        _lib.check(L.coral_cigar_scan(C.byref(rs), min_gap, min_mapq, summary.data_ptr(), gaps.data_ptr(), cnt.data_ptr(), cap,
                                      dr.stream()), "coral_cigar_scan")
        if prof is not None:
            e1.record(torch.cuda.current_stream(dev))
            prof.append((e0, e1))
        return gaps, cnt

    state = [launch(gap_cap), gap_cap]

    def staged_copy():
        """Right behind the kernel, without waiting for it: the whole gap-row buffer, the first / last block of every row's
        record and the counter go to pinned host memory on the copy stream — by the time the host logic asks for the rows
        (find_smalldel_breakpoints, after the interval search) they have long landed."""
        (gaps, cnt), cap = state
        if dev.type != "cuda" or n == 0:
            return None
        idx = gaps[:, 0].clamp(0, n - 1).to(torch.int64)              # (rows beyond the count hold garbage: clamped, never read)
        extra = summary[:n][idx][:, 2:4].contiguous()
        st = Staged(dr)
        return st, st.start("gaps", dict(gaps=gaps, extra=extra, cnt=cnt))
    early = staged_copy() if dr.world == 1 else None

    def rows():
        if early is not None:
            st, h = early
            st.wait("gaps")
            k = int(h["cnt"][0]) & 0xFFFFFFFF
            if k <= state[1]:
                g = np.concatenate([h["gaps"][:k].astype(np.int64), h["extra"][:k].astype(np.int64)], axis=1) if k else np.zeros((0, 6), dtype=np.int64)
                st.close()
                return torch.from_numpy(g)
            st.close()
        while True:
            (gaps, cnt), cap = state
            k = int(cnt[0].item()) & 0xFFFFFFFF
            if k <= cap:
                break
            cap = 1 << int(np.ceil(np.log2(k + 1)))          # more gap rows than slots: run again with room for all of them
            state[0], state[1] = launch(cap), cap
        g = gaps[:k].to(torch.int64)
        return torch.cat([g, summary[:n][g[:, 0]][:, 2:4].to(torch.int64)], dim=1) if k else torch.zeros((0, 6), dtype=torch.int64, device=dev)

    return summary[:n], rows


def cigar_scan(dr, min_gap: int = 600, min_mapq: int = 20, gap_cap: int = 1 << 16, _worker=False) -> ScanResult:
    from . import sharding
    if dr.world > 1 and not _worker:
        sharding.command(dr, sharding.CMD_SCAN, (min_gap, min_mapq, gap_cap))
    summary, pending = _scan_local(dr, min_gap, min_mapq, gap_cap)

    def gather():
        rows = pending()
        if dr.lo:
            rows[:, 0] += dr.lo
        if dr.world > 1:
            rows = sharding.allgather_rows(dr, rows)
        g = rows.cpu().numpy()
        if len(g):
            g = g[np.lexsort((g[:, 1], g[:, 0]))]      # (record ordinal, op index): the reference's iteration order
        return g

    if dr.world > 1:
        return ScanResult(summary, gaps=gather())      # the exchange is a collective: every rank takes part now
    return ScanResult(summary, pending=gather)


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
        tse = torch.from_numpy(np.ascontiguousarray(sg[batch].T, dtype=np.int32)).to(dev)      # one upload: rows = contig, start, end
        out = torch.zeros((2, len(batch) + 1), dtype=torch.int64, device=dev)                  # (+ one column: the straddler counter)
        cnt = out[0, len(batch):].view(torch.int32)
        _lib.check(L.coral_segment_coverage(C.byref(rs), scan.summary.data_ptr(), len(batch),
                                            tse[0].data_ptr(), tse[1].data_ptr(), tse[2].data_ptr(), out[0].data_ptr(),
                                            out[1].data_ptr(), strad.data_ptr(), cnt.data_ptr(), dr.stream()),
                   "coral_segment_coverage")
        out_all[:, torch.from_numpy(batch).to(dev)] = out[:, :len(batch)]
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
        sharding.command(dr, sharding.CMD_COVERAGE, payload=sg)
    out = _coverage_local(dr, scan, sg)
    if dr.world > 1:
        out = sharding.allreduce_sum(dr, out)
    o = out.cpu().numpy()
    n_reads, n_bases = o[0].copy(), o[1].copy()
    if len(dr.h_nonacgt_rec):
        # aligned non-ACGT bases inside each segment: two binary searches per segment in the (contig, position)-sorted list of
        # those bases (sorted once per records object) instead of a pass over the list per segment
        keys = getattr(dr, "_nonacgt_keys", None)
        if keys is None:
            keys = dr._nonacgt_keys = np.sort((dr.h_tid[dr.h_nonacgt_rec].astype(np.int64) << 32) | dr.h_nonacgt_pos.astype(np.int64))
        clip = lambda v: np.clip(v, 0, (1 << 32) - 1)          # (positions are non-negative int32: clamping keeps the comparison)
        lo = np.searchsorted(keys, (sg[:, 0] << 32) + clip(sg[:, 1]), side="left")
        hi = np.searchsorted(keys, (sg[:, 0] << 32) + clip(sg[:, 2]), side="left")
        n_bases -= np.maximum(hi - lo, 0)
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
        _lib.check(L.coral_point_cover(C.byref(rs), len(uniq), t.data_ptr(), p.data_ptr(), dr.max_span, pairs.data_ptr(),
                                       cnt.data_ptr(), pair_cap, dr.stream()), "coral_point_cover")
        k = int(cnt.item()) & 0xFFFFFFFF
        if k <= pair_cap:
            return pairs[:k]
        pair_cap = 1 << int(np.ceil(np.log2(k + 1)))


class PointCover:
    """Result of ``point_cover``: for point j the record ordinals (file order) covering it are ``rec[begin[j]:end[j]]`` — ONE
    int32 array for all points (equal points share their slice); ``cover[j]`` gives the slice."""

    def __init__(self, rec: np.ndarray, begin: np.ndarray, end: np.ndarray, keep=None):
        self.rec, self.begin, self.end, self._keep = rec, begin, end, keep

    def __len__(self):
        return len(self.begin)

    def __getitem__(self, j):
        return self.rec[self.begin[j]:self.end[j]]

    def __iter__(self):
        return (self[j] for j in range(len(self)))


def point_cover(dr, points: Sequence[Tuple[int, int]], pair_cap: int = 0, _worker=False) -> PointCover:
    """For every (tid, pos) the ordinals (file order) of the records with pos <= p < end."""
    from . import sharding
    P = len(points)
    if P == 0:
        return PointCover(np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    pts = np.asarray(points, dtype=np.int64).reshape(P, 2)
    if not pair_cap:
        pair_cap = getattr(dr, "_pair_cap_hint", 1 << 20)       # (what the last call on these records needed: one launch, not two)
    if dr.world > 1 and not _worker:
        sharding.command(dr, sharding.CMD_POINTS, (pair_cap,), payload=pts)
    # distinct points, sorted by (tid, pos): one int64 key per point
    key = (pts[:, 0] << 32) | (pts[:, 1] & 0xFFFFFFFF) if bool((pts[:, 1] >= 0).all()) else None
    if key is not None:
        ukey, inverse = np.unique(key, return_inverse=True)
        uniq = np.stack([ukey >> 32, ukey & 0xFFFFFFFF], axis=1)
    else:
        uniq, inverse = np.unique(pts, axis=0, return_inverse=True)
    inverse = inverse.reshape(-1)
    pairs = _points_local(dr, uniq, pair_cap) + dr.lo                # record ordinal -> global
    if pairs.numel() > pair_cap // 2 or pairs.numel() < pair_cap // 8:
        dr._pair_cap_hint = max(1 << 16, 1 << int(np.ceil(np.log2(2 * pairs.numel() + 2))))
    if dr.world > 1:
        pairs = sharding.allgather_rows(dr, pairs[:, None])[:, 0]
    keys = torch.sort(pairs).values                                   # (point, record ordinal) order == fetch order per point
    # per-point slices are computed where the pairs are; only the int32 record ordinals and the small bounds cross to the host
    bounds = torch.searchsorted(keys >> 32, torch.arange(len(uniq) + 1, dtype=torch.int64, device=keys.device))
    rec32 = (keys & 0xFFFFFFFF).to(torch.int32)
    keep = None
    if rec32.is_cuda:
        st = Staged(dr)
        h = st.start("cover", dict(rec=rec32, bounds=bounds))
        st.wait("cover")
        rec, b, keep = h["rec"], h["bounds"], st
    else:
        rec, b = rec32.numpy(), bounds.numpy()
    return PointCover(rec, b[:-1][inverse], b[1:][inverse], keep)


_SIDE_STREAMS: dict = {}


def _side_stream(device, role: str):
    """The process's ONE copy stream / ONE table stream per device (not one per records object): a process has few hardware
    queues (ROCm's default is four per device) and streams beyond them share queues — which silently serialises work that is
    meant to overlap (measured: the BAM decoder's inflate, checksum and parse streams ran 0.6 s slower per 2 M-read file in a
    process that had made side streams for two records objects before)."""
    key = (str(device), role)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class _PinnedPool:
    """Page-locked staging buffers for the device -> host copies of one build (the chimeric table and the pair table, tens of
    MB at 2 M reads): a pageable copy of that size costs several ms on the critical path, a pinned asynchronous one overlaps
    with the CIGAR scan.  A buffer is leased to ONE ChimericTable for that table's lifetime (the table's numpy arrays are views
    of it) and re-used afterwards, so consecutive builds alternate between two sets instead of pinning memory every time."""
    _free: dict = {}

    @classmethod
    def lease(cls, device, nbytes: int) -> torch.Tensor:
        if torch.device(device).type != "cuda":
            return torch.empty(max(nbytes, 1), dtype=torch.uint8)
        pool = cls._free.setdefault(str(device), [])
        fits = [k for k, t in enumerate(pool) if nbytes <= t.numel() <= max(4 * nbytes, 1 << 20)]
        if fits:
            return pool.pop(min(fits, key=lambda k: pool[k].numel()))
        return torch.empty(max(int(nbytes * 1.25), 1 << 16), dtype=torch.uint8, pin_memory=True)

    @classmethod
    def release(cls, device, t: torch.Tensor):
        if t.is_pinned():
            pool = cls._free.setdefault(str(device), [])
            if len(pool) < 16:
                pool.append(t)


class Staged:
    """Host arrays of one coral_sa_table + coral_bp_pair_table run, landing asynchronously in leased pinned memory."""

    def __init__(self, dr):
        self.device = dr.device
        self._leases = []
        self._side = None
        if dr.device.type == "cuda":
            self._side = _side_stream(dr.device, "copy")
        self.events = {}

    def start(self, name: str, tensors: dict):
        """Copy the device tensors to pinned host arrays on the copy stream (after everything queued so far on the records'
        stream); ``wait(name)`` blocks until that group has landed.  Returns {key: numpy view}."""
        out = {}
        if self._side is None:
            return {k: t.numpy() for k, t in tensors.items()}
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            for k, t in tensors.items():
                nbytes = t.numel() * t.element_size()
                buf = _PinnedPool.lease(self.device, nbytes)
                self._leases.append(buf)
                host = buf[:nbytes].view(t.dtype).view(t.shape)
                host.copy_(t, non_blocking=True)
                t.record_stream(self._side)
                out[k] = host.numpy()
            ev = torch.cuda.Event()
            ev.record(self._side)
        self.events[name] = ev
        return out

    def wait(self, name: str):
        ev = self.events.pop(name, None)
        if ev is not None:
            ev.synchronize()

    def close(self):
        for ev in self.events.values():
            ev.synchronize()
        self.events = {}
        for buf in self._leases:
            _PinnedPool.release(self.device, buf)
        self._leases = []

    def __del__(self):
        try:
            self.close()
        except Exception:          # noqa: BLE001 — interpreter shutdown
            pass


def _sa_table_local(dr):
    """coral_sa_table (K3) + coral_bp_pair_table (K4) on this process's GPU over ALL records' SA rows (they are tiny next to
    the CIGARs; the table is consumed by the host logic, so it is built where that runs).  Returns
    (cols int64[8, n_rows], off int64[n_reads + 1], name_id, failed, read_length (device int32, or a host array), pairs int32[2 * n_rows, 8], device rows,
    staging): the host arrays live in ``staging``'s pinned buffers; everything but ``pairs`` has landed on return, ``pairs``
    after ``staging.wait("pairs")``."""
    d = dr.sa_device_arrays()
    if dr.device.type != "cuda":
        return _sa_table_on_current_stream(dr, d)
    # on a stream of its own: the table kernels are tiny and have three host round trips, the CIGAR scan launched just before
    # on the records' stream keeps the GPU busy meanwhile
    side = _side_stream(dr.device, "table")
    side.wait_stream(torch.cuda.current_stream(dr.device))
    with torch.cuda.stream(side):
        return _sa_table_on_current_stream(dr, d)


def _sa_table_on_current_stream(dr, d):
    import os
    import sys
    import time
    trace = os.environ.get("CORAL_TRACE_OPEN") == "1"
    tt = [time.perf_counter()]

    def lap(what):
        if trace:
            tt.append(time.perf_counter())
            sys.stderr.write("  sa_table: %-22s %.1f ms\n" % (what, (tt[-1] - tt[-2]) * 1e3))
    L = _lib.lib()
    dev = dr.device
    n_sa = dr.n_sa
    lap("enter")
    ws_bytes = max(1 << 20, 104 * max(n_sa, 1) + 8 * dr.n_names + (8 << 20))
    out_rows = torch.empty((max(n_sa, 1), 8), dtype=torch.int32, device=dev)
    out_off = torch.empty(max(n_sa, 1) + 1, dtype=torch.int32, device=dev)
    out_name = torch.empty(max(n_sa, 1), dtype=torch.int32, device=dev)
    out_failed = torch.empty(max(n_sa, 1), dtype=torch.int32, device=dev)
    out_rl = torch.empty(max(dr.n_names, 1), dtype=torch.int32, device=dev)
    counts = (C.c_int32 * 2)()
    lap("output tensors")
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
    lap("coral_sa_table")
    if rc == -4:
        raise KeyError("SA CIGAR shape outside SM/MS/SMS/SMD/MDS/SMDS/SMI/MIS/SMIS")        # cp:255
    if rc == -5:
        raise ZeroDivisionError("float division by zero")                                # cp:268
    if rc != 0:
        raise _lib.CoralHipError("coral_sa_table failed (%d): %s" % (rc, L.coral_sa_last_error().decode()))
    n_reads, n_rows = int(counts[0]), int(counts[1])
    pairs = pair_table(dr, out_off, out_rows, n_reads, n_rows)
    lap("pair table issue")
    # the rows leave the GPU column by column (transposed and widened there): the host wants seven contiguous int64 columns,
    # and cutting them out of a row-major [n, 8] array costs more than the whole kernel
    st = Staged(dr)
    h = st.start("table", dict(cols=out_rows[:n_rows].t().contiguous().to(torch.int64), off=out_off[:n_reads + 1].to(torch.int64),
                               name=out_name[:n_reads].to(torch.int64), failed=out_failed[:n_reads].to(torch.bool)))
    lap("staged copy: table")
    hp = st.start("pairs", dict(pairs=pairs[:2 * n_rows]))
    lap("staged copy: pairs")
    st.wait("table")
    lap("wait for the table")
    # read_length per name id (2 M entries at config 3) stays on the device: the build never looks at it — it backs the lazily
    # filled `read_length` dict of the object surface (ibg:141-143) and is fetched when somebody reads that dict
    return h["cols"], h["off"], h["name"], h["failed"], out_rl[:dr.n_names], hp["pairs"], out_rows[:n_rows], st


def pair_table(dr, off: torch.Tensor, rows: torch.Tensor, n_reads: int, n_rows: int, cutoff=100, min_mapq=20, gap_=100,
               gap_mapq=10) -> torch.Tensor:
    """coral_bp_pair_table (K4) on device arrays in coral_sa_table's layout (off int32[n_reads + 1], rows int32[n_rows, 8]):
    int32 [2 * n_rows, 8] on the device, launched on the records' stream."""
    dev = dr.device
    assert off.dtype == torch.int32 and rows.dtype == torch.int32 and rows.is_contiguous() and off.is_contiguous()
    assert off.numel() >= n_reads + 1 and rows.numel() >= 8 * n_rows
    pairs = torch.empty((2 * max(n_rows, 1), 8), dtype=torch.int32, device=dev)
    chr_rank = torch.from_numpy(dr.chr_rank).to(dev)
    _lib.check(_lib.lib().coral_bp_pair_table(n_reads, n_rows, off.data_ptr(), rows.data_ptr(), chr_rank.data_ptr(), len(dr.chr_rank),
                                              cutoff, min_mapq, gap_, gap_mapq, pairs.data_ptr(), dr.stream()), "coral_bp_pair_table")
    return pairs


def sa_table(dr):
    """(columns int64 [8, n_rows] = qs, qe, tid, ra, rb, strand, mapq, nm; row offsets per read; name id per read; failed flag
    per read; read length per name id; pair table; the rows as they stay on the device; the staging object that owns the host
    buffers) — coral_sa_table + coral_bp_pair_table over all SA rows."""
    return _sa_table_local(dr)


def _hash_rows_local(dr, T, seg: np.ndarray, tid_has_segs: np.ndarray):
    """coral_hash_rows on the device rows of ``T``; ``seg`` int32 [4, n_seg] = contig id, start, end (exclusive), index within
    the contig — sorted by (contig, start), disjoint.  Returns (cni0, cni1, e_key, e_row) as int64 numpy arrays."""
    L = _lib.lib()
    dev = dr.device
    n_rows = T.n_rows
    rows = T.dev_rows
    assert rows is not None and rows.shape == (n_rows, 8) and rows.dtype == torch.int32 and rows.is_contiguous()
    sg = torch.from_numpy(np.ascontiguousarray(seg, dtype=np.int32)).to(dev)
    has = torch.from_numpy(np.ascontiguousarray(tid_has_segs, dtype=np.int32)).to(dev)
    c0 = torch.empty(max(n_rows, 1), dtype=torch.int32, device=dev)
    c1 = torch.empty_like(c0)
    e_key = torch.empty(2 * max(n_rows, 1), dtype=torch.int64, device=dev)
    e_row = torch.empty(2 * max(n_rows, 1), dtype=torch.int32, device=dev)
    ws_bytes = 48 * max(n_rows, 1) + (4 << 20)
    n_ent = C.c_int32(0)
    while True:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        rc = L.coral_hash_rows(n_rows, rows.data_ptr(), seg.shape[1], sg[0].data_ptr(), sg[1].data_ptr(), sg[2].data_ptr(),
                               sg[3].data_ptr(), has.data_ptr(), len(tid_has_segs), ws.data_ptr(), ws_bytes, c0.data_ptr(),
                               c1.data_ptr(), e_key.data_ptr(), e_row.data_ptr(), C.byref(n_ent), dr.stream())
        if rc == -3:
            ws_bytes = (int(n_ent.value) + 1) << 20
            continue
        break
    if rc != 0:
        raise _lib.CoralHipError("coral_hash_rows failed (%d): %s" % (rc, L.coral_sa_last_error().decode()))
    k = int(n_ent.value)
    return (c0[:n_rows].to(torch.int64).cpu().numpy(), c1[:n_rows].to(torch.int64).cpu().numpy(), e_key[:k].cpu().numpy(),
            e_row[:k].to(torch.int64).cpu().numpy())


def hash_rows(dr, T, seg, tid_has_segs):
    return _hash_rows_local(dr, T, seg, tid_has_segs)
