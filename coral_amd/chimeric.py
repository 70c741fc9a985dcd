"""Chimeric (split) alignments of all reads as one structure of arrays.

Host-side mirror of the reference's ``fetch`` + ``alignment_from_satags`` + ``hash_alignment_to_seg``
(/root/reference/src/infer_breakpoint_graph.py:139-210, /root/reference/src/cigar_parsing.py:17-269) and of the
read -> breakpoint-candidate functions ``alignment2bp`` / ``alignment2bp_l`` / ``interval2bp``
(/root/reference/src/breakpoint_utilities.py:70-96, :129-186, :289-295), vectorised over every SA row instead
of looping read by read.  Row order inside a read is the reference's (qs, qe)-sorted order, reads are kept in
the insertion order of the reference's ``chimeric_alignments`` dict, so every ``(name, i, j)`` tuple and every
first-seen ordering downstream is identical.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import numpy as np

SHAPE_OK, SHAPE_NO_S_OR_M, SHAPE_UNKNOWN = 0, 1, 2


class ChimericTable:
    """SoA over the parsed SA rows of every chimeric read that has a primary alignment.

    reads (index = position in the reference's dict):  name_id[R], failed[R] (the ``([], [], [])`` case),
                                                        off[R+1] row offsets
    rows (sorted by read, then (qs, qe)):               qs, qe, tid, ra, rb (= rint[1], rint[2]; ra > rb on '-'),
                                                        strand (0 '+', 1 '-'), mapq, nm (float), cni0, cni1
    """

    def __init__(self):
        self.name_id = np.zeros(0, np.int64)
        self.failed = np.zeros(0, bool)
        self.off = np.zeros(1, np.int64)
        for k in ("qs", "qe", "tid", "ra", "rb", "strand", "mapq", "cni0", "cni1", "read"):
            setattr(self, k, np.zeros(0, np.int64))
        self.nm = np.zeros(0, np.float64)
        self._read_length = np.zeros(0, np.int64)     # per name id, -1 = no primary seen (a device tensor until somebody asks)
        self.pairs = np.zeros((0, 8), np.int32)       # coral_bp_pair_table: two slots per row (see csrc/coral_kernels.hip, K4)
        self.dev_rows = None                          # coral_sa_table's rows as they stay in HBM (int32 [n_rows, 8])
        self.staging = None                           # owner of the pinned host buffers the arrays above are views of
        self.n_mapq60_plain = 0

    @property
    def read_length(self):
        """Query length of the first primary record per name id (-1: none), as a host int64 array — fetched from the device the
        first time it is asked for (the graph build itself never needs it)."""
        rl = self._read_length
        if not isinstance(rl, np.ndarray):
            rl = self._read_length = rl.to("cpu").numpy().astype(np.int64)
        return rl

    @read_length.setter
    def read_length(self, v):
        self._read_length = v

    @property
    def n_reads(self):
        return len(self.name_id)

    @property
    def n_rows(self):
        return len(self.qs)


def _nothing_to_finish():
    pass


def build_chimeric_table(dr, defer_nm_stats: bool = False) -> ChimericTable:
    """ibg:139-174 + cp:232-269 for all reads: coral_sa_table (K3) does the SA-row work on the GPU; this wrapper adds the
    float NM rate (cp:268) and the NM statistics of the non-chimeric MAPQ-60 records (ibg:153-157)."""
    from . import kernels
    import ctypes as C
    from . import _lib
    T = ChimericTable()
    cnt, s0, s1 = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
    box = {}

    def nm_stats():
        # one pass over all records, sums in the reference's order (a chain of dependent float adds: ~2 ms at 2 M records) — on a
        # thread of its own (ctypes releases the GIL), next to the table kernels' round trips and read_cns
        box["rc"] = _lib.lib().coral_nm_stats(dr.n_total, dr.h_tid.ctypes.data, dr.h_sa_off.ctypes.data, dr.h_mapq.ctypes.data,
                                              dr.h_nm.ctypes.data, dr.h_qlen.ctypes.data, C.byref(cnt), C.byref(s0), C.byref(s1))

    def finish():
        T.finish_nm_stats = _nothing_to_finish          # (and the table no longer refers to this closure: no reference cycle)
        th = box.pop("thread", None)
        if th is not None:
            th.join()
            rc = box["rc"]
            if rc == -5:                                  # CORAL_ERR_ZERODIV: a counted record without SEQ (ibg:154)
                raise ZeroDivisionError("division by zero")
            _lib.check(rc, "coral_nm_stats")
            T.n_mapq60_plain, T.nm_sum, T.nm_sum_sq = int(cnt.value), float(s0.value), float(s1.value)
    if dr.n_total >= 200000:
        import threading
        box["thread"] = threading.Thread(target=nm_stats, name="coral-nm-stats")
        box["thread"].start()
    else:
        nm_stats()
        box["thread"] = type("_Done", (), {"join": staticmethod(lambda: None)})()
    T.finish_nm_stats = finish
    try:
        _fill_from_sa_table(T, dr)
    except Exception:
        finish()            # the reference meets a record it cannot divide by (ibg:154) before it parses any SA tag (cp:255): that error first
        raise
    if not defer_nm_stats:
        finish()
    return T


def _fill_from_sa_table(T, dr):
    from . import kernels
    cols, off, name_id, failed, rl, T.pairs, T.dev_rows, T.staging = kernels.sa_table(dr)
    T.read_length = rl
    T.name_id, T.failed, T.off = name_id, failed, off
    T.qs, T.qe, T.tid, T.ra, T.rb, T.strand, T.mapq = (cols[k] for k in range(7))
    n_rows = cols.shape[1]
    T.read = np.repeat(np.arange(len(name_id), dtype=np.int64), np.diff(off))
    T.nm = cols[7].astype(np.float64) / (T.qe - T.qs) if n_rows else np.zeros(0)
    T.cni0 = np.full(n_rows, -1, dtype=np.int64)
    T.cni1 = np.full(n_rows, -1, dtype=np.int64)


# ----------------------------------------------------------------------------------------------
# interval tests and breakpoint candidates on rows
# ----------------------------------------------------------------------------------------------
def rows_overlap(T: ChimericTable, rows, itid: int, istart: int, iend: int):
    """interval_overlap(rint, [chr, istart, iend]) (bu:11-15).  For '-' rows ra > rb, so this is true only when
    the interval contains the whole segment."""
    return (T.tid[rows] == itid) & (T.ra[rows] <= iend) & (istart <= T.rb[rows])


class Candidates:
    """Breakpoint candidates (11 fields of bu:81 / ibg:772) as arrays; chromosomes as BAM tids."""
    FIELDS = ("c1", "p1", "o1", "c2", "p2", "o2", "read", "i", "j", "gap", "swapped", "mqa", "mqb")

    def __init__(self, **kw):
        for k in self.FIELDS:
            setattr(self, k, np.asarray(kw.get(k, np.zeros(0, np.int64)), dtype=np.int64))

    def __len__(self):
        return len(self.c1)

    def take(self, idx) -> "Candidates":
        return Candidates(**{k: getattr(self, k)[idx] for k in self.FIELDS})

    @staticmethod
    def concat(parts: Sequence["Candidates"]) -> "Candidates":
        parts = [p for p in parts if len(p)]
        if not parts:
            return Candidates()
        return Candidates(**{k: np.concatenate([getattr(p, k) for p in parts]) for k in Candidates.FIELDS})


def first_interval_overlap(T: ChimericTable, intervals: Sequence[Tuple[int, int, int]]) -> np.ndarray:
    """interval_overlap_l (bu:37-44) for every row: index of the first interval overlapping it, -1 if none."""
    io = np.full(T.n_rows, -1, dtype=np.int64)
    rows = np.arange(T.n_rows)
    for idx, (t, s, e) in enumerate(intervals):
        m = (io < 0) & rows_overlap(T, rows, t, s, e)
        io[m] = idx
    return io


class PairSearch:
    """Native side of the interval search over one chimeric table (csrc/coral_search.cpp): the reach sets of ibg:369-384
    replayed as CPython sets, the runs of neighbouring segments, and alignment2bp / alignment2bp_l as a FILTER over the pair
    table the GPU built once (coral_bp_pair_table).  Every method returns ``Candidates`` with ``read`` = name id."""

    def __init__(self, T: ChimericTable, read_hash: np.ndarray, e_key: np.ndarray, e_row: np.ndarray, seg_off, seg_start, seg_end):
        import ctypes as C
        from . import _lib
        self._C, self._lib, self._L = C, _lib, _lib.lib()
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        if T.staging is not None:
            T.staging.wait("pairs")                   # the pair table's device -> host copy ran beside the CIGAR scan
        # the handle borrows these arrays: keep them alive (and unchanged in place) for its lifetime
        self._keep = [T, i64(T.off), i64(T.read), i64(T.tid), i64(T.ra), i64(T.rb), T.cni0, T.cni1, i64(read_hash), i64(T.name_id),
                      i64(e_key), i64(e_row), np.ascontiguousarray(T.pairs, dtype=np.int32), i64(seg_off), i64(seg_start), i64(seg_end)]
        assert T.cni0.dtype == np.int64 and T.cni1.dtype == np.int64 and T.cni0.flags.c_contiguous and T.cni1.flags.c_contiguous
        k = self._keep[1:]
        assert k[11].shape == (2 * T.n_rows, 8), "pair table must hold two slots per table row"
        ptr = lambda a: a.ctypes.data
        self._h = self._L.coral_search_create(T.n_reads, T.n_rows, ptr(k[0]), ptr(k[1]), ptr(k[2]), ptr(k[3]), ptr(k[4]), ptr(k[5]),
                                              ptr(k[6]), ptr(k[7]), ptr(k[8]), len(k[9]), ptr(k[9]), ptr(k[10]), ptr(k[11]),
                                              len(k[12]) - 1, ptr(k[12]), ptr(k[13]), ptr(k[14]))
        if not self._h:
            raise _lib.CoralHipError("coral_search_create failed")

    def close(self):
        if self._h:
            self._L.coral_search_free(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc, what):
        if rc == -4:                                      # CORAL_ERR_FORMAT: contig outside chr1..22,X,Y,M reaches interval2bp
            raise KeyError("contig name outside chr1..22,X,Y,M")      # gn:13-18 lookup at bu:293
        if rc != 0:
            msg = self._L.coral_search_error(self._h).decode()
            if "segment index out of range" in msg:
                raise IndexError("list index out of range")             # by[c][cni] in the reference
            raise self._lib.CoralHipError("%s failed (%d): %s" % (what, rc, msg))

    def _result(self, want_orders=False):
        """(groups int64 [G, 4], [Candidates per run], [read order per run] | None, [(cluster sizes, calls) per run]) of the last
        native call; calls = [(head, p1, p2, support index array, stats list)] as bpcluster.call_breakpoints returns them."""
        C = self._C
        nm, nc, ns = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        mp, cp, sp, oo = (C.POINTER(C.c_int64)() for _ in range(4))
        stp, op = C.POINTER(C.c_double)(), C.POINTER(C.c_int32)()
        self._lib.check(self._L.coral_search_result(self._h, C.byref(nm), C.byref(mp), C.byref(nc), C.byref(cp), C.byref(ns),
                                                    C.byref(sp), C.byref(stp), C.byref(oo), C.byref(op)), "coral_search_result")
        meta = mp[:nm.value]
        ng = meta[0]
        rows = np.ctypeslib.as_array(cp, shape=(nc.value, 13)).copy() if nc.value else np.zeros((0, 13), dtype=np.int64)
        sup = np.ctypeslib.as_array(sp, shape=(ns.value,)).copy() if ns.value else np.zeros(0, dtype=np.int64)
        groups, cands, called = np.zeros((ng, 4), dtype=np.int64), [], []
        at, m, n_calls_seen = 0, 1, 0
        for g in range(ng):
            t, b0, b1, n, ncl, ncall = meta[m:m + 6]
            groups[g] = (t, b0, b1, n)
            part = rows[at:at + n]
            at += n
            cands.append(Candidates(**{k: part[:, j] for j, k in enumerate(Candidates.FIELDS)}))
            m += 6
            sizes = meta[m:m + ncl]
            m += ncl
            calls = []
            if ncall:
                stats = stp[6 * n_calls_seen:6 * (n_calls_seen + ncall)]
                for k in range(ncall):
                    head, p1, p2, flags, s0, s1 = meta[m:m + 6]
                    m += 6
                    st = stats[6 * k:6 * k + 6]
                    if flags & 1:
                        st[2] = 0                             # the reference's ValueError branch stores the integer 0
                    if flags & 2:
                        st[3] = 0
                    calls.append((head, p1, p2, sup[s0:s1], st))
                n_calls_seen += ncall
            called.append((sizes, calls))
        orders = [] if want_orders else None
        if want_orders and ng:
            off = oo[:ng + 1]
            flat = np.ctypeslib.as_array(op, shape=(off[-1],)).copy() if off[-1] else np.zeros(0, dtype=np.int32)
            orders = [flat[off[g]:off[g + 1]].astype(np.int64) for g in range(ng)]
        return groups, cands, orders, called

    def set_params(self, min_cluster_cutoff, max_seq_len, bp_distance_cutoff, match_cutoff, accept_floor, n_threads):
        """Parameters of the build (ibg:385-391, :436-457) and the number of look-ahead threads; once, before the first step."""
        self._lib.check(self._L.coral_search_params(self._h, float(min_cluster_cutoff), int(max_seq_len), int(bp_distance_cutoff),
                                                    int(match_cutoff), float(accept_floor), int(n_threads)), "coral_search_params")

    def prefetch(self, tid, s, e, si, ei):
        """Have the step of interval (tid, s, e) on segments si..ei computed ahead on a worker thread (pure function)."""
        self._lib.check(self._L.coral_search_prefetch(self._h, int(tid), int(s), int(e), int(si), int(ei)), "coral_search_prefetch")

    def step(self, tid, s, e, si, ei, want_orders=False):
        """One step of the interval search for interval (tid, s, e) lying on segments si..ei: (groups int64 [G, 4] =
        contig id, first segment, last segment, candidates; [Candidates per group]; [read order per group] or None;
        [coral_call_breakpoints result per group])."""
        self._check(self._L.coral_search_step(self._h, int(tid), int(s), int(e), int(si), int(ei)), "coral_search_step")
        return self._result(want_orders)

    def bfs(self, seeds, seg_cn, seg_ix, chr_rank, tid_has_rows, cn_gain, interval_delta, log_level):
        """The whole interval search in one native call (coral_search_bfs).  ``seeds`` int64 [n, 4] = contig id, start, end, ccid.
        Returns {which: numpy array} of coral_search_bfs_get (copies) — see include/coral_hip.h."""
        C = self._C
        i64, f64 = (lambda a: np.ascontiguousarray(a, dtype=np.int64)), (lambda a: np.ascontiguousarray(a, dtype=np.float64))
        seeds, seg_cn, seg_ix = i64(seeds).reshape(-1, 4), f64(seg_cn), i64(seg_ix)
        chr_rank, has = np.ascontiguousarray(chr_rank, dtype=np.int32), np.ascontiguousarray(tid_has_rows, dtype=np.uint8)
        if len(seg_cn) == 0:
            seg_cn, seg_ix = np.zeros(1), np.zeros(1, dtype=np.int64)
        rc = self._L.coral_search_bfs(self._h, len(seeds), seeds.ctypes.data, seg_cn.ctypes.data, seg_ix.ctypes.data, chr_rank.ctypes.data,
                                      has.ctypes.data, float(cn_gain), int(interval_delta), int(log_level))
        if rc in (-10, -11):
            raise KeyError(self._L.coral_search_error(self._h).decode())
        if rc == -12:
            raise IndexError("list index out of range")
        self._check(rc, "coral_search_bfs")
        out = {}
        for which in range(12):
            ptr, n = C.c_void_p(), C.c_int64(0)
            self._lib.check(self._L.coral_search_bfs_get(self._h, which, C.byref(ptr), C.byref(n)), "coral_search_bfs_get")
            ty = C.c_double if which == 3 else C.c_int64
            out[which] = (np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ty)), shape=(n.value,)).copy() if n.value else
                          np.zeros(0, dtype=np.float64 if which == 3 else np.int64))
        return out

    def within(self, intervals) -> Candidates:
        """alignment2bp_l (bu:129-186) of every chimeric read, dict order, against [(tid, start, end)]."""
        iv = np.ascontiguousarray(np.asarray(intervals, dtype=np.int64).reshape(-1, 3).T)
        self._check(self._L.coral_search_within(self._h, iv.shape[1], iv[0].ctypes.data, iv[1].ctypes.data, iv[2].ctypes.data),
                    "coral_search_within")
        return self._result()[1][0]

    def between(self, reads, I1, I2) -> Candidates:
        """alignment2bp (bu:70-96) of ``reads`` (table indices, iteration order) between intervals I1 and I2 = (tid, start, end)."""
        r = np.ascontiguousarray(reads, dtype=np.int32)
        self._check(self._L.coral_search_between(self._h, len(r), r.ctypes.data, *[int(v) for v in I1], *[int(v) for v in I2]),
                    "coral_search_between")
        return self._result()[1][0]
