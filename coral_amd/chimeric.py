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
        self.read_length = np.zeros(0, np.int64)      # per name id, -1 = no primary seen
        self.n_mapq60_plain = 0

    @property
    def n_reads(self):
        return len(self.name_id)

    def device_arrays(self, device):
        """int32 device copies of (off, qs, qe, tid, ra, rb, strand, mapq), uploaded once per table."""
        import torch
        if getattr(self, "_dev", None) is None or self._dev[0] != str(device):
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(device)
            self._dev = (str(device), [up(x) for x in (self.off, self.qs, self.qe, self.tid, self.ra, self.rb, self.strand, self.mapq)])
        return self._dev[1]

    @property
    def n_rows(self):
        return len(self.qs)


def build_chimeric_table(dr) -> ChimericTable:
    """ibg:139-174 + cp:232-269 for all reads: coral_sa_table (K3) does the SA-row work on the GPU; this wrapper adds the
    float NM rate (cp:268) and the NM statistics of the non-chimeric MAPQ-60 records (ibg:153-157)."""
    from . import kernels
    import ctypes as C
    from . import _lib
    T = ChimericTable()
    cnt, s0, s1 = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
    rc = _lib.lib().coral_nm_stats(dr.n_total, dr.h_tid.ctypes.data, dr.h_sa_off.ctypes.data, dr.h_mapq.ctypes.data,
                                   dr.h_nm.ctypes.data, dr.h_qlen.ctypes.data, C.byref(cnt), C.byref(s0), C.byref(s1))
    if rc == -5:                                  # CORAL_ERR_ZERODIV: a counted record without SEQ (ibg:154)
        raise ZeroDivisionError("division by zero")
    _lib.check(rc, "coral_nm_stats")
    T.n_mapq60_plain, T.nm_sum, T.nm_sum_sq = int(cnt.value), float(s0.value), float(s1.value)
    cols, off, name_id, failed, rl = kernels.sa_table(dr)
    T.read_length = rl
    T.name_id, T.failed, T.off = name_id, failed, off
    T.qs, T.qe, T.tid, T.ra, T.rb, T.strand, T.mapq = (cols[k] for k in range(7))
    n_rows = cols.shape[1]
    T.read = np.repeat(np.arange(len(name_id), dtype=np.int64), np.diff(off))
    T.nm = cols[7].astype(np.float64) / (T.qe - T.qs) if n_rows else np.zeros(0)
    T.cni0 = np.full(n_rows, -1, dtype=np.int64)
    T.cni1 = np.full(n_rows, -1, dtype=np.int64)
    return T


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


def candidates_between(T: ChimericTable, reads: np.ndarray, I1, I2, chr_rank, dr, min_bp_match_cutoff=100, min_mapq=20,
                       gap_mapq=10) -> Candidates:
    """alignment2bp (bu:70-96) for ``reads`` (indices into T, in iteration order) between intervals I1 and I2
    (each (tid, start, end)) — coral_bp_candidates, mode 1."""
    from . import kernels
    return kernels.bp_candidates(dr, T, reads, 1, [I1, I2], chr_rank, min_bp_match_cutoff, min_mapq, 100, gap_mapq)


def first_interval_overlap(T: ChimericTable, intervals: Sequence[Tuple[int, int, int]]) -> np.ndarray:
    """interval_overlap_l (bu:37-44) for every row: index of the first interval overlapping it, -1 if none."""
    io = np.full(T.n_rows, -1, dtype=np.int64)
    rows = np.arange(T.n_rows)
    for idx, (t, s, e) in enumerate(intervals):
        m = (io < 0) & rows_overlap(T, rows, t, s, e)
        io[m] = idx
    return io


def candidates_within(T: ChimericTable, intervals, chr_rank, dr, min_bp_match_cutoff=100, min_mapq=20, gap_=100,
                      gap_mapq=10) -> Candidates:
    """alignment2bp_l (bu:129-186) for every chimeric read in dict order — coral_bp_candidates, mode 0."""
    from . import kernels
    return kernels.bp_candidates(dr, T, None, 0, list(intervals), chr_rank, min_bp_match_cutoff, min_mapq, gap_, gap_mapq)
