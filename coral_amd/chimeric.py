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

from typing import Dict, List, Optional, Sequence, Tuple

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


def _parse_rows(c5, m, x, c3, strand, rl):
    """Vectorised cigar2pos* (cp:17-215).  Returns qs, qe, al."""
    fwd = strand == 0
    has5, has3 = c5 > 0, c3 > 0
    ins, dele = x > 0, x < 0
    al = m + np.where(dele, -x, 0)
    both = has5 & has3
    only5 = has5 & ~has3
    only3 = ~has5 & has3
    plain = ~ins & ~dele
    qs = np.zeros_like(m)
    qe = np.zeros_like(m)
    # clip on the 5' side only: SM / SMD / SMI
    s = only5 & fwd
    qs[s] = c5[s]; qe[s] = rl[s] - 1
    s = only5 & ~fwd & plain
    qs[s] = 0; qe[s] = al[s] - 1
    s = only5 & ~fwd & dele
    qs[s] = 0; qe[s] = m[s] - 1
    s = only5 & ~fwd & ins
    qs[s] = 0; qe[s] = rl[s] - c5[s] - 1
    # clip on the 3' side only: MS / MDS / MIS
    s = only3 & ~fwd
    qs[s] = c3[s]; qe[s] = rl[s] - 1
    s = only3 & fwd & plain
    qs[s] = 0; qe[s] = al[s] - 1
    s = only3 & fwd & dele
    qs[s] = 0; qe[s] = m[s] - 1
    s = only3 & fwd & ins
    qs[s] = 0; qe[s] = rl[s] - c3[s] - 1
    # both clips: SMS / SMDS / SMIS
    s = both & plain
    qs[s] = np.where(fwd[s], c5[s], c3[s]); qe[s] = qs[s] + al[s] - 1
    s = both & ~plain & fwd
    qs[s] = c5[s]; qe[s] = rl[s] - c3[s] - 1
    s = both & ~plain & ~fwd
    qs[s] = c3[s]; qe[s] = rl[s] - c5[s] - 1
    return qs, qe, al


def build_chimeric_table(dr) -> ChimericTable:
    """ibg:139-174 + cp:232-269 on the decoded records ``dr`` (a DeviceRecords; host mirrors are used)."""
    T = ChimericTable()
    n = dr.n_total
    mapped = dr.h_tid >= 0
    nid = dr.h_name_id.astype(np.int64)
    # read_length[name] = query_length of the first record with flag < 256 (ibg:142-143)
    rl = np.full(dr.n_names, -1, dtype=np.int64)
    idx = np.nonzero(mapped & (dr.h_flag < 256))[0]
    if len(idx):
        rev = idx[::-1]
        rl[nid[rev]] = dr.h_qlen[rev]          # duplicate indices: the last write wins -> the FIRST record in file order
    T.read_length = rl
    sa_cnt = np.diff(dr.h_sa_off)
    has_sa = (sa_cnt > 0) & mapped
    # records without SA and MAPQ 60 feed nm_stats (ibg:153-157); the count must be non-zero (ibg:159)
    plain60 = mapped & ~has_sa & (dr.h_mapq == 60)
    T.n_mapq60_plain = int(plain60.sum())
    T.nm_e = dr.h_nm[plain60] / dr.h_qlen[plain60].astype(np.float64) if T.n_mapq60_plain else np.zeros(0)
    if not has_sa.any():
        return T
    # SA rows are stored record by record, so rows of records with SA are simply all rows of mapped records
    all_rec_of_row = np.repeat(np.arange(n), sa_cnt)
    keep_mapped = mapped[all_rec_of_row]
    row_idx = np.nonzero(keep_mapped)[0]
    row_name = nid[all_rec_of_row[row_idx]]
    fields = np.column_stack([row_name, dr.h_sa[row_idx].astype(np.int64), dr.h_sa_nm[row_idx].astype(np.int64)])
    # first-seen de-duplication by string equality == equality of all tokens (ibg:146-151)
    from . import _lib
    fields = np.ascontiguousarray(fields)
    is_first = np.empty(len(fields), dtype=np.uint8)
    _lib.check(_lib.lib().coral_first_seen_rows(len(fields), fields.shape[1], fields.ctypes.data, is_first.ctypes.data),
               "coral_first_seen_rows")
    fields = fields[is_first.astype(bool)]
    row_name = fields[:, 0]
    # dict insertion order: first SA-bearing record of each name
    first_row = np.full(dr.n_names, -1, dtype=np.int64)
    first_row[row_name[::-1]] = np.arange(len(row_name) - 1, -1, -1)          # first row of every name (last write wins)
    has_rows = np.nonzero(first_row >= 0)[0]
    order_names = has_rows[np.argsort(first_row[has_rows], kind="stable")]
    rank_of_name = np.empty(dr.n_names, dtype=np.int64)
    rank_of_name[order_names] = np.arange(len(order_names))
    # drop reads without a primary alignment (ibg:163-173)
    has_primary = rl[order_names] >= 0
    new_rank = np.cumsum(has_primary) - 1
    keep = has_primary[rank_of_name[row_name]]
    fields = fields[keep]
    read = new_rank[rank_of_name[fields[:, 0]]]
    read_names = order_names[has_primary]
    R = len(read_names)
    tid, pos1, strand, c5, m, x, c3, mapq, nm = (fields[:, k] for k in range(1, 10))
    shape = np.where((c5 <= 0) & (c3 <= 0), SHAPE_NO_S_OR_M, SHAPE_OK)
    shape = np.where(m <= 0, SHAPE_NO_S_OR_M, shape)
    shape = np.where(c5 == -2, SHAPE_UNKNOWN, shape)          # decoder marks unparseable CIGAR shapes with c5 = -2
    # a read fails as a whole at its first offending entry (cp:246-255)
    order0 = np.argsort(read, kind="stable")
    read, tid, pos1, strand, c5, m, x, c3, mapq, nm, shape = (a[order0] for a in (read, tid, pos1, strand, c5, m, x, c3, mapq, nm, shape))
    bad = shape != SHAPE_OK
    failed = np.zeros(R, dtype=bool)
    if bad.any():
        first_bad = np.full(R, -1, dtype=np.int64)
        bi = np.nonzero(bad)[0][::-1]
        first_bad[read[bi]] = bi
        fb = first_bad[first_bad >= 0]
        if (shape[fb] == SHAPE_UNKNOWN).any():
            raise KeyError("SA CIGAR shape outside SM/MS/SMS/SMD/MDS/SMDS/SMI/MIS/SMIS")     # cp:255
        failed[read[fb]] = True
    ok = ~failed[read]
    read, tid, pos1, strand, c5, m, x, c3, mapq, nm = (a[ok] for a in (read, tid, pos1, strand, c5, m, x, c3, mapq, nm))
    qs, qe, al = _parse_rows(c5, m, x, c3, strand, rl[read_names[read]])
    ra = np.where(strand == 0, pos1 - 1, pos1 + al - 2)
    rb = np.where(strand == 0, pos1 + al - 2, pos1 - 1)
    order = np.lexsort((np.arange(len(read)), qe, qs, read))       # stable (qs, qe) sort inside each read (cp:263)
    read, tid, strand, mapq, nm, qs, qe, ra, rb = (a[order] for a in (read, tid, strand, mapq, nm, qs, qe, ra, rb))
    if ((qe - qs) == 0).any():
        raise ZeroDivisionError("float division by zero")                            # cp:268
    T.name_id = read_names
    T.failed = failed
    T.off = np.zeros(R + 1, dtype=np.int64)
    np.cumsum(np.bincount(read, minlength=R), out=T.off[1:])
    T.read, T.qs, T.qe, T.tid, T.ra, T.rb, T.strand, T.mapq = read, qs, qe, tid, ra, rb, strand, mapq
    T.nm = nm.astype(np.float64) / (qe - qs)
    T.cni0 = np.full(len(read), -1, dtype=np.int64)
    T.cni1 = np.full(len(read), -1, dtype=np.int64)
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
