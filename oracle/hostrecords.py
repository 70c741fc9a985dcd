"""Host-side (numpy) view of alignment records for the CPU oracle.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Mirrors, on plain numpy arrays, the pysam
behaviours the reference relies on; these semantics are restated from knowledge of
pysam/htslib and are not executed against pysam ("parity unpinned" at that boundary).
"""
from __future__ import annotations

import numpy as np

# per-op tables indexed by BAM op code (MIDNSHP=X, then unused codes incl. the layout pad 15)
REF_ADV = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=np.int64)
IS_ALN = np.array([1, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=bool)
READLEN = np.array([1, 1, 0, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=np.int64)


def _sa_cigar(c5, m, x, c3):
    s = ("%dS" % c5) if c5 > 0 else ""
    s += "%dM" % m
    if x > 0:
        s += "%dI" % x
    elif x < 0:
        s += "%dD" % (-x)
    if c3 > 0:
        s += "%dS" % c3
    return s


class HostRecords:
    """Records in BAM file order.  Built from any object exposing the SoA tensors of
    ``coral_amd.synth.Records`` (the synthetic generator or the BAM decoder)."""

    def __init__(self, rec):
        g = lambda t: t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)
        self.n = int(rec.n)
        self.chroms = list(rec.header_chroms)
        self.tid = g(rec.tid).astype(np.int64)
        self.pos = g(rec.pos).astype(np.int64)
        self.end = g(rec.end).astype(np.int64)
        self.flag = g(rec.flag).astype(np.int64)
        self.mapq = g(rec.mapq).astype(np.int64)
        self.qlen = np.where(g(rec.has_seq) != 0, g(rec.qlen), 0).astype(np.int64)   # pysam query_length
        self.has_seq = g(rec.has_seq).astype(bool)
        self.nm = g(rec.nm).astype(np.int64)
        self.name_id = g(rec.name_id).astype(np.int64)
        self.n_cigar = g(rec.n_cigar).astype(np.int64)
        self.cigar_off = g(rec.cigar_off).astype(np.int64)
        self.cigar = g(rec.cigar).view(np.uint32)
        self.names = list(rec.materialise_names())
        sa_off = g(rec.sa_off)
        sa = g(rec.sa)
        sa_nm = g(rec.sa_nm)
        self.sa_str = [None] * self.n
        for i in np.nonzero(sa_off[1:] > sa_off[:-1])[0]:
            ents = []
            for j in range(sa_off[i], sa_off[i + 1]):
                tid, pos1, st, c5, m, x, c3, mq = (int(v) for v in sa[j])
                ents.append("%s,%d,%s,%s,%d,%d" % (self.chroms[tid], pos1, "+-"[st], _sa_cigar(c5, m, x, c3), mq,
                                                   int(sa_nm[j])))
            self.sa_str[i] = ";".join(ents) + ";"
        self.nonacgt_rec = g(rec.nonacgt_rec).astype(np.int64)
        self.nonacgt_pos = g(rec.nonacgt_pos).astype(np.int64)
        self.tid_of = {c: k for k, c in enumerate(self.chroms)}
        t = np.arange(len(self.chroms))
        mapped = self.tid[:int(np.count_nonzero(self.tid >= 0))]     # unplaced reads (tid -1) end a coordinate-sorted file
        assert not (mapped < 0).any() and not (np.diff(mapped) < 0).any(), "records must be sorted by contig"
        self._lo = np.searchsorted(mapped, t, side="left")
        self._hi = np.searchsorted(mapped, t, side="right")

    # -- pysam-like primitives ------------------------------------------------------------
    def ops(self, i):
        c = self.cigar[self.cigar_off[i]: self.cigar_off[i] + self.n_cigar[i]]
        return (c & 15).astype(np.int64), (c >> 4).astype(np.int64)

    def region(self, chrom, start, stop):
        """Indices (file order) of records with pos < stop and endpos > start (htslib overlap rule)."""
        t = self.tid_of[chrom]
        lo, hi = self._lo[t], self._hi[t]
        m = (self.pos[lo:hi] < stop) & (self.end[lo:hi] > start)
        return lo + np.nonzero(m)[0]

    def blocks(self, i):
        """pysam get_blocks(): one (start, end) per M/=/X op."""
        op, ln = self.ops(i)
        adv = REF_ADV[op] * ln
        st = self.pos[i] + np.cumsum(adv) - adv
        m = IS_ALN[op]
        return list(zip(st[m].tolist(), (st[m] + ln[m]).tolist()))

    def infer_read_length(self, i):
        op, ln = self.ops(i)
        tot = int((READLEN[op] * ln).sum())
        return tot if tot > 0 else None

    def count_coverage_sum(self, chrom, start, stop):
        """Σ of the four count_coverage arrays with quality_threshold=0, read_callback='nofilter'."""
        total = 0
        for i in self.region(chrom, start, stop):
            if not self.has_seq[i] or self.n_cigar[i] == 0:
                continue
            op, ln = self.ops(i)
            adv = REF_ADV[op] * ln
            st = self.pos[i] + np.cumsum(adv) - adv
            m = IS_ALN[op]
            ov = np.minimum(st[m] + ln[m], stop) - np.maximum(st[m], start)
            total += int(ov[ov > 0].sum())
        if len(self.nonacgt_rec):
            t = self.tid_of[chrom]
            sel = (self.tid[self.nonacgt_rec] == t) & (self.nonacgt_pos >= start) & (self.nonacgt_pos < stop)
            total -= int(sel.sum())
        return total
