"""In-memory stand-in for the parts of ``pysam`` the reference's graph-build path touches.

TEST INFRASTRUCTURE ONLY (golden generation inside the build container).  pysam / htslib are not
installed here, so the reference modules under /root/reference/src are imported with this module
registered as ``pysam``.  It is the builder's *reading* of pysam semantics — it
pins the reference's own logic, not pysam's:

  * ``AlignmentFile.fetch()``                whole file, mapped records (tid >= 0) in file order
  * ``AlignmentFile.fetch(c, s, e)``         htslib overlap rule  pos < e and endpos > s
  * ``count_coverage(.., quality_threshold=0, read_callback='nofilter')``
        per aligned (M/=/X) A/C/G/T base with s <= refpos < e; records without SEQ are skipped
  * ``AlignedSegment.get_blocks()``          one (start, end) per M/=/X op; D/N advance, I/S/H/P do not
  * ``get_cigar_stats()[0][-1]``             the NM tag
  * ``infer_read_length()``                  M+I+S+=+X+H length, None when 0
  * ``get_tag('SA:Z:')``                     htslib reads only the first two characters -> tag SA
"""
from __future__ import annotations

import numpy as np

_REGISTRY = {}

CONSUME_REF = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=bool)
CONSUME_QRY = np.array([1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=bool)
IS_MATCH = np.array([1, 0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=bool)
INFER_LEN = np.array([1, 1, 0, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=bool)


def register(path, host):
    """``host`` is the dict produced by ``records_to_host``."""
    _REGISTRY[path] = host


def records_to_host(rec):
    """Convert a ``coral_amd.synth.Records`` (or decoder output) into plain numpy + python lists."""
    from coral_amd import synth
    g = lambda t: t.cpu().numpy()
    names = rec.materialise_names()
    sa_off = g(rec.sa_off)
    sa = g(rec.sa)
    sa_nm = g(rec.sa_nm)
    sa_str = [None] * rec.n
    for i in np.nonzero(sa_off[1:] > sa_off[:-1])[0]:
        ents = [synth.sa_entry_string(sa[j], int(sa_nm[j]), rec.header_chroms) for j in range(sa_off[i], sa_off[i + 1])]
        sa_str[i] = ";".join(ents) + ";"
    host = dict(n=rec.n, tid=g(rec.tid), pos=g(rec.pos), end=g(rec.end), flag=g(rec.flag), mapq=g(rec.mapq),
                qlen=g(rec.qlen), has_seq=g(rec.has_seq), nm=g(rec.nm), name_id=g(rec.name_id),
                n_cigar=g(rec.n_cigar), cigar_off=g(rec.cigar_off), cigar=g(rec.cigar).view(np.uint32),
                names=names, sa_str=sa_str, chroms=list(rec.header_chroms),
                nonacgt_rec=g(rec.nonacgt_rec), nonacgt_pos=g(rec.nonacgt_pos))
    return host


class AlignedSegment:
    __slots__ = ("_h", "_i")

    def __init__(self, host, i):
        self._h = host
        self._i = i

    @property
    def query_name(self):
        return self._h["names"][self._h["name_id"][self._i]]

    @property
    def flag(self):
        return int(self._h["flag"][self._i])

    @property
    def query_length(self):
        return int(self._h["qlen"][self._i]) if self._h["has_seq"][self._i] else 0

    @property
    def mapping_quality(self):
        return int(self._h["mapq"][self._i])

    mapq = mapping_quality

    @property
    def reference_name(self):
        return self._h["chroms"][self._h["tid"][self._i]]

    @property
    def reference_start(self):
        return int(self._h["pos"][self._i])

    @property
    def reference_end(self):
        return int(self._h["end"][self._i]) if self._h["n_cigar"][self._i] > 0 else None

    def _ops(self):
        h, i = self._h, self._i
        c = h["cigar"][h["cigar_off"][i]: h["cigar_off"][i] + h["n_cigar"][i]]
        return c & 15, c >> 4

    def get_tag(self, tag):
        if tag[:2] == "SA" and self._h["sa_str"][self._i] is not None:
            return self._h["sa_str"][self._i]
        raise KeyError("tag '%s' not present" % tag)

    def get_cigar_stats(self):
        op, ln = self._ops()
        base = [int(ln[op == k].sum()) for k in range(10)] + [int(self._h["nm"][self._i])]
        cnt = [int((op == k).sum()) for k in range(10)] + [0]
        return base, cnt

    def get_blocks(self):
        op, ln = self._ops()
        adv = np.where(CONSUME_REF[op], ln, 0).astype(np.int64)
        start = int(self._h["pos"][self._i]) + np.cumsum(adv) - adv
        m = IS_MATCH[op]
        return [(int(s), int(s + l)) for s, l in zip(start[m], ln[m])]

    def infer_read_length(self):
        op, ln = self._ops()
        tot = int(ln[INFER_LEN[op]].sum())
        return tot if tot > 0 else None


class AlignmentFile:
    def __init__(self, path, mode="rb", **kw):
        self._h = _REGISTRY[path]
        self.filename = path
        h = self._h
        self._tid_of = {c: k for k, c in enumerate(h["chroms"])}
        # records are (tid, pos) sorted: per-tid slices
        self._lo = np.searchsorted(h["tid"], np.arange(len(h["chroms"])), side="left")
        self._hi = np.searchsorted(h["tid"], np.arange(len(h["chroms"])), side="right")

    def close(self):
        pass

    def _region(self, contig, start, stop):
        h = self._h
        t = self._tid_of[contig]
        lo, hi = self._lo[t], self._hi[t]
        idx = np.arange(lo, hi)
        m = (h["pos"][lo:hi] < stop) & (h["end"][lo:hi] > start)
        return idx[m]

    def fetch(self, contig=None, start=None, stop=None, region=None, reference=None, end=None, **kw):
        h = self._h
        if contig is None and reference is not None:
            contig = reference
        if stop is None and end is not None:
            stop = end
        if contig is None:
            for i in np.nonzero(h["tid"] >= 0)[0]:
                yield AlignedSegment(h, int(i))
            return
        if start is None:
            start = 0
        if stop is None:
            stop = 1 << 40
        for i in self._region(contig, start, stop):
            yield AlignedSegment(h, int(i))

    def count_coverage(self, contig, start=None, stop=None, region=None, quality_threshold=15,
                       read_callback="all", reference=None, end=None):
        h = self._h
        total = 0
        idx = self._region(contig, start, stop)
        for i in idx:
            if read_callback == "all" and (h["flag"][i] & (0x4 | 0x100 | 0x200 | 0x400)):
                continue
            if not h["has_seq"][i] or h["n_cigar"][i] == 0:
                continue
            c = h["cigar"][h["cigar_off"][i]: h["cigar_off"][i] + h["n_cigar"][i]]
            op, ln = c & 15, (c >> 4).astype(np.int64)
            adv = np.where(CONSUME_REF[op], ln, 0)
            s = int(h["pos"][i]) + np.cumsum(adv) - adv
            m = IS_MATCH[op]
            ov = np.minimum(s[m] + ln[m], stop) - np.maximum(s[m], start)
            total += int(ov[ov > 0].sum())
        # aligned non-ACGT bases are not counted in any of the four arrays
        if len(h["nonacgt_rec"]):
            r = h["nonacgt_rec"]
            p = h["nonacgt_pos"]
            t = self._tid_of[contig]
            sel = (h["tid"][r] == t) & (p >= start) & (p < stop)
            if read_callback == "all":
                sel &= (h["flag"][r] & (0x4 | 0x100 | 0x200 | 0x400)) == 0
            total -= int(sel.sum())
        # the reference only ever sums the four arrays
        return ([total], [0], [0], [0])
