"""MI355X-native breakpoint-graph construction behind the reference's operator interface.

Drop-in for ``infer_breakpoint_graph.reconstruct_graph(args)``
(/root/reference/src/infer_breakpoint_graph.py:1333-1395): same arguments, same side effects (log file,
``{prefix}_amplicon{N}_graph.txt`` / ``_breakpoints.txt``), same attribute surface on the returned
``bam_to_breakpoint_nanopore`` object.

What runs where
  * BAM is decoded ONCE into structure-of-arrays records resident in HBM (coral_amd.records / coral_amd.bam);
  * every per-record loop of the reference is a HIP kernel behind the C ABI (include/coral_hip.h):
      coral_cigar_scan        <- get_blocks() walk of find_smalldel_breakpoints (ibg:750-762) + per-record sums
      coral_segment_coverage  <- count_coverage / fetch counting of read_cns (ibg:130-133) and assign_cov (ibg:1031-1034)
      coral_point_cover       <- the four point fetches per concordant edge (ibg:1043-1046)
  * the per-read SA parsing and read->candidate steps are vectorised over all SA rows (coral_amd.chimeric);
  * the small, order-sensitive steps (interval BFS, first-fit clustering, graph assembly) stay on the host with
    the reference's container types, because the reference's own output order depends on them (set-of-str
    iteration, dict insertion order — SURVEY.md Appendix A Q21).

There is no CPU fallback: importing works without a GPU, running needs libcoral_hip.so and an MI355X.
"""
from __future__ import annotations

import logging
import math
import sys
import time
from operator import itemgetter
from typing import Dict, Optional

import numpy as np

import os
import weakref

from . import _lib, global_names, hostpools, kernels
from . import _pyobjects          # CPython extension built by __graft_entry__.build(); no Python fallback
from .bpcluster import call_breakpoints
from .breakpoint_graph import BreakpointGraph, compute_cn_lr, output_breakpoint_graph_lr, output_breakpoint_info_lr
from .chimeric import Candidates, ChimericTable, PairSearch, build_chimeric_table
from .global_names import chr_idx
from .lazysets import ReadNameSet, ReadSupportSet

_ORI = "+-"
_VERIFY_SET_ORDER = os.environ.get("CORAL_VERIFY_SET_ORDER") == "1"     # tests: cross-check the set replay against real sets


def _t():
    return "#TIME " + '%.4f\t' % (time.time() - global_names.TSTART)


def interval_overlap(a, b):
    return a[0] == b[0] and int(a[1]) <= int(b[2]) and int(b[1]) <= int(a[2])


def interval_adjacent(a, b):
    if a[0] != b[0]:
        return False
    return (b[1] == a[2] + 1) if a[1] <= b[1] else (a[1] == b[2] + 1)


def interval_overlap_l(a, lst):
    for k, b in enumerate(lst):
        if interval_overlap(a, b):
            return k
    return -1


def interval_exclusive(a, lst):
    """Parts of ``a`` not covered by the intervals of ``lst`` and the indices it overlaps (bu:54-67)."""
    hit = set()
    parts = [list(a)]
    for k, b in enumerate(lst):
        for j in range(len(parts) - 1, -1, -1):
            p = parts[j]
            if interval_overlap(p, b):
                hit.add(k)
                del parts[j]
                if p[1] < b[1]:
                    parts.append([p[0], p[1], b[1] - 1, -1])
                if p[2] > b[2]:
                    parts.append([p[0], b[2] + 1, p[2], -1])
    return hit, parts


def rows_by_interval(r_tid, r_pos, r_end, intervals) -> np.ndarray:
    """Indices of the records (contig, pos, end) overlapping each interval (tid, start, end inclusive) by the htslib rule
    ``pos < e + 1 and end > s`` — interval by interval in list order, records in their own order within an interval, a record
    in two intervals counting twice: the order in which ibg:750-766 meets them (Appendix A Q11)."""
    g_tid, g_pos, g_end = (np.asarray(a).astype(np.int64) for a in (r_tid, r_pos, r_end))
    ivs = np.array(intervals, dtype=np.int64).reshape(-1, 3)
    if len(ivs) == 0 or len(g_tid) == 0:
        return np.zeros(0, dtype=np.int64)
    o = np.lexsort((ivs[:, 1], ivs[:, 0]))
    srt = ivs[o]
    disjoint = len(srt) < 2 or bool(((srt[1:, 0] != srt[:-1, 0]) | (srt[1:, 1] > srt[:-1, 2])).all())
    if disjoint and bool((srt[:, 2] >= srt[:, 1]).all()) and int(srt[:, 2].max()) < (1 << 40) and int(srt[:, 1].min()) >= 0 and \
            int(g_end.max()) < (1 << 40) and int(g_pos.min()) >= 0:
        # the intervals of a contig are disjoint and sorted, so the ones a record overlaps are consecutive: two binary searches
        # per record instead of a pass over all records per interval
        key_lo = np.searchsorted((srt[:, 0] << 41) + srt[:, 2] + 1, (g_tid << 41) + g_pos, side="right")      # first with e + 1 > pos
        key_hi = np.searchsorted((srt[:, 0] << 41) + srt[:, 1], (g_tid << 41) + g_end, side="left")           # first with s >= end
        n_hit = np.maximum(key_hi - key_lo, 0)
        rows = np.repeat(np.arange(len(g_tid)), n_hit)
        within = np.arange(len(rows)) - np.repeat(np.cumsum(n_hit) - n_hit, n_hit)
        ai = o[np.repeat(key_lo, n_hit) + within]
        return rows[np.lexsort((rows, ai))]
    parts = []
    for t, s_, e_ in ivs.tolist():
        parts.append(np.nonzero((g_tid == t) & (g_pos < e_ + 1) & (g_end > s_))[0])
    return np.concatenate(parts)


class _ChimericAlignments(dict):
    """``name -> (qint, rint(+cniset), qual, nm)`` exactly as the reference stores it (cp:269, ibg:200-210), over the
    ChimericTable: the keys (read names, in the reference's insertion order) are created the first time anything but the
    SIZE is asked for, a value the first time that entry is read.  Every access path of ``dict`` is overridden — including
    ``__iter__``, so that ``dict(x)`` / ``x.copy()`` / ``{**x}`` take the generic route through ``keys()`` and
    ``__getitem__`` instead of copying the raw table."""

    def __init__(self, owner, name_ids, keep=None):
        super().__init__()
        self._owner = weakref.proxy(owner)      # no reference cycle: the result is freed by reference counting, not by the GC
        self._name_ids = name_ids
        self._keep = keep                       # owner of the memory `name_ids` is a view of
        self._names_ = None                     # read names in dict order
        self._index_ = None
        self._filled = False
        self._n = len(name_ids)
        self._made = []                         # keys holding a materialised value

    # -- lazy parts ---------------------------------------------------------------------------------
    @property
    def _names(self):
        if self._names_ is None:
            self._names_ = self._owner._names_of(self._name_ids)
        return self._names_

    def _fill(self):
        if not self._filled:
            self._filled = True
            dict.update(self, dict.fromkeys(self._names))

    @property
    def _index(self):
        if self._index_ is None:
            self._index_ = {nm: k for k, nm in enumerate(self._names)}
        return self._index_

    def _make(self, key):
        o = self._owner
        T = o._chim
        r = self._index[key]
        if T.failed[r]:
            return ([], [], [])
        a, b = int(T.off[r]), int(T.off[r + 1])
        chroms = o.rec.header_chroms
        qint = [[int(T.qs[k]), int(T.qe[k])] for k in range(a, b)]
        rint = []
        for k in range(a, b):
            seg = [chroms[T.tid[k]], int(T.ra[k]), int(T.rb[k]), _ORI[T.strand[k]]]
            if o._hashed:
                seg.append(o._cniset(k))
            rint.append(seg)
        qual = [int(T.mapq[k]) for k in range(a, b)]
        nm = [float(T.nm[k]) for k in range(a, b)]
        return (qint, rint, qual, nm)

    # -- dict protocol --------------------------------------------------------------------------------
    def __len__(self):
        return dict.__len__(self) if self._filled else self._n

    def __contains__(self, key):
        self._fill()
        return dict.__contains__(self, key)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def __getitem__(self, key):
        self._fill()
        v = dict.__getitem__(self, key)
        if v is None:
            v = self._make(key)
            dict.__setitem__(self, key, v)
            self._made.append(key)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __setitem__(self, key, value):
        self._fill()
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        self._fill()
        dict.__delitem__(self, key)

    def pop(self, key, *default):
        self._fill()
        if dict.__contains__(self, key):
            v = self[key]
            dict.__delitem__(self, key)
            return v
        if default:
            return default[0]
        raise KeyError(key)

    def setdefault(self, key, default=None):
        if key not in self:
            self[key] = default
        return self[key]

    def update(self, *a, **kw):
        self._fill()
        dict.update(self, *a, **kw)

    def copy(self):
        return dict(self.items())

    def __eq__(self, other):
        return dict(self.items()) == other

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        return repr(dict(self.items()))

    def __reduce__(self):
        return (dict, (self.items(),))

    def invalidate(self):
        for k in self._made:
            if dict.__contains__(self, k):
                dict.__setitem__(self, k, None)
        self._made = []


class _RecordsFile:
    """Stand-in for ``lr_bamfh`` after decoding: region fetch + close, as compute_path_constraints uses them
    (ibg:1306-1310)."""

    class _Rec:
        __slots__ = ("query_name", "mapq", "mapping_quality", "reference_name", "reference_start", "reference_end", "flag")

    def __init__(self, dr):
        self._dr = dr

    def fetch(self, contig=None, start=None, stop=None, **kw):
        dr = self._dr
        if contig is None:
            idx = np.nonzero(dr.h_tid >= 0)[0]
        else:
            tid = dr.header_chroms.index(contig)
            idx = dr.region(tid, 0 if start is None else start, (1 << 40) if stop is None else stop)
        names = dr.names.take(dr.h_name_id[idx])
        for k, i in enumerate(idx):
            r = self._Rec()
            r.query_name = names[k]
            r.mapq = r.mapping_quality = int(dr.h_mapq[i])
            r.reference_name = dr.header_chroms[dr.h_tid[i]]
            r.reference_start = int(dr.h_pos[i])
            r.reference_end = int(dr.h_end[i])
            r.flag = int(dr.h_flag[i])
            yield r

    def close(self):
        pass


class bam_to_breakpoint_nanopore():
    max_seq_len = 2000000
    cn_gain = 5.0
    min_bp_match_cutoff_ = 100
    interval_delta = 100000
    max_breakpoint_distance_cutoff = 2000
    min_del_len = 600

    def __init__(self, lr_bamfile, seedfile, records=None, device="cuda:0"):
        if records is None:
            from .bam import load_bam
            from .records import DeviceRecords
            records = DeviceRecords(load_bam(lr_bamfile, device), device)
        self.rec = records
        self.lr_bamfh = _RecordsFile(records)
        self.lr_graph = []
        self.min_bp_cov_factor = 1.0
        self.min_cluster_cutoff = 3
        self.read_length = dict()
        self.chimeric_alignments = dict()
        self.chimeric_alignments_seg = dict()
        self.large_indel_alignments = dict()
        self.nm_stats = [0.0, 0.0, 0]
        self.nm_filter = False
        self.amplicon_intervals = []
        self.amplicon_interval_connections = dict()
        self.cns_intervals = []
        self.cns_intervals_by_chr = dict()
        self.log2_cn = []
        self.cns_tree = dict()
        self.normal_cov = 0.0
        self.ccid2id = dict()
        self.new_bp_list = []
        self._search_ctx: Optional[PairSearch] = None      # native side of the interval search (coral_search_*)
        self.new_bp_stats = []
        self.new_bp_ccids = []
        self.source_edges = []
        self.source_edge_ccids = []
        self.path_constraints = dict()
        self.longest_path_constraints = dict()
        self.cycles = dict()
        self.cycle_weights = dict()
        self.path_constraints_satisfied = dict()
        self._scan = None
        self._chim: Optional[ChimericTable] = None
        self._hashed = False
        chroms = self.rec.header_chroms
        self._tid_of = {c: k for k, c in enumerate(chroms)}
        with open(seedfile, 'r') as fp:
            for line in fp:
                s = line.strip().split()
                self.amplicon_intervals.append([s[0], int(s[1]), int(s[2]), -1])
        logging.debug(_t() + "Parsed %d seed amplicon intervals." % (len(self.amplicon_intervals)))

    # ------------------------------------------------------------------------------------------
    def scan(self):
        """The fused CIGAR pass (coral_cigar_scan); run once, reused by every later step."""
        if self._scan is None:
            self._scan = kernels.cigar_scan(self.rec, self.min_del_len, 20)
        return self._scan

    def _coverage(self, segs):
        """[(chr, start, end_inclusive)] -> (n_reads, n_bases) via coral_segment_coverage."""
        tri = [(self._tid_of[c], int(s), int(e) + 1) for c, s, e in segs]
        return kernels.segment_coverage(self.rec, self.scan(), tri)

    # ---- A1 / A2 -----------------------------------------------------------------------------
    def read_cns(self, cns):
        """Parse the CN segments and estimate the normal long-read coverage (ibg:75-136)."""
        self.cns_intervals, self.log2_cn = [], []
        tree: Dict[str, list] = {}
        idx = 0
        with open(cns, 'r') as fp:
            for line in fp:
                s = line.strip().split()
                if s[0] == "chromosome":
                    continue
                self.cns_intervals.append([s[0], int(s[1]), int(s[2]) - 1])
                if s[0] not in tree:
                    tree[s[0]] = []
                    self.cns_intervals_by_chr[s[0]] = []
                    idx = 0
                tree[s[0]].append((int(s[1]), int(s[2]), idx))
                idx += 1
                if cns.endswith(".cns"):
                    self.cns_intervals_by_chr[s[0]].append([s[0], int(s[1]), int(s[2]) - 1, 2 * (2 ** float(s[4]))])
                    self.log2_cn.append(float(s[4]))
                elif cns.endswith(".bed"):
                    self.cns_intervals_by_chr[s[0]].append([s[0], int(s[1]), int(s[2]) - 1, float(s[3])])
                    self.log2_cn.append(np.log2(float(s[3]) / 2.0))
                else:
                    sys.stderr.write(cns + "\n")
                    sys.stderr.write("Invalid cn_seg file format!\n")
        self.cns_tree = {}
        for c, v in tree.items():
            st, en, ix = (np.array(col, dtype=np.int64) for col in zip(*v))
            o = np.argsort(st, kind="stable")
            disjoint = bool((en[o][:-1] <= st[o][1:]).all())
            self.cns_tree[c] = (st, en, ix, o, disjoint)
        logging.debug(_t() + "Total num LR copy number segments: %d." % (len(self.log2_cn)))
        order = np.argsort(self.log2_cn)
        im = int(len(order) / 2.4)
        ip = im + 1
        picked = [self.cns_intervals[order[ip]], self.cns_intervals[order[im]]]
        total_int_len = sum(p[2] - p[1] + 1 for p in picked)
        i = 1
        while total_int_len < 10000000:
            for p in (self.cns_intervals[order[ip + i]], self.cns_intervals[order[im - i]]):
                picked.append(p)
                total_int_len += p[2] - p[1] + 1
            i += 1
        logging.debug(_t() + "Use %d LR copy number segments." % (len(picked)))
        logging.debug(_t() + "Total length of LR copy number segments: %d." % (total_int_len))
        _, n_bases = self._coverage(picked)
        nnc = int(n_bases.sum())
        self.normal_cov = nnc * 1.0 / total_int_len
        logging.info(_t() + "LR normal cov = %f." % (self.normal_cov))
        self.min_cluster_cutoff = max(self.min_cluster_cutoff, self.min_bp_cov_factor * self.normal_cov)
        logging.debug(_t() + "Reset min_cluster_cutoff to %f." % (self.min_cluster_cutoff))

    def pos2cni(self, chr, pos):
        """Indices of the CN segments containing ``pos`` (the reference's IntervalTree point query, ibg:177-178)."""
        st, en, ix, _, _ = self.cns_tree[chr]
        return ix[(st <= pos) & (pos < en)].tolist()

    def _pos2cni_many(self, chr, pos: np.ndarray) -> np.ndarray:
        """Vectorised point query: CN-segment index or -1; asserts at most one hit like ibg:191."""
        st, en, ix, o, disjoint = self.cns_tree[chr]
        if disjoint:
            k = np.searchsorted(st[o], pos, side="right") - 1
            kk = np.clip(k, 0, len(o) - 1)
            ok = (k >= 0) & (pos < en[o][kk])
            return np.where(ok, ix[o][kk], -1)
        hits = (st[None, :] <= pos[:, None]) & (pos[:, None] < en[None, :])
        assert (hits.sum(1) <= 1).all()
        return np.where(hits.any(1), ix[np.argmax(hits, 1)], -1)

    # ---- A3 ----------------------------------------------------------------------------------
    def launch_record_kernels(self):
        """Issue the two passes that depend on the records alone before any host logic runs: the fused CIGAR scan (launched
        without waiting for it) and, on a stream of their own, the SA table + pair table (K3, K4), whose host round trips then
        overlap with the scan.  What fetch() would raise is kept and raised there."""
        trace = os.environ.get("CORAL_TRACE_OPEN") == "1"
        t0 = time.perf_counter()
        self.scan()
        t1 = time.perf_counter()
        try:
            self._chim_early = build_chimeric_table(self.rec, defer_nm_stats=True)
        except Exception as exc:                      # noqa: BLE001 — re-raised by fetch(), where the reference raises
            self._chim_early = exc
        T = self._chim_early
        if isinstance(T, ChimericTable) and len(T.name_id) >= 20000:
            # hash(read name) of every chimeric read (the set-order replay of the interval search needs them): from the name
            # bytes, on a thread of its own (no interpreter lock is held for it), beside read_cns / fetch / hash_alignment_to_seg
            import threading
            names, box = self.rec.names, {}

            def hashes(ids=T.name_id):
                box["h"] = names.hashes(ids)
            th = threading.Thread(target=hashes, name="coral-name-hashes")
            th.start()
            T._hash_job = (th, box)
        if trace:
            sys.stderr.write("launch_record_kernels: scan issue %.1f ms, chimeric table %.1f ms\n" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))

    def fetch(self):
        """Collect chimeric alignments of every read from the SA tags (ibg:139-174)."""
        T = getattr(self, "_chim_early", None)
        self._chim_early = None
        if isinstance(T, Exception):
            raise T
        if T is None:
            T = build_chimeric_table(self.rec)
        self._chim = T
        T.finish_nm_stats()                           # (started with the table kernels; raises what the reference raises at ibg:154)
        if T.n_mapq60_plain == 0:
            raise ZeroDivisionError("float division by zero")                # ibg:159
        s0, s1 = T.nm_sum, T.nm_sum_sq                                        # sequential adds, as the reference
        mean = s0 / T.n_mapq60_plain
        self.nm_stats = [mean, math.sqrt(s1 / T.n_mapq60_plain - mean ** 2), T.n_mapq60_plain]
        names = self.rec.names
        # (T's host arrays are views of pinned staging buffers leased for T's lifetime: whatever outlives this object and still
        # looks at one of them holds the staging too, or the buffer would go back to the pool under it)
        self.read_length = _LazyReadLength(names, T)
        self.chimeric_alignments = _ChimericAlignments(self, T.name_id, keep=T.staging)
        logging.info(_t() + "Fetched %d chimeric reads." % (len(self.chimeric_alignments)))
        logging.info(_t() + "Computed alignment intervals on all chimeric reads.")

    # ---- A4 ----------------------------------------------------------------------------------
    def _cniset(self, row):
        T = self._chim
        if T.cni0[row] == -3:
            return set([-1])
        s = set([int(T.cni0[row]), int(T.cni1[row])])
        if len(s) > 1 and -1 in s:
            s.remove(-1)
        return s

    def hash_alignment_to_seg(self):
        """CN-segment indices of both ends of every SA segment + the inverted index (ibg:181-210)."""
        T = self._chim
        chroms = self.rec.header_chroms
        n_tid = len(chroms)
        if all(v[4] for v in self.cns_tree.values()):
            # every chromosome's segments are disjoint (the usual case): coral_hash_rows on the SA table's device rows — two
            # binary searches per alignment and one stable radix sort for the inverted index
            parts = [(self._tid_of[c], v) for c, v in self.cns_tree.items() if c in self._tid_of]
            has_tree = np.zeros(n_tid, dtype=np.int32)
            seg = np.zeros((4, 0), dtype=np.int32)
            if parts:
                g_tid = np.concatenate([np.full(len(v[3]), t, dtype=np.int64) for t, v in parts])
                g_st = np.concatenate([v[0][v[3]] for _, v in parts])
                g_en = np.concatenate([v[1][v[3]] for _, v in parts])
                g_ix = np.concatenate([v[2][v[3]] for _, v in parts])
                o = np.lexsort((g_st, g_tid))
                seg = np.stack([g_tid[o], g_st[o], g_en[o], g_ix[o]]).astype(np.int32)
                has_tree[[t for t, _ in parts]] = 1
            c0, c1, self._e_key, self._e_row = kernels.hash_rows(self.rec, T, seg, has_tree)
            T.cni0[:] = c0
            T.cni1[:] = c1
        else:
            lo = np.minimum(T.ra, T.rb)
            hi = np.maximum(T.ra, T.rb)
            T.cni0[:] = -3                     # -3: chromosome absent from the CN file -> set([-1]) (ibg:210)
            T.cni1[:] = -3
            for t in np.unique(T.tid):
                c = chroms[t]
                if c not in self.cns_tree:
                    continue
                m = T.tid == t
                T.cni0[m] = self._pos2cni_many(c, lo[m])
                T.cni1[m] = self._pos2cni_many(c, hi[m])
            # inverted index (chr, cni) -> reads, in the reference's append order (read, then segment): up to two entries per
            # row, generated in row order, then one STABLE sort by (chromosome, segment)
            rows = np.nonzero(T.cni0 != -3)[0]
            a, b = T.cni0[rows], T.cni1[rows]
            ent_row = np.repeat(rows, 2)
            ent_cni = np.stack([a, b], axis=1).ravel()
            valid = np.stack([a >= 0, (b >= 0) & (b != a)], axis=1).ravel()
            ent_row, ent_cni = ent_row[valid], ent_cni[valid]
            key = T.tid[ent_row] * (1 << 32) + ent_cni
            o = np.argsort(key, kind="stable")
            self._e_row, self._e_key = ent_row[o], key[o]
        self._hashed = True
        self.chimeric_alignments.invalidate()
        rows = np.nonzero(T.cni0 != -3)[0]
        kt = T.tid[rows]
        present = np.nonzero(np.bincount(kt, minlength=n_tid))[0] if len(kt) else np.zeros(0, dtype=np.int64)
        first_at = {int(t): int(np.argmax(kt == t)) for t in present}
        self._seg_tids = sorted(first_at, key=first_at.get)                       # dict key order of ibg:201-202
        self.chimeric_alignments_seg = _SegIndexView(self)

    # ---- A5 ----------------------------------------------------------------------------------
    def find_amplicon_intervals(self):
        by = self.cns_intervals_by_chr
        logging.debug(_t() + "Updating seed amplicon intervals according to CN segments.")
        for iv in self.amplicon_intervals:
            c = iv[0]
            lcni = self.pos2cni(c, iv[1])[0]
            rcni = self.pos2cni(c, iv[2])[0]
            iv[1] = by[c][lcni][1]
            if self.pos2cni(c, by[c][lcni][1] - self.interval_delta):
                iv[1] = by[c][lcni][1] - self.interval_delta
            iv[2] = by[c][rcni][2]
            if self.pos2cni(c, by[c][rcni][2] + self.interval_delta):
                iv[2] = by[c][rcni][2] + self.interval_delta
        ccid = 0
        _lib.check_pyset_replay()            # once per process: the set replay must match this interpreter's sets
        if not _VERIFY_SET_ORDER and os.environ.get("CORAL_SEARCH_BFS", "native") != "python" and not self.new_bp_list and \
                not self.amplicon_interval_connections:
            self._find_intervals_native()
        else:                                # the same search step by step from Python (tests: every set union re-checked)
            if len(self._chim.read):
                for ai in range(len(self.amplicon_intervals)):
                    self._prefetch_step(ai)
            for ai in range(len(self.amplicon_intervals)):
                if self.amplicon_intervals[ai][3] == -1:
                    self.find_interval_i(ai, ccid)
                    ccid += 1
        logging.debug(_t() + "Identified %d amplicon intervals in total." % len(self.amplicon_intervals))
        self._merge_intervals()

    def _find_intervals_native(self):
        """The loop over the seeds + find_interval_i (ibg:343-673) as ONE native call (coral_search_bfs): the pure steps run
        ahead on worker threads as before, and the order-dependent half (addbp, interval refinement, interval_exclusive,
        connections) runs next to them instead of one Python round trip per interval.  What comes back is turned into the
        reference's containers here: interval lists, new_bp_list rows with lazy support sets, the connections dict in its
        insertion order, and the log lines in the order the reference emits them."""
        chroms = self.rec.header_chroms
        by = self.cns_intervals_by_chr
        S = self._search()
        n_tid = len(chroms)
        seg_cn = np.fromiter((float(v[3]) for c in chroms for v in by.get(c, ())), dtype=np.float64)
        seg_ix = np.concatenate([np.asarray(self.cns_tree[c][2], dtype=np.int64) if c in self.cns_tree else np.zeros(0, dtype=np.int64)
                                 for c in chroms]) if n_tid else np.zeros(0, dtype=np.int64)
        assert len(seg_ix) == len(seg_cn)
        has = np.zeros(max(n_tid, 1), dtype=np.uint8)
        has[list(self._seg_tids)] = 1
        seeds = [[self._tid_of[iv[0]], int(iv[1]), int(iv[2]), int(iv[3])] for iv in self.amplicon_intervals]
        log = logging.getLogger()
        level = 2 if log.isEnabledFor(logging.DEBUG) else 1 if log.isEnabledFor(logging.WARNING) else 0
        R = S.bfs(seeds, seg_cn, seg_ix, self.rec.chr_rank, has, self.cn_gain, self.interval_delta, level)
        iv = R[0].reshape(-1, 5).tolist()
        self.amplicon_intervals = [[chroms[t], bool(s) if isb else s, e, cc] for t, s, e, cc, isb in iv]
        names = self.rec.names
        bp, meta, stats = R[1].reshape(-1, 11), R[2].reshape(-1, 4).tolist(), R[3].reshape(-1, 6)
        chunks, c_read, c_i, c_j = R[4].reshape(-1, 2).tolist(), R[5], R[6], R[7]
        head_names = names.take(bp[:, 6]) if len(bp) else []
        for k, f in enumerate(bp.tolist()):
            flags, cc, ch0, ch1 = meta[k]
            a, b = chunks[ch0]
            support = ReadSupportSet(names, c_read[a:b], c_i[a:b], c_j[a:b])
            for a, b in chunks[ch0 + 1:ch1]:
                support |= ReadSupportSet(names, c_read[a:b], c_i[a:b], c_j[a:b])
            st = stats[k].tolist()
            if flags & 1:
                st[2] = 0                                 # the reference's ValueError branch stores the integer 0
            if flags & 2:
                st[3] = 0
            self.new_bp_list.append([chroms[f[0]], f[1], _ORI[f[2]], chroms[f[3]], f[4], _ORI[f[5]], (head_names[k], f[7], f[8]),
                                     f[9], f[10], support])
            self.new_bp_ccids.append(cc)
            self.new_bp_stats.append(st)
        conn = self.amplicon_interval_connections
        keys, off, vals = R[8].reshape(-1, 2).tolist(), R[9].tolist(), R[10].tolist()
        for q, (a, b) in enumerate(keys):
            s_ = conn[(a, b)] = set()
            for v in vals[off[q]:off[q + 1]]:
                s_.add(v)
        if level:
            for ty, a, b, c, d, e in R[11].reshape(-1, 6).tolist():
                if ty == 3:
                    logging.warning(_t() + "\t\tExact breakpoint outside amplicon interval.")
                elif ty == 0:
                    logging.debug(_t() + "\t\tNext amplicon interval %d: %s." % (a, [chroms[b], c, d, e]))
                elif ty == 1:
                    logging.debug(_t() + "\t\tFound %d reads connecting the two intervals." % a)
                elif ty == 2:
                    logging.debug(_t() + "New cluster of size %d." % a)
                elif ty == 4:
                    logging.debug(_t() + "\t\tAdded new interval %s to the amplicon interval list." % [chroms[a], bool(b) if d else b, c, -1])

    def _merge_intervals(self):
        """Sort, merge adjacent / overlapping intervals, remap connections, relabel components (ibg:236-319)."""
        ivs = self.amplicon_intervals
        order = sorted(range(len(ivs)), key=lambda i: (chr_idx[ivs[i][0]], ivs[i][1]))
        srt = [ivs[i] for i in order]
        runs, first = [], 0
        for k in range(len(srt) - 1):
            if not (interval_adjacent(srt[k + 1], srt[k]) or interval_overlap(srt[k], srt[k + 1])):
                if k > first:
                    runs.append((first, k))
                first = k + 1
        if srt and first < len(srt) - 1:
            runs.append((first, len(srt) - 1))
        conn = self.amplicon_interval_connections
        for a, b in reversed(runs):
            srt[a][2] = srt[b][2]
            for k in range(a + 1, b + 1):
                if srt[k][3] != srt[a][3]:
                    old = srt[k][3]
                    for iv in srt:
                        if iv[3] == old:
                            iv[3] = srt[a][3]
            cmap = {key: key for key in conn}
            for k in range(a + 1, b + 1):
                keep, gone = order[a], order[k]
                for key in cmap:
                    u, v = cmap[key]
                    if gone == u:
                        u = keep
                    cmap[key] = (u, v)
                    if gone == cmap[key][1]:
                        cmap[key] = (cmap[key][0], keep)
                    if cmap[key][1] < cmap[key][0]:
                        cmap[key] = (cmap[key][1], cmap[key][0])
            for key, new in cmap.items():
                if key != new:
                    if new not in conn:
                        conn[new] = conn[key]
                    else:
                        conn[new] |= conn[key]
                    del conn[key]
                    if new[0] == new[1]:
                        del conn[new]
            for k in range(b, a, -1):
                del srt[k]
                del order[k]
        self.amplicon_intervals = list(srt)
        where = {order[i]: i for i in range(len(order))}
        renamed = {key: (min(where[key[0]], where[key[1]]), max(where[key[0]], where[key[1]])) for key in conn}
        self.amplicon_interval_connections = {renamed[key]: conn[key] for key in conn}
        seen = [False] * len(self.amplicon_intervals)
        for ai in range(len(self.amplicon_intervals)):
            if seen[ai]:
                continue
            label = self.amplicon_intervals[ai][3]
            queue = [ai]
            while queue:
                cur = queue.pop(0)
                seen[cur] = True
                self.amplicon_intervals[cur][3] = label
                for (p, q) in self.amplicon_interval_connections:
                    if p == cur and not seen[q]:
                        queue.append(q)
                    elif q == cur and not seen[p]:
                        queue.append(p)
        logging.debug(_t() + "There are %d amplicon intervals after merging." % len(self.amplicon_intervals))

    def addbp(self, bp_, bpr_, bp_stats_, ccid):
        """Append a breakpoint, or merge its reads into the first one within 200 bp at both ends (ibg:326-340).
        ``bpr_``: the supporting (name, i, j) tuples — a ReadSupportSet from this build, or any iterable of tuples."""
        for k, bp in enumerate(self.new_bp_list):
            if bp[0] == bp_[0] and bp[3] == bp_[3] and bp[2] == bp_[2] and bp[5] == bp_[5] and \
                    abs(bp[1] - bp_[1]) < 200 and abs(bp[4] - bp_[4]) < 200:
                if isinstance(bpr_, ReadSupportSet):
                    bp[-1] |= bpr_                       # both still arrays: the chunk is appended, nothing is built
                else:
                    bp[-1] |= set(bpr_)
                return k
        self.new_bp_list.append(bp_ + [bpr_])
        self.new_bp_ccids.append(ccid)
        self.new_bp_stats.append(bp_stats_)
        return len(self.new_bp_list) - 1

    # -- candidate clusters -> breakpoints ---------------------------------------------------------
    def _names_of(self, ids) -> list:
        """Read-name strings of an array of name ids (coral_amd.names.NameTable: one C loop, no per-item interpreter work)."""
        return self.rec.names.take(ids)

    def _support(self, c: Candidates, idx) -> ReadSupportSet:
        """The ``set((name, i, j))`` of the candidates ``idx`` (bu:81 / ibg:772) — as arrays until somebody iterates it."""
        return ReadSupportSet(self.rec.names, c.read[idx], c.i[idx], c.j[idx])

    def _cluster_and_call(self, c: Candidates, advance_subcluster: bool):
        """Native part of _call_breakpoints: (cluster sizes, accepted calls) — a pure function of the candidates."""
        floor = max(self.normal_cov * self.min_bp_cov_factor, 3.0)
        return call_breakpoints(c, self.min_cluster_cutoff, self.max_breakpoint_distance_cutoff, self.min_bp_match_cutoff_, floor,
                                advance_subcluster)

    def _call_breakpoints(self, c: Candidates, advance_subcluster: bool, called=None):
        """Cluster the candidates and yield (bp list, support set, stats) for every accepted (sub)cluster.

        ``advance_subcluster`` is False inside the interval BFS, where the reference never increments its
        sub-cluster counter (ibg:442-457, Appendix A Q4), and True in find_breakpoints / find_smalldel_breakpoints.
        ``called``: the result of ``_cluster_and_call`` when it was computed ahead.
        """
        chroms = self.rec.header_chroms
        sizes, calls = called if called is not None else self._cluster_and_call(c, advance_subcluster)
        if logging.getLogger().isEnabledFor(logging.DEBUG):
            for sz in sizes:
                logging.debug(_t() + "New cluster of size %d." % sz)
        for head, p1, p2, sup, st in calls:
            bp = [chroms[c.c1[head]], p1, _ORI[c.o1[head]], chroms[c.c2[head]], p2, _ORI[c.o2[head]],
                  (self.rec.names[c.read[head]], int(c.i[head]), int(c.j[head])), int(c.gap[head]), int(c.swapped[head])]
            yield bp, self._support(c, sup), st

    # -- BFS helpers ---------------------------------------------------------------------------------
    def _read_hashes(self) -> np.ndarray:
        """hash(read name) of every chimeric read (index = position in the chimeric table): the interpreter's own str hash,
        computed from the name bytes (coral_amd.names.NameTable.hashes) — the work behind the reference's set-of-str
        construction at ibg:379-384, :412-418, without the 164 k str objects."""
        T = self._chim
        if getattr(T, "_hashes", None) is None:
            job = getattr(T, "_hash_job", None)
            if job is not None:                       # started with the record kernels
                T._hash_job = None
                job[0].join()
                T._hashes = job[1].get("h")
            if getattr(T, "_hashes", None) is None:
                T._hashes = self.rec.names.hashes(T.name_id)
        return T._hashes

    def _search(self) -> PairSearch:
        """Native side of the interval search over this build's chimeric table (created after hash_alignment_to_seg)."""
        if self._search_ctx is None:
            chroms = self.rec.header_chroms
            by = self.cns_intervals_by_chr
            seg_off = np.zeros(len(chroms) + 1, dtype=np.int64)
            for t, c in enumerate(chroms):
                seg_off[t + 1] = seg_off[t] + len(by.get(c, ()))
            seg_start = np.fromiter((v[1] for c in chroms for v in by.get(c, ())), dtype=np.int64, count=int(seg_off[-1]))
            seg_end = np.fromiter((v[2] for c in chroms for v in by.get(c, ())), dtype=np.int64, count=int(seg_off[-1]))
            self._search_ctx = PairSearch(self._chim, self._read_hashes(), self._e_key, self._e_row, seg_off, seg_start, seg_end)
            # a few thousand chimeric reads: a step costs less than handing it to another thread
            small = len(self._chim.read) < int(os.environ.get("CORAL_SEARCH_MIN_READS", "20000"))
            n_threads = 0 if (_VERIFY_SET_ORDER or small) else int(os.environ.get("CORAL_SEARCH_THREADS", "6"))
            floor = max(self.normal_cov * self.min_bp_cov_factor, 3.0)
            self._search_ctx.set_params(self.min_cluster_cutoff, self.max_seq_len, self.max_breakpoint_distance_cutoff,
                                        self.min_bp_match_cutoff_, floor, n_threads)
        return self._search_ctx

    def _prefetch_step(self, idx):
        """Tell the native search that interval ``idx`` will be searched: its step (a pure function of the coordinates) is
        computed ahead on a worker thread while this thread does the order-dependent part of earlier steps."""
        chrom, s, e = self.amplicon_intervals[idx][:3]
        try:
            si = self.pos2cni(chrom, s)[0]
            ei = self.pos2cni(chrom, e)[0]
        except Exception:
            return
        tid = self._tid_of[chrom]
        if tid in self._seg_tids:
            self._search().prefetch(tid, s, e, si, ei)

    def _search_step(self, chrom, s, e):
        """The part of one step of the interval search that is a pure function of the interval's coordinates (ibg:362-434):
        reachable segments and their read sets, the runs of neighbouring segments, the iteration order of every run's reads and
        the breakpoint candidates between each run and the interval — ONE native call (coral_search_step) that replays the
        reference's sets of read names on index arrays and filters the GPU-built pair table; then the clustering of every
        run's candidates into exact breakpoints (coral_call_breakpoints, inside the same job).  The job is usually ready: it was
        requested when the interval entered the queue (_prefetch_step).  Returns None (nothing reachable / interval off
        the CN segments) or (plan [(chr, first segment, last segment)], candidates per run, calls per run)."""
        try:
            si = self.pos2cni(chrom, s)[0]
            ei = self.pos2cni(chrom, e)[0]
        except Exception:
            return None
        tid = self._tid_of[chrom]
        if tid not in self._seg_tids:
            raise KeyError(chrom)                       # self.chimeric_alignments_seg[chr] at ibg:371
        groups, all_cands, orders, called = self._search().step(tid, s, e, si, ei, want_orders=_VERIFY_SET_ORDER)
        chroms = self.rec.header_chroms
        plan = [(chroms[int(g[0])], int(g[1]), int(g[2])) for g in groups]
        if _VERIFY_SET_ORDER:
            self._verify_step(tid, si, ei, plan, orders)
        return plan, all_cands, called

    def _verify_step(self, tid, si, ei, plan, orders):
        """Tests only (CORAL_VERIFY_SET_ORDER=1): ibg:369-419 with REAL sets of str; the native step must give the same runs
        and, for every run, the same iteration order of the united set."""
        T = self._chim
        names, chroms, by = self.chimeric_alignments._names, self.rec.header_chroms, self.cns_intervals_by_chr
        lo = np.searchsorted(self._e_key, tid * (1 << 32) + si, side="left")
        hi = np.searchsorted(self._e_key, tid * (1 << 32) + ei + 1, side="left")
        real: Dict[str, Dict[int, set]] = {}
        done = set()
        for row in self._e_row[lo:hi].tolist():
            r = int(T.read[row])
            if r in done:
                continue
            done.add(r)
            for k in range(int(T.off[r]), int(T.off[r + 1])):
                c0, c1, t = int(T.cni0[k]), int(T.cni1[k]), int(T.tid[k])
                for cni in ([c0] if c0 >= 0 else []) + ([c1] if c1 >= 0 and c1 != c0 else []):
                    if t != tid or cni <= si or cni >= ei:
                        d = real.setdefault(chroms[t], {})
                        if cni in d:
                            d[cni].add(names[r])
                        else:
                            d[cni] = set([names[r]])
        want = []
        for c in real:
            bins = sorted(j for j in real[c] if not len(real[c][j]) < self.min_cluster_cutoff)
            if not bins:
                continue
            first, acc = 0, set([])
            for k in range(len(bins) - 1):
                acc |= real[c][bins[k]]
                if bins[k + 1] - bins[k] > 2 or by[c][bins[k + 1]][1] - by[c][bins[k]][2] > self.max_seq_len:
                    want.append(((c, bins[first], bins[k]), list(acc)))
                    first, acc = k + 1, set([])
            acc |= real[c][bins[-1]]
            want.append(((c, bins[first], bins[-1]), list(acc)))
        assert [w[0] for w in want] == plan, "runs of reachable segments diverged from the reference's"
        index = self.chimeric_alignments._index
        for (_, members), got in zip(want, orders):
            assert [index[nm] for nm in members] == got.tolist(), "set-order replay diverged from CPython"

    def find_interval_i(self, ai, ccid):
        """Breadth-first search for intervals connected to interval ``ai`` by breakpoint edges (ibg:343-673)."""
        by = self.cns_intervals_by_chr
        half = int(self.max_seq_len / 2)
        D = self.interval_delta
        T = self._chim
        queue = [ai]
        while queue:
            cur = queue.pop(0)
            chrom, s, e = self.amplicon_intervals[cur][:3]
            if self.amplicon_intervals[cur][3] == -1:
                self.amplicon_intervals[cur][3] = ccid
            logging.debug(_t() + "\t\tNext amplicon interval %d: %s." % (cur, self.amplicon_intervals[cur]))
            prepared = self._search_step(chrom, s, e)
            if prepared is None:
                continue
            plan, all_cands, called = prepared
            here = self.amplicon_intervals[cur]            # not modified before all groups are done (ibg:385-612)
            refined, refined_bps = [], []
            for gi, (c, b0, b1) in enumerate(plan):
                ns, ne = by[c][b0][1], by[c][b1][2]
                tgt = [c, ns, ne]
                cands = all_cands[gi]
                logging.debug(_t() + "\t\tFound %d reads connecting the two intervals." % len(cands))
                found = []
                for bp, support, st in self._call_breakpoints(cands, advance_subcluster=False, called=called[gi]):
                    k = self.addbp(bp, support, st, ccid)
                    if k not in found:
                        found.append(k)
                inside, outside = [], []
                for k in found:
                    bp = self.new_bp_list[k][:6]
                    e1, e2 = [bp[0], bp[1], bp[1]], [bp[3], bp[4], bp[4]]
                    try:
                        if interval_overlap(e1, here) and interval_overlap(e2, tgt):
                            inside.append([self.pos2cni(bp[3], bp[4])[0], bp[4], k])
                        elif interval_overlap(e2, here) and interval_overlap(e1, tgt):
                            inside.append([self.pos2cni(bp[0], bp[1])[0], bp[1], k])
                        else:
                            logging.warning(_t() + "\t\tExact breakpoint outside amplicon interval.")
                            o1, o2 = interval_overlap(e1, tgt), interval_overlap(e2, tgt)
                            if o1:
                                inside.append([self.pos2cni(bp[0], bp[1])[0], bp[1], k])
                            else:
                                outside.append([bp[0], self.pos2cni(bp[0], bp[1])[0], bp[1], k])
                            if o2:
                                inside.append([self.pos2cni(bp[3], bp[4])[0], bp[4], k])
                            else:
                                outside.append([bp[3], self.pos2cni(bp[3], bp[4])[0], bp[4], k])
                    except Exception:
                        pass
                if not found:
                    continue
                inside.sort(key=lambda t: (t[0], t[1]))
                outside.sort(key=lambda t: (chr_idx[t[0]], t[1], t[2]))
                segs = by[c]
                gain = self.cn_gain

                def split_inside(k):
                    nil, ncn = segs[inside[k + 1][0]][1], segs[inside[k + 1][0]][3]
                    lir, lcn = segs[inside[k][0]][2], segs[inside[k][0]][3]
                    amp = ncn >= gain or lcn >= gain
                    dpos = inside[k + 1][1] - inside[k][1]
                    return (inside[k + 1][0] - inside[k][0] > 2 or nil - lir > self.max_seq_len / 2 or
                            dpos > self.max_seq_len or (not amp and nil - lir > 2 * D) or (not amp and dpos > 3 * D))

                first = 0
                for k in range(len(inside) - 1):
                    if not split_inside(k):
                        continue
                    f, z = inside[first], inside[k]
                    lir = segs[z[0]][2]
                    l = max((f[1] if not segs[f[0]][3] >= gain else segs[f[0]][1]) - D, segs[0][1])
                    r = min((z[1] if not segs[z[0]][3] >= gain else lir) + D, segs[-1][2])
                    if segs[f[0]][3] and f[1] - half > l:            # Q3: CN value used as a truth value
                        l = f[1] - half
                    if z[1] + half < r:
                        r = z[1] + half
                    if not self.pos2cni(c, l):
                        l = segs[f[0]][1]
                    if not self.pos2cni(c, r):
                        r = lir
                    refined.append([c, l, r, -1])
                    refined_bps.append([inside[j][2] for j in range(first, k + 1)])
                    first = k + 1
                if inside:
                    f, z = inside[first], inside[-1]
                    l = max((f[1] if not segs[f[0]][3] >= gain else segs[f[0]][1]) - D, segs[0][1])
                    r = min((z[1] if not segs[z[0]][3] >= gain else segs[z[0]][2]) + D, segs[-1][2])
                    if f[1] - half > l:
                        l = f[1] - half > l                         # Q2: the comparison result is stored
                    if z[1] + half < r:
                        r = z[1] + half
                    if not self.pos2cni(c, l):
                        l = segs[f[0]][1]
                    if not self.pos2cni(c, r):
                        r = segs[z[0]][2]
                    refined.append([c, l, r, -1])
                    refined_bps.append([inside[j][2] for j in range(first, len(inside))])

                def split_outside(k):
                    a, b = outside[k], outside[k + 1]
                    nil, ncn = by[b[0]][b[1]][1], by[b[0]][b[1]][3]
                    lir, lcn = by[a[0]][a[1]][2], by[a[0]][a[1]][3]
                    amp = ncn >= gain or lcn >= gain
                    return (b[0] != a[0] or b[1] - a[1] > 2 or nil - lir > self.max_seq_len / 2 or
                            b[2] - a[2] > self.max_seq_len or (not amp and nil - lir > 2 * D) or
                            (not amp and b[2] - a[2] > 3 * D))

                first = 0
                for k in range(len(outside) - 1):
                    if not split_outside(k):
                        continue
                    f, z = outside[first], outside[k]
                    lir = by[z[0]][z[1]][2]
                    l = max((f[2] if not by[f[0]][f[1]][3] >= gain else by[f[0]][f[1]][1]) - D, by[f[0]][0][1])
                    r = min((z[2] if not by[z[0]][z[1]][3] >= gain else lir) + D, by[z[0]][-1][2])
                    if f[2] - half > l:
                        l = f[2] - half
                    if z[2] + half < r:
                        r = z[2] + half
                    if not self.pos2cni(f[0], l):
                        l = by[f[0]][f[1]][1]
                    if not self.pos2cni(z[0], r):
                        r = lir
                    refined.append([f[0], l, r, -1])
                    refined_bps.append([])
                    first = k + 1
                if outside:
                    f, z = outside[first], outside[-1]
                    l = max((f[2] if not by[f[0]][f[1]][3] >= gain else by[f[0]][f[1]][1]) - D, by[f[0]][0][1])
                    r = min((z[2] if not by[z[0]][z[1]][3] >= gain else by[z[0]][z[1]][2]) + D, by[z[0]][-1][2])
                    if f[2] - half > l:
                        l = f[2] - half
                    if z[2] + half < r:
                        r = z[2] + half
                    if not self.pos2cni(f[0], l):
                        l = by[f[0]][f[1]][1]
                    if not self.pos2cni(f[0], r):
                        r = by[f[0]][z[1]][2]
                    refined.append([f[0], l, r, -1])
                    refined_bps.append([])

            conn = self.amplicon_interval_connections
            for ni, cand_iv in enumerate(refined):
                hit, parts = interval_exclusive(cand_iv, self.amplicon_intervals)
                if not parts:
                    for k in refined_bps[ni]:
                        bp = self.new_bp_list[k][:6]
                        for o in hit:
                            if (o != cur and interval_overlap([bp[0], bp[1], bp[1]], self.amplicon_intervals[o])) or \
                                    interval_overlap([bp[3], bp[4], bp[4]], self.amplicon_intervals[o]):       # Q5
                                conn.setdefault((min(cur, o), max(cur, o)), set()).add(k)
                    for o in hit:
                        if o != cur and self.amplicon_intervals[o][3] < 0:
                            queue.append(o)
                            self._prefetch_step(o)
                else:
                    for part in parts:
                        nai = len(self.amplicon_intervals)
                        self.amplicon_intervals.append(part)
                        logging.debug(_t() + "\t\tAdded new interval %s to the amplicon interval list." % part)
                        conn[(cur, nai)] = set()
                        for k in refined_bps[ni]:
                            if not hit:
                                conn[(cur, nai)].add(k)
                                continue
                            bp = self.new_bp_list[k][:6]
                            for o in hit:
                                if interval_overlap([bp[0], bp[1], bp[1]], self.amplicon_intervals[o]) or \
                                        interval_overlap([bp[3], bp[4], bp[4]], self.amplicon_intervals[o]):
                                    conn.setdefault((min(cur, o), max(cur, o)), set()).add(k)
                                else:
                                    conn[(cur, nai)].add(k)
                        queue.append(nai)
                        self._prefetch_step(nai)

    def _add_clustered(self, cands: Candidates):
        """Tail shared by find_breakpoints and find_smalldel_breakpoints (ibg:691-718, ibg:775-802)."""
        for bp, support, st in self._call_breakpoints(cands, advance_subcluster=True):
            io1 = interval_overlap_l([bp[0], bp[1], bp[1]], self.amplicon_intervals)
            io2 = interval_overlap_l([bp[3], bp[4], bp[4]], self.amplicon_intervals)
            if io1 >= 0 and io2 >= 0:
                assert self.amplicon_intervals[io1][3] == self.amplicon_intervals[io2][3]
                k = self.addbp(bp, support, st, self.amplicon_intervals[io1][3])
                self.amplicon_interval_connections.setdefault((min(io1, io2), max(io1, io2)), set()).add(k)

    # ---- A6 ----------------------------------------------------------------------------------
    def find_smalldel_breakpoints(self):
        """Large deletions inside single alignment records (ibg:721-802), from the gap rows of coral_cigar_scan."""
        dr = self.rec
        sc = self.scan()
        g = sc.gaps
        parts = []
        if len(g):
            g_rec = g[:, 0].astype(np.int64)
            ivs = [(self._tid_of[iv[0]], int(iv[1]), int(iv[2])) for iv in self.amplicon_intervals]
            parts.append(rows_by_interval(dr.h_tid[g_rec], dr.h_pos[g_rec], dr.h_end[g_rec], ivs))
        sel = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
        cands = Candidates()
        if len(sel):
            rec = g[sel, 0].astype(np.int64)
            b0, b1 = g[sel, 4], g[sel, 5]
            nid = dr.h_name_id[rec].astype(np.int64)
            nxt, prv = g[sel, 3].astype(np.int64), g[sel, 2].astype(np.int64)
            # dict insertion order = first appearance of the name; list order = appearance order
            u, first, inv = np.unique(nid, return_index=True, return_inverse=True)
            rank = np.empty(len(u), dtype=np.int64)
            rank[np.argsort(first, kind="stable")] = np.arange(len(u))
            o = np.argsort(rank[inv], kind="stable")
            grp = rank[inv][o]
            k_in = np.arange(len(o)) - np.searchsorted(grp, grp, side="left")
            ro = rec[o]
            starts = np.nonzero(k_in == 0)[0]
            # name -> [[chr, next block start, previous block end, first block start, last block end, mapq], ...] (ibg:746-762),
            # names in first-appearance order; the Python lists are made when the dict is first looked into
            self.large_indel_alignments = _LazyIndelAlignments(self, nid[o][starts], np.append(starts, len(o)), dr.h_tid[ro], nxt[o], prv[o],
                                                               b0[o], b1[o], dr.h_mapq[ro])
            a = nxt[o]
            b = np.minimum(prv[o], a)                                            # aliasing swap (Q7)
            tid = dr.h_tid[rec[o]].astype(np.int64)
            z = np.zeros(len(o), dtype=np.int64)
            cands = Candidates(c1=tid, p1=a, o1=z + 1, c2=tid, p2=b, o2=z, read=nid[o], i=k_in, j=k_in, gap=z, swapped=z,
                               mqa=z - 1, mqb=z - 1)
        logging.info(_t() + "Fetched %d reads with large indels in CIGAR." % (len(self.large_indel_alignments)))
        self._add_clustered(cands)

    # ---- A7 ----------------------------------------------------------------------------------
    def find_breakpoints(self):
        """Breakpoints from chimeric alignments inside the amplicon intervals (ibg:676-718)."""
        T = self._chim
        ivs = [(self._tid_of[iv[0]], iv[1], iv[2]) for iv in self.amplicon_intervals]
        cands = self._search().within(ivs)              # alignment2bp_l as a filter over the pair table (cutoffs 100 / 20 / 100 / 10)
        logging.debug(_t() + "Found %d reads with new breakpoints." % (len(cands)))
        self._add_clustered(cands)

    # ---- A9 ----------------------------------------------------------------------------------
    def build_graph(self, graph_class=BreakpointGraph):
        """Split the intervals at breakpoint ends and assemble one BreakpointGraph per amplicon (ibg:864-1016)."""
        cuts: Dict[int, list] = {}
        for k, bp in enumerate(self.new_bp_list):
            for ai, seg in enumerate(self.amplicon_intervals):
                for (ci, pi, oi) in ((0, 1, 2), (3, 4, 5)):
                    if bp[ci] == seg[0] and seg[1] < bp[pi] < seg[2]:
                        if bp[oi] == '+':
                            cuts.setdefault(ai, []).append((bp[pi], bp[pi] + 1, k, pi, '+'))
                        if bp[oi] == '-':
                            cuts.setdefault(ai, []).append((bp[pi] - 1, bp[pi], k, pi, '-'))
        nxt = 1
        for seg in self.amplicon_intervals:
            if seg[3] not in self.ccid2id:
                self.ccid2id[seg[3]] = nxt
                nxt += 1
        for _ in range(len(self.ccid2id)):
            self.lr_graph.append(graph_class())
        graph_of = lambda seg: self.lr_graph[self.ccid2id[seg[3]] - 1]
        for ai in cuts:
            cuts[ai].sort(key=lambda t: t[0])
            seg = self.amplicon_intervals[ai]
            g, c = graph_of(seg), seg[0]
            prev_cut = None
            for cut in cuts[ai]:
                if prev_cut is None:
                    left = seg[1]
                elif cut[0] > prev_cut[0]:
                    left = prev_cut[1]
                else:
                    prev_cut = cut
                    continue
                g.add_node((c, left, '-'))
                g.add_node((c, cut[0], '+'))
                g.add_node((c, cut[1], '-'))
                g.add_sequence_edge(c, left, cut[0])
                g.add_concordant_edge(c, cut[0], '+', c, cut[1], '-')
                prev_cut = cut
            g.add_node((c, cuts[ai][-1][1], '-'))
            g.add_node((c, seg[2], '+'))
            g.add_sequence_edge(c, cuts[ai][-1][1], seg[2])
        for ai, seg in enumerate(self.amplicon_intervals):
            if ai not in cuts:
                g = graph_of(seg)
                g.add_node((seg[0], seg[1], '-'))
                g.add_node((seg[0], seg[2], '+'))
                g.add_sequence_edge(seg[0], seg[1], seg[2])
        for g in self.lr_graph:
            g.sort_edges()
        for seg in self.amplicon_intervals:
            g = graph_of(seg)
            g.amplicon_intervals.append([seg[0], seg[1], seg[2]])
            g.add_endnode((seg[0], seg[1], '-'))
            g.add_endnode((seg[0], seg[2], '+'))
        for k, bp in enumerate(self.new_bp_list):
            io1 = interval_overlap_l([bp[0], bp[1], bp[1]], self.amplicon_intervals)
            io2 = interval_overlap_l([bp[3], bp[4], bp[4]], self.amplicon_intervals)
            assert self.amplicon_intervals[io1][3] == self.amplicon_intervals[io2][3]
            cc = self.amplicon_intervals[io1][3]
            if cc != self.new_bp_ccids[k]:
                self.new_bp_ccids[k] = cc
            self.lr_graph[self.ccid2id[cc] - 1].add_discordant_edge(bp[0], bp[1], bp[2], bp[3], bp[4], bp[5],
                                                                     lr_count=len(bp[-1]), reads=bp[-1])
        for srci, srce in enumerate(self.source_edges):
            self.lr_graph[self.ccid2id[self.source_edge_ccids[srci]] - 1].add_source_edge(srce[3], srce[4], srce[5])

    # ---- A10 ---------------------------------------------------------------------------------
    def assign_cov(self):
        """Long-read coverage of every sequence edge and read support of every concordant edge (ibg:1019-1056)."""
        todo = [(g, k) for g in self.lr_graph for k, e in enumerate(g.sequence_edges) if e[5] == -1]
        if todo:
            n_reads, n_bases = self._coverage([g.sequence_edges[k][:3] for g, k in todo])
            for (g, k), nr, nb in zip(todo, n_reads, n_bases):
                g.sequence_edges[k][5] = int(nr)
                g.sequence_edges[k][6] = int(nb)
        cut = self.min_bp_match_cutoff_
        edges = [(g, e) for g in self.lr_graph for e in g.concordant_edges]
        pts = []
        for g, e in edges:
            t1, t2 = self._tid_of[e[0]], self._tid_of[e[3]]
            pts += [(t1, e[1]), (t2, e[4]), (t1, e[1] - cut - 1), (t2, e[4] + cut)]
        cover = kernels.point_cover(self.rec, pts)
        if not edges:
            return
        nid = self.rec.h_name_id
        names = self.rec.names
        # one native pass over all edges (coral_concordant_counts): the four fetches of an edge as slices of ONE array of record
        # ordinals, the reads of the discordant edges at its two nodes as name ids
        sup_parts, sup_off, by_name = [], [0], {}
        for q, (g, e) in enumerate(edges):
            e[9] = ReadNameSet(names, nid, cover[4 * q], cover[4 * q + 1], keep=cover)      # rls | rrs (ibg:1054) — name strings only on demand
            n_here = 0
            for node in ((e[0], e[1], e[2]), (e[3], e[4], e[5])):
                for k in g.nodes[node][2]:
                    s_ = g.discordant_edges[k][10]
                    ids = s_.name_ids() if isinstance(s_, ReadSupportSet) else None
                    if ids is None:                                     # somebody materialised / replaced a support set: by name
                        by_name.setdefault(q, set()).update(t[0] for t in s_)
                    else:
                        sup_parts.append(ids)
                        n_here += len(ids)
            sup_off.append(sup_off[-1] + n_here)
        sup = np.ascontiguousarray(np.concatenate(sup_parts), dtype=np.int64) if sup_parts else np.zeros(0, dtype=np.int64)
        sup_off = np.asarray(sup_off, dtype=np.int64)
        count = np.zeros(len(edges), dtype=np.int64)
        nid32 = np.ascontiguousarray(nid, dtype=np.int32)
        pt_rec = np.ascontiguousarray(cover.rec, dtype=np.int32)
        pt_begin, pt_end = np.ascontiguousarray(cover.begin, dtype=np.int64), np.ascontiguousarray(cover.end, dtype=np.int64)
        _lib.check(_lib.lib().coral_concordant_counts(len(edges), pt_begin.ctypes.data, pt_end.ctypes.data, pt_rec.ctypes.data, len(pt_rec),
                                                      nid32.ctypes.data, len(nid32), self.rec.n_names, sup_off.ctypes.data, sup.ctypes.data,
                                                      count.ctypes.data), "coral_concordant_counts")
        for q, (g, e) in enumerate(edges):
            e[8] = int(count[q])
            if q in by_name:                                            # rare: redo this edge with the names themselves
                sets = [set(self._names_of(np.unique(nid[cover[4 * q + d]]))) for d in range(4)]
                e[8] = len((sets[0] & sets[1] & sets[2] & sets[3]) - by_name[q] -
                           {names[i] for i in sup[sup_off[q]:sup_off[q + 1]].tolist()})

    # ---- SURVEY.md §8(f) item 2 -------------------------------------------------------------------
    def compute_path_constraints(self):
        """Reads -> subpath constraints per amplicon (ibg:1059-1323); called by the cycle step (cd:2067)."""
        from . import path_constraints as pc
        cutoff = self.min_bp_match_cutoff_
        for amplicon_idx, g in enumerate(self.lr_graph):
            store = [[], [], []]
            self.path_constraints[amplicon_idx] = store
            self.longest_path_constraints[amplicon_idx] = [[], [], []]

            def add_path(path, label):
                if len(path) > 5 and pc.valid_path(g, path):
                    if path in store[0]:
                        store[1][store[0].index(path)] += 1
                    elif path[::-1] in store[0]:
                        store[1][store[0].index(path[::-1])] += 1
                    else:
                        store[0].append(path)
                        store[1].append(1)
                        store[2].append(label)

            # reads supporting discordant edges: [chimeric (i != j) entries, small-deletion (i == j) entries]
            bp_reads: Dict[str, list] = {}
            for di, bp in enumerate(g.discordant_edges):
                for r_ in bp[10]:
                    slot = 1 if r_[1] == r_[2] else 0
                    bp_reads.setdefault(r_[0], [[], []])[slot].append([r_[1], r_[2], di])
            for rn, (chim, sdel) in bp_reads.items():
                paths = []
                if len(chim) == 1 and not sdel:
                    rints = [a[:4] for a in self.chimeric_alignments[rn][1]]
                    paths.append(pc.chimeric_alignment_to_path_i(g, rints, chim[0][0], chim[0][1], chim[0][2]))
                elif len(chim) > 1 and not sdel:
                    chim = sorted(chim, key=lambda it: min(it[0], it[1]))
                    if self._overlapping_query_intervals(rn, cutoff):
                        continue
                    for block in self._chain_blocks(chim):
                        bps = [chim[b][2] for b in block]
                        if len(set(bps)) < len(bps):
                            continue
                        rints = [a[:4] for a in self.chimeric_alignments[rn][1]]
                        paths.append(pc.chimeric_alignment_to_path(g, rints, [chim[b][:2] for b in block], bps))
                elif not chim and len(sdel) == 1:
                    rints = self.large_indel_alignments[rn][0]
                    if rints[3] < rints[4]:
                        if not rints[2] < rints[1]:
                            continue
                        rints = [[rints[0], rints[3], rints[2], '+'], [rints[0], rints[1], rints[4], '+']]
                    else:
                        if not rints[2] > rints[1]:
                            continue
                        rints = [[rints[0], rints[3], rints[2], '-'], [rints[0], rints[1], rints[4], '-']]
                    di = sdel[0][2]
                    if rints[0][3] == '+':
                        paths.append(pc.chimeric_alignment_to_path_i(g, rints, 1, 0, di))
                    else:
                        paths.append(pc.chimeric_alignment_to_path_i(g, rints, 0, 1, di))
                elif not chim and len(sdel) > 1:
                    gaps = self.large_indel_alignments[rn]
                    spans = set((x[0], min(x[3], x[4]), max(x[3], x[4])) for x in gaps)
                    if len(spans) > 1 or len(gaps) <= 1:
                        continue
                    pieces = [[x[0], min(x[3], x[4]), max(x[3], x[4]), '+'] for x in gaps]
                    gaps = sorted(gaps, key=lambda it: min(it[1], it[2]))
                    for ri, x in enumerate(gaps):
                        pieces.append([x[0], min(x[3], x[4]), max(x[3], x[4]), '+'])
                        pieces[ri][2] = min(x[1], x[2])
                        pieces[ri + 1][1] = max(x[1], x[2])
                    sdel = sorted(sdel, key=lambda it: it[0])
                    blocks, last = [[]], 0
                    for i, it in enumerate(sdel):
                        if i == 0 or it[0] == last + 1:
                            blocks[-1].append(i)
                        else:
                            blocks.append([i])
                        last = it[0]
                    for block in blocks:
                        bps = [sdel[b][2] for b in block]
                        if len(set(bps)) < len(bps):
                            continue
                        paths.append(pc.chimeric_alignment_to_path(g, pieces, [[sdel[b][0], sdel[b][0] + 1] for b in block], bps))
                else:
                    rints = [a[:4] for a in self.chimeric_alignments[rn][1]]
                    gaps = self.large_indel_alignments[rn]
                    split_at = []
                    for x in gaps:
                        for ri, seg in enumerate(rints):
                            if x[0] == seg[0] and min(x[1], x[2]) > min(seg[1], seg[2]) and max(x[1], x[2]) < max(seg[1], seg[2]):
                                split_at.append(ri)
                                break
                        else:
                            split_at = None
                            break
                    if split_at is None:
                        continue
                    for rsi, ri in enumerate(split_at):
                        rints.insert(ri, rints[ri][:])
                        x = gaps[rsi]
                        if rints[ri][3] == '+':
                            rints[ri][2] = min(x[1], x[2])
                            rints[ri + 1][1] = max(x[1], x[2])
                        else:
                            rints[ri][2] = max(x[1], x[2])
                            rints[ri + 1][1] = min(x[1], x[2])
                        for it in chim:
                            if it[0] >= ri and it[1] >= ri:
                                it[0] += 1
                                it[1] += 1
                        for it in sdel:
                            if it[0] == rsi:
                                chim.append([ri + 1, ri, it[2]] if rints[ri][3] == '+' else [ri, ri + 1, it[2]])
                    chim = sorted(chim, key=lambda it: min(it[0], it[1]))
                    blocks = self._chain_blocks(chim)
                    if self._overlapping_query_intervals(rn, cutoff):
                        continue
                    for block in blocks:
                        bps = [chim[b][2] for b in block]
                        if len(set(bps)) < len(bps):
                            continue
                        paths.append(pc.chimeric_alignment_to_path(g, rints, [chim[b][:2] for b in block], bps))
                for path in paths:
                    add_path(path, amplicon_idx)
            logging.debug(_t() + "There are %d distinct subpaths due to reads involving breakpoints in amplicon %d."
                          % (len(store[0]), amplicon_idx + 1))

            # reads without breakpoints: every alignment record of the amplicon's intervals (ibg:1296-1321)
            dr = self.rec
            is_conc = np.zeros(dr.n_names, dtype=bool)
            by_name = []                                   # name sets somebody materialised or replaced: looked up by name
            for ce in g.concordant_edges:
                ids = ce[9].name_ids() if isinstance(ce[9], ReadNameSet) else None
                if ids is not None:
                    is_conc[ids] = True
                else:
                    by_name.append(ce[9])
            if by_name:
                name_to_id = self._name_ids()
                for rn in set().union(*by_name):
                    is_conc[name_to_id[rn]] = True
            if is_conc.any():                              # minus the reads with breakpoints (ibg:1299)
                lia = self.large_indel_alignments
                if lia:
                    name_to_id = self._name_ids()
                    is_conc[np.fromiter((name_to_id[rn] for rn in lia), dtype=np.int64, count=len(lia))] = False
                if self._chim is not None and len(self._chim.name_id) and len(self.chimeric_alignments) == len(self._chim.name_id):
                    is_conc[self._chim.name_id] = False
                else:
                    name_to_id = self._name_ids()
                    for rn in self.chimeric_alignments:
                        is_conc[name_to_id[rn]] = False
            if is_conc.any():
                for aint in self.amplicon_intervals:
                    if amplicon_idx != self.ccid2id[aint[3]] - 1:
                        continue
                    recs = dr.region(self._tid_of[aint[0]], aint[1], aint[2] + 1)
                    recs = recs[(dr.h_mapq[recs] >= 20) & is_conc[dr.h_name_id[recs]]]
                    if not len(recs):
                        continue
                    start, end = dr.h_pos[recs].astype(np.int64), dr.h_end[recs].astype(np.int64)
                    lo, hi, order = pc.classify_alignments(g, aint[0], start, end)
                    keep = lo <= hi
                    code = lo[keep] * (len(order) + 1) + hi[keep]
                    u, first, cnt = np.unique(code, return_index=True, return_counts=True)
                    for k in np.argsort(first, kind="stable"):          # first appearance in fetch order
                        a = g.sequence_edges[order[int(u[k]) // (len(order) + 1)]]
                        b = g.sequence_edges[order[int(u[k]) % (len(order) + 1)]]
                        path = pc.traverse_through_sequence_edge(g, (a[0], a[1], '-'), (b[0], b[2], '+'))[1:-1]
                        for _ in range(int(cnt[k])):
                            add_path(path, amplicon_idx)
            logging.debug(_t() + "There are %d distinct subpaths in total in amplicon %d." % (len(store[0]), amplicon_idx + 1))

    def _name_ids(self):
        return self.rec.names.index_map()

    @staticmethod
    def _chain_blocks(entries):
        """Consecutive (i, j) entries that share an alignment index form one block (ibg:1102-1109)."""
        blocks = [[0]]
        last = max(entries[0][0], entries[0][1])
        for i in range(1, len(entries)):
            if min(entries[i][0], entries[i][1]) == last:
                blocks[-1].append(i)
            else:
                blocks.append([i])
            last = max(entries[i][0], entries[i][1])
        return blocks

    def _overlapping_query_intervals(self, rn, cutoff):
        """True when two consecutive local alignments of the read overlap by more than ``cutoff`` on the read (ibg:1112-1117)."""
        qints = self.chimeric_alignments[rn][0]
        return any(qints[k + 1][0] - qints[k][1] < -cutoff for k in range(len(qints) - 1))

    def closebam(self):
        self.lr_bamfh.close()


class _LazyIndelAlignments(dict):
    """``large_indel_alignments``: read name -> list of ``[chr, next block start, previous block end, first block start, last
    block end, mapq]`` (ibg:746-762), names in first-appearance order.  Built from arrays on first use; only ``len()`` is
    answered without building."""

    def __init__(self, owner, group_name_ids, bounds, tid, nxt, prv, b0, b1, mapq):
        super().__init__()
        self._src = (weakref.proxy(owner), group_name_ids, bounds, tid, nxt, prv, b0, b1, mapq)
        self._n = len(group_name_ids)

    def _fill(self):
        if self._src is not None:
            o, gids, bounds, tid, nxt, prv, b0, b1, mapq = self._src
            self._src = None
            chroms = o.rec.header_chroms
            rows = list(map(list, zip([chroms[t] for t in tid.tolist()], nxt.tolist(), prv.tolist(), b0.tolist(), b1.tolist(),
                                      mapq.tolist())))
            names = o._names_of(gids)
            bnd = bounds.tolist()
            dict.update(self, {nm: rows[bnd[k]:bnd[k + 1]] for k, nm in enumerate(names)})

    def __len__(self):
        return self._n if self._src is not None else dict.__len__(self)

    def __bool__(self):
        return len(self) > 0

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._fill()
        return dict.get(self, k, default)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def __setitem__(self, k, v):
        self._fill()
        dict.__setitem__(self, k, v)

    def __delitem__(self, k):
        self._fill()
        dict.__delitem__(self, k)

    def setdefault(self, k, default=None):
        self._fill()
        return dict.setdefault(self, k, default)

    def pop(self, k, *default):
        self._fill()
        return dict.pop(self, k, *default)

    def update(self, *a, **kw):
        self._fill()
        dict.update(self, *a, **kw)

    def copy(self):
        self._fill()
        return dict(self)

    def __eq__(self, other):
        self._fill()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)

    def __reduce__(self):
        self._fill()
        return (dict, (list(dict.items(self)),))


class _LazyReadLength(dict):
    """``read name -> query length`` for reads with a primary record (ibg:141-143); built from the chimeric table's per-name
    array — which stays on the device until then — on first use."""

    def __init__(self, names, table):
        super().__init__()
        self._names, self._table, self._done, self._has = names, table, False, None

    def _which(self):
        if self._has is None:
            self._has = np.nonzero(self._table.read_length >= 0)[0]
        return self._has

    def _fill(self):
        if not self._done:
            self._done = True
            has = self._which()
            dict.update(self, zip(self._names.take(has), self._table.read_length[has].tolist()))
            self._table = None

    def __len__(self):
        return dict.__len__(self) if self._done else len(self._which())

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        self._fill()
        return dict.get(self, k, default)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def __eq__(self, other):
        self._fill()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)

    def copy(self):
        self._fill()
        return dict(self)

    def __reduce__(self):
        self._fill()
        return (dict, (list(dict.items(self)),))


class _SegIndexView:
    """``chimeric_alignments_seg[chr][cni] -> [read names]`` view over the sorted inverted index."""

    def __init__(self, owner):
        self._o = weakref.proxy(owner)

    def as_dict(self):
        o = self._o
        T = o._chim
        chroms = o.rec.header_chroms
        out: Dict[str, Dict[int, list]] = {}
        # chromosome key order: first appearance along (read, segment) order
        for t in o._seg_tids:
            out[chroms[t]] = {}
        for row, key in zip(o._e_row.tolist(), o._e_key.tolist()):
            out[chroms[T.tid[row]]].setdefault(key & 0xFFFFFFFF, []).append(o.chimeric_alignments._names[T.read[row]])
        return out

    def __contains__(self, c):
        o = self._o
        return c in o._tid_of and o._tid_of[c] in o._seg_tids


_HEAP_FROZEN = False
PHASE_SECONDS: Dict[str, float] = {}      # wall time of every phase of the last build (same phases the reference logs)


def build_graph_from_records(records, seedfile, cn_seg, output_prefix=None, min_bp_support=1.0, output_bp=False,
                             graph_class=BreakpointGraph, gc_policy="pause"):
    """The call sequence of reconstruct_graph (ibg:1349-1394) on already-decoded records.

    The build allocates a few hundred thousand small containers (read tuples, name sets) and no reference cycles, so the
    cyclic garbage collector can only cost time here.  ``gc_policy``:
      "pause"   (default) collector off during the build, back on afterwards; the first build of a process also moves the heap
                that exists at that moment (imported modules) to the permanent generation, once (``gc.freeze()``: nothing is
                collected, later full collections just stop re-walking it);
      "freeze"  as "pause", then ``gc.freeze()``: the result (and whatever else the process holds at that moment) moves to
                the permanent generation and is never walked again; everything is still freed by reference counting
                (the result holds no reference cycles).  For processes whose main job is this build; measured no faster
                than "pause" once the result stopped being cyclic garbage;
      "none"    leave the collector alone.
    """
    import gc
    if gc_policy not in ("pause", "freeze", "none"):
        raise ValueError("gc_policy must be 'pause', 'freeze' or 'none'")
    gc_was_enabled = gc.isenabled()
    if gc_policy != "none":
        gc.disable()
        global _HEAP_FROZEN
        if not _HEAP_FROZEN:
            # once per process: what exists before the first build (the imported modules of torch, numpy, ... — half a million
            # container objects) moves to the collector's permanent generation.  Nothing is collected or skipped by this; it only
            # stops every later full collection (one per ~10 builds in a loop of builds) from walking those objects again, which
            # cost a build-sized pause each time (measured: 2 of 20 steps at 61 and 82 ms against 44 ms).
            gc.freeze()
            _HEAP_FROZEN = True
    try:
        hostpools.apply_once()
        b2bn = _build_graph_from_records(records, seedfile, cn_seg, output_prefix, min_bp_support, output_bp, graph_class)
        if gc_policy == "freeze":
            gc.freeze()
        return b2bn
    finally:
        if gc_was_enabled and gc_policy != "none":
            gc.enable()


def _build_graph_from_records(records, seedfile, cn_seg, output_prefix, min_bp_support, output_bp, graph_class):
    clock = time.perf_counter
    t0 = clock()

    def lap(name, msg):
        nonlocal t0
        t1 = clock()
        PHASE_SECONDS[name] = t1 - t0
        t0 = t1
        logging.info(_t() + msg)

    PHASE_SECONDS.clear()
    b2bn = bam_to_breakpoint_nanopore(None, seedfile, records=records)
    b2bn.min_bp_cov_factor = min_bp_support
    if os.environ.get("CORAL_NO_EARLY_LAUNCH") != "1":
        b2bn.launch_record_kernels()
    lap("open", "Opened LR bam files.")
    b2bn.read_cns(cn_seg)
    lap("read_cns", "Completed parsing CN segment files.")
    b2bn.fetch()
    lap("fetch", "Completed fetching reads containing breakpoints.")
    b2bn.hash_alignment_to_seg()
    lap("hash_alignment_to_seg", "Completed hashing chimeric reads to CN segments.")
    b2bn.find_amplicon_intervals()
    lap("find_amplicon_intervals", "Completed finding amplicon intervals.")
    b2bn.find_smalldel_breakpoints()
    lap("find_smalldel_breakpoints", "Completed finding small del breakpoints.")
    b2bn.find_breakpoints()
    lap("find_breakpoints", "Completed finding all discordant breakpoints.")
    b2bn.build_graph(graph_class)
    lap("build_graph", "Breakpoint graph built for all amplicons.")
    if output_bp:
        for gi in range(len(b2bn.lr_graph)):
            bp_stats_i = []
            for de in b2bn.lr_graph[gi].discordant_edges:
                for bpi in range(len(b2bn.new_bp_list)):
                    if de[:6] == b2bn.new_bp_list[bpi][:6]:
                        bp_stats_i.append(b2bn.new_bp_stats[bpi])
                        break
            if output_prefix is not None:
                output_breakpoint_info_lr(b2bn.lr_graph[gi], output_prefix + "_amplicon" + str(gi + 1) + "_breakpoints.txt",
                                          bp_stats_i)
        lap("output", "Wrote breakpoint information, for all amplicons, to %s." % (str(output_prefix) + '_amplicon*_breakpoints.txt'))
    else:
        b2bn.assign_cov()
        lap("assign_cov", "Fetched read coverage for all sequence and concordant edges.")
        for g in b2bn.lr_graph:
            compute_cn_lr(g, b2bn.normal_cov)
        lap("compute_cn_lr", "Computed CN for all edges.")
        if output_prefix is not None:
            for gi in range(len(b2bn.lr_graph)):
                output_breakpoint_graph_lr(b2bn.lr_graph[gi], output_prefix + "_amplicon" + str(gi + 1) + "_graph.txt")
        lap("output", "Wrote breakpoint graph for all complicons to %s." % (str(output_prefix) + '_amplicon*_graph.txt'))
    return b2bn


def reconstruct_graph(args):
    """Same contract as the reference's reconstruct_graph(args) (ibg:1333-1395)."""
    global_names.TSTART = time.time()
    log_fn = "infer_breakpoint_graph.log"
    if getattr(args, "log_fn", None):
        log_fn = args.log_fn
    logging.basicConfig(filename=log_fn, filemode='w', level=logging.DEBUG, format='[%(name)s:%(levelname)s]\t%(message)s')
    logging.info("Python version " + sys.version + "\n")
    commandstring = 'Commandline: '
    for arg in sys.argv:
        commandstring += ('"{}" '.format(arg) if ' ' in arg else "{} ".format(arg))
    logging.info(_t() + commandstring)
    from .bam import load_bam
    from .records import DeviceRecords
    records = DeviceRecords(load_bam(args.lr_bam, getattr(args, "device", "cuda:0")), getattr(args, "device", "cuda:0"))
    return build_graph_from_records(records, args.cnv_seed, args.cn_seg, args.output_prefix, args.min_bp_support,
                                    args.output_bp, gc_policy=getattr(args, "gc_policy", "pause"))


def print_complete_message():
    logging.info(_t() + "Total runtime.")
