"""CPU oracle: a plain restatement of CoRAL's breakpoint-graph construction.

TEST INFRASTRUCTURE ONLY.  Nothing under ``coral_amd/`` imports this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, as the checker.

It restates, function by function, the reference's path
``reconstruct_graph`` (/root/reference/src/infer_breakpoint_graph.py:1333-1395) and what it calls
in cigar_parsing.py (``cp``), breakpoint_utilities.py (``bu``) and breakpoint_graph.py (``bg``),
operating on decoded records (``oracle.hostrecords.HostRecords``) instead of pysam.  Reference
quirks that are observable in the output are reproduced on purpose.

Pinned by: tests/golden/unit_vectors.json (cp/bu functions imported from the reference) and
tests/golden/e2e_*.json (the reference's own methods run phase by phase behind a fake pysam).
NOT pinned: pysam's own semantics (restated in hostrecords.py) and cvxopt's solver — CN values are
the exact optimum of the reference's objective (bg:546-563), "CN parity vs cvxopt unpinned".
"""
from __future__ import annotations

import math
from collections import Counter

import numpy as np

CHR_IDX = {c: i for i, c in enumerate([f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY", "chrM"])}  # gn:13-18
FLIP = {"+": "-", "-": "+"}                                                                        # gn:9


# ======================================================================================
# cigar_parsing.py
# ======================================================================================
def split_cigar(cigar):
    """'12S30M4D' -> ('SMD', [12, 30, 4])."""
    letters, nums, cur = [], [], ""
    for ch in cigar:
        if ch.isdigit():
            cur += ch
        else:
            letters.append(ch)
            nums.append(int(cur))
            cur = ""
    return "".join(letters), nums


def cigar2pos(cigar, strand, read_length):
    """(qs, qe, al) for one SA CIGAR — the nine shapes of cp:17-215, dispatch cp:219-229.

    Unknown shapes raise KeyError exactly as the reference's dict lookup does (cp:255).
    """
    shape, n = split_cigar(cigar)
    fwd = strand == "+"
    if shape == "SM":            # cp:17-36
        al = n[1]
        return (n[0], read_length - 1, al) if fwd else (0, al - 1, al)
    if shape == "MS":            # cp:39-58
        al = n[0]
        return (0, al - 1, al) if fwd else (n[1], read_length - 1, al)
    if shape == "SMS":           # cp:61-80
        al = n[1]
        qs = n[0] if fwd else n[2]
        return (qs, qs + al - 1, al)
    if shape == "SMD":           # cp:83-103
        al = n[1] + n[2]
        return (n[0], read_length - 1, al) if fwd else (0, n[1] - 1, al)
    if shape == "MDS":           # cp:106-126
        al = n[0] + n[1]
        return (0, n[0] - 1, al) if fwd else (n[2], read_length - 1, al)
    if shape == "SMDS":          # cp:129-149
        al = n[1] + n[2]
        return (n[0], read_length - n[3] - 1, al) if fwd else (n[3], read_length - n[0] - 1, al)
    if shape == "SMI":           # cp:152-171
        al = n[1]
        return (n[0], read_length - 1, al) if fwd else (0, read_length - n[0] - 1, al)
    if shape == "MIS":           # cp:174-193
        al = n[0]
        return (0, read_length - n[2] - 1, al) if fwd else (n[2], read_length - 1, al)
    if shape == "SMIS":          # cp:196-215
        al = n[1]
        return (n[0], read_length - n[3] - 1, al) if fwd else (n[3], read_length - n[0] - 1, al)
    raise KeyError(shape)


def alignment_from_satags(sa_list, read_length):
    """cp:232-269.  4-tuple (qint, rint, qual, nm) on success, the 3-tuple ([], [], []) on failure."""
    rows = []
    for sa in sa_list:
        f = sa.split(",")
        if "S" not in f[3] or "M" not in f[3]:           # cp:248-253
            return ([], [], [])
        qs, qe, al = cigar2pos(f[3], f[2], read_length)
        p = int(f[1])
        if f[2] == "+":
            r = [f[0], p - 1, p + al - 2, "+"]           # cp:258
        else:
            r = [f[0], p + al - 2, p - 1, "-"]           # cp:260
        rows.append(([qs, qe], r, int(f[4]), float(f[-1])))
    order = sorted(range(len(rows)), key=lambda i: (rows[i][0][0], rows[i][0][1]))   # cp:263 (stable)
    qint = [rows[i][0] for i in order]
    rint = [rows[i][1] for i in order]
    qual = [rows[i][2] for i in order]
    nm = [rows[i][3] / (rows[i][0][1] - rows[i][0][0]) for i in order]               # cp:268
    return (qint, rint, qual, nm)


# ======================================================================================
# breakpoint_utilities.py
# ======================================================================================
def interval_overlap(a, b):          # bu:11-15
    return a[0] == b[0] and int(a[1]) <= int(b[2]) and int(b[1]) <= int(a[2])


def interval_include(a, b):          # bu:18-22
    return a[0] == b[0] and int(a[1]) >= int(b[1]) and int(a[2]) <= int(b[2])


def interval_adjacent(a, b):         # bu:25-34
    if a[0] != b[0]:
        return False
    if a[1] <= b[1]:
        return b[1] == a[2] + 1
    return a[1] == b[2] + 1


def interval_overlap_l(a, lst):      # bu:37-44
    for k, b in enumerate(lst):
        if interval_overlap(a, b):
            return k
    return -1


def interval_exclusive(a, lst):      # bu:54-67
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


def interval2bp(R1, R2, r=(), rgap=0):                   # bu:289-295
    c1, c2 = CHR_IDX[R1[0]], CHR_IDX[R2[0]]
    if c2 < c1 or (c2 == c1 and R2[1] < R1[2]):
        return [R1[0], R1[2], R1[3], R2[0], R2[1], FLIP[R2[3]], r, rgap, 0]
    return [R2[0], R2[1], FLIP[R2[3]], R1[0], R1[2], R1[3], (r[0], r[2], r[1]), rgap, 1]


def alignment2bp(rn, ca, min_bp_match_cutoff, min_mapq, intrvl1, intrvl2, gap_mapq=10):
    """bu:70-96 — candidates between two given intervals (used inside the interval BFS)."""
    qi, ri, mq = ca[0], ca[1], ca[2]
    n = len(ri)
    out = []
    used = [0] * max(0, n - 1)
    for k in range(n - 1):
        gap = int(qi[k + 1][0]) - int(qi[k][1])
        if gap + min_bp_match_cutoff < 0 or mq[k] < min_mapq or mq[k + 1] < min_mapq:
            continue
        if (interval_overlap(ri[k], intrvl1) and interval_overlap(ri[k + 1], intrvl2)) or \
                (interval_overlap(ri[k + 1], intrvl1) and interval_overlap(ri[k], intrvl2)):
            out.append(interval2bp(ri[k], ri[k + 1], (rn, k, k + 1), gap) + [mq[k], mq[k + 1]])
            used[k] = 1
    for k in range(1, n - 1):
        if used[k - 1] or used[k] or not (mq[k] < gap_mapq and mq[k - 1] >= min_mapq and mq[k + 1] >= min_mapq):
            continue
        if (interval_overlap(ri[k - 1], intrvl1) and interval_overlap(ri[k + 1], intrvl2)) or \
                (interval_overlap(ri[k + 1], intrvl1) and interval_overlap(ri[k - 1], intrvl2)):
            gap = int(qi[k + 1][0]) - int(qi[k - 1][1])
            out.append(interval2bp(ri[k - 1], ri[k + 1], (rn, k - 1, k + 1), gap) + [mq[k - 1], mq[k + 1]])
    return out


def _discordant_pair(qa, qb, ra, rb, gap_):
    """Shared test of bu:146-161 / bu:174-185 for two segments lying in the same interval."""
    if rb[3] != ra[3]:
        return True
    gr = int(qb[0]) - int(qa[1])
    if rb[3] == "+":
        grr = int(rb[1]) - int(ra[2])
    else:
        grr = int(ra[2]) - int(rb[1])
    return abs(gr - grr) > max(gap_, abs(gr * 0.2))


def alignment2bp_l(rn, ca, min_bp_match_cutoff, min_mapq, gap_, intrvls, gap_mapq=10):
    """bu:129-186 — candidates whose two segments fall in the same amplicon interval."""
    qi, ri, mq = ca[0], ca[1], ca[2]
    n = len(ri)
    out = []
    used = [0] * max(0, n - 1)
    for k in range(n - 1):
        io1 = interval_overlap_l(ri[k], intrvls)
        io2 = interval_overlap_l(ri[k + 1], intrvls)
        gap = int(qi[k + 1][0]) - int(qi[k][1])
        if gap + min_bp_match_cutoff >= 0 and io1 >= 0 and io2 >= 0 and io1 == io2:
            if _discordant_pair(qi[k], qi[k + 1], ri[k], ri[k + 1], gap_) and mq[k] >= min_mapq and mq[k + 1] >= min_mapq:
                out.append(interval2bp(ri[k], ri[k + 1], (rn, k, k + 1), gap) + [mq[k], mq[k + 1]])
                used[k] = 1
    for k in range(1, n - 1):
        io1 = interval_overlap_l(ri[k - 1], intrvls)
        io2 = interval_overlap_l(ri[k + 1], intrvls)
        if used[k - 1] == 0 and used[k] == 0 and mq[k] < gap_mapq and mq[k - 1] >= min_mapq and mq[k + 1] >= min_mapq \
                and io1 >= 0 and io2 >= 0 and io1 == io2:
            if _discordant_pair(qi[k - 1], qi[k + 1], ri[k - 1], ri[k + 1], gap_):
                gap = int(qi[k + 1][0]) - int(qi[k - 1][1])
                out.append(interval2bp(ri[k - 1], ri[k + 1], (rn, k - 1, k + 1), gap) + [mq[k - 1], mq[k + 1]])
    return out


def cluster_bp_list(bp_list, min_cluster_size, bp_distance_cutoff):
    """bu:252-286 — group by (chr1, chr2, o1, o2) in first-seen order, greedy first-fit inside a group."""
    groups = {}
    for i, bp in enumerate(bp_list):
        groups.setdefault((bp[0], bp[3], bp[2], bp[5]), []).append(i)
    clusters = []
    for members in groups.values():
        if len(members) < min_cluster_size:
            clusters.append([bp_list[i] for i in members])
            continue
        local = []
        for i in members:
            bp = bp_list[i]
            home = -1
            for ci, cl in enumerate(local):
                if any(abs(int(bp[1]) - int(o[1])) < bp_distance_cutoff and
                       abs(int(bp[4]) - int(o[4])) < bp_distance_cutoff for o in cl):
                    home = ci
                    break
            if home >= 0:
                local[home].append(bp)
            else:
                local.append([bp])
        clusters += local
    return clusters


def bp_match(bp1, bp2, rgap, cutoff):                    # bu:391-416
    if not (bp1[0] == bp2[0] and bp1[3] == bp2[3] and bp1[2] == bp2[2] and bp1[5] == bp2[5]):
        return False
    d1 = abs(int(bp1[1]) - int(bp2[1])) < cutoff[0]
    d2 = abs(int(bp1[4]) - int(bp2[4])) < cutoff[1]
    if rgap <= 0:
        return d1 and d2
    left = rgap
    used = [False, False]
    for e, (pi, oi, ci) in enumerate(((1, 2, 0), (4, 5, 1))):
        a, b = int(bp1[pi]), int(bp2[pi])
        if bp1[oi] == "+" and a <= b - cutoff[ci]:
            left -= (b - cutoff[ci] - a + 1)
            used[e] = True
        if bp1[oi] == "-" and a >= b + cutoff[ci]:
            left -= (a - b - cutoff[ci] + 1)
            used[e] = True
    return ((used[0] and left >= 0) or d1) and ((used[1] and left >= 0) or d2)


def _consensus_pos(values, last_orientation):
    """Mode if unique, otherwise the median rounded toward the junction (bu:336-357).

    ``last_orientation`` is the orientation of the LAST cluster member (the reference reads its loop
    variable after the loop, bu:343/354 — Appendix A Q6).
    """
    top = Counter(values).most_common(2)
    if len(top) == 1 or top[0][1] > top[1][1]:
        return top[0][0]
    med = np.median(values)
    if len(values) % 2 == 1:
        return int(med)
    return int(math.ceil(med)) if last_orientation == "+" else int(math.floor(med))


def bpc2bp(cluster, cutoff):
    """bu:299-388.  Returns (bp, supporting read tuples, stats, unexplained members)."""
    bp = list(cluster[0][:-2])
    bp[1] = 0 if bp[2] == "+" else 1000000000
    bp[4] = 0 if bp[5] == "+" else 1000000000
    n = len(cluster) * 1.0
    s1 = sum(m[1] for m in cluster); s11 = sum(m[1] * m[1] for m in cluster)
    s4 = sum(m[4] for m in cluster); s44 = sum(m[4] * m[4] for m in cluster)
    mu1, mu4, q1, q4 = s1 / n, s4 / n, s11 / n, s44 / n
    try:
        sd1 = max(cutoff / 2.99, math.sqrt(q1 - mu1 * mu1))
    except ValueError:
        sd1 = cutoff / 2.99
    try:
        sd4 = max(cutoff / 2.99, math.sqrt(q4 - mu4 * mu4))
    except ValueError:
        sd4 = cutoff / 2.99
    keep1, keep4 = [], []
    for m in cluster:
        if mu1 - 3 * sd1 <= m[1] <= mu1 + 3 * sd1 and mu4 - 3 * sd4 <= m[4] <= mu4 + 3 * sd4:
            keep1.append(m[1])
            keep4.append(m[4])
    last = cluster[-1]
    if keep1:
        bp[1] = _consensus_pos(keep1, last[2])
    if keep4:
        bp[4] = _consensus_pos(keep4, last[5])
    support, rest = [], []
    st = [0, 0, 0, 0, 0, 0]
    for m in cluster:
        if bp_match(m, bp, m[7] * 1.2, [cutoff, cutoff]):
            support.append(m[6])
            st[0] += m[1]; st[2] += m[1] * m[1]
            st[1] += m[4]; st[3] += m[4] * m[4]
            if m[-3] == 0:
                st[4] += m[-2]; st[5] += m[-1]
            else:
                st[4] += m[-1]; st[5] += m[-2]
        else:
            rest.append(m)
    if not support:
        return bp, support, [0, 0, 0, 0, 0, 0], []
    k = len(support) * 1.0
    st = [v / k for v in st]
    for j in (2, 3):
        try:
            st[j] = math.sqrt(st[j] - st[j - 2] * st[j - 2])
        except ValueError:
            st[j] = 0
    return bp, support, st, rest


# ======================================================================================
# breakpoint_graph.py — container, CN model, writers
# ======================================================================================
class OracleBreakpointGraph:
    """Field layout follows bg:83-207 (lists of lists, node adjacency in insertion order)."""

    def __init__(self):
        self.amplicon_intervals = []
        self.sequence_edges = []
        self.concordant_edges = []
        self.discordant_edges = []
        self.source_edges = []
        self.nodes = {}
        self.endnodes = {}
        self.max_cn = 0.0

    def add_node(self, node):                       # bg:113-124 (re-adding RESETS the adjacency)
        self.nodes[node] = [[], [], [], []]

    def add_endnode(self, node):                    # bg:127-139
        if node not in self.endnodes:
            self.endnodes[node] = []

    def add_sequence_edge(self, c, l, r):           # bg:167-176
        k = len(self.sequence_edges)
        self.nodes[(c, l, "-")][0].append(k)
        self.nodes[(c, r, "+")][0].append(k)
        self.sequence_edges.append([c, l, r, -1, "d", -1, 0, r - l + 1, 0.0])

    def add_concordant_edge(self, c1, p1, o1, c2, p2, o2):   # bg:179-190
        k = len(self.concordant_edges)
        self.nodes[(c1, p1, o1)][1].append(k)
        self.nodes[(c2, p2, o2)][1].append(k)
        self.concordant_edges.append([c1, p1, o1, c2, p2, o2, -1, "d", -1, set(), 0.0])

    def add_discordant_edge(self, c1, p1, o1, c2, p2, o2, lr_count, reads):   # bg:193-207
        k = len(self.discordant_edges)
        self.nodes[(c1, p1, o1)][2].append(k)
        self.nodes[(c2, p2, o2)][2].append(k)
        if (c1, p1, o1) in self.endnodes:
            self.endnodes[(c1, p1, o1)].append(k)
        if (c2, p2, o2) in self.endnodes:
            self.endnodes[(c2, p2, o2)].append(k)
        self.discordant_edges.append([c1, p1, o1, c2, p2, o2, -1, "d", 0.0, lr_count, reads, 0.0])

    def sort_edges(self):                           # bg:348-363
        self.sequence_edges.sort(key=lambda e: (CHR_IDX[e[0]], e[1]))
        self.concordant_edges.sort(key=lambda e: (CHR_IDX[e[0]], e[1]))
        for k, e in enumerate(self.sequence_edges):
            self.nodes[(e[0], e[1], "-")][0] = [k]
            self.nodes[(e[0], e[2], "+")][0] = [k]
        for k, e in enumerate(self.concordant_edges):
            self.nodes[(e[0], e[1], e[2])][1] = [k]
            self.nodes[(e[3], e[4], e[5])][1] = [k]

    # ---- CN model (bg:495-606) ----
    def cn_problem(self, normal_cov):
        ls, lc, ld, lsrc = (len(self.sequence_edges), len(self.concordant_edges), len(self.discordant_edges),
                            len(self.source_edges))
        lin = [0.5 * normal_cov * e[7] for e in self.sequence_edges] + [normal_cov] * (lc + ld) + [0.5 * normal_cov] * lsrc
        lg = [-0.5] * ls + [e[8] * 1.0 for e in self.concordant_edges] + [e[9] * 1.0 for e in self.discordant_edges] + [-0.5] * lsrc
        inv = [0.5 * e[6] ** 2 / (normal_cov * e[7]) for e in self.sequence_edges] + [0.0] * (lc + ld) + \
              [0.5 * e[-1] ** 2 / normal_cov for e in self.source_edges]
        rows = []
        for node, adj in self.nodes.items():
            if node in self.endnodes:
                continue
            row = np.zeros(ls + lc + ld + lsrc)
            for k in adj[0]:
                row[k] = 1
            for k in adj[1]:
                row[ls + k] = -1
            for k in adj[2]:
                row[ls + lc + k] = -1          # assignment, not accumulation: a self-loop counts once (Q19)
            for k in adj[3]:
                row[ls + lc + ld + k] = -1
            rows.append(row)
        A = np.array(rows) if rows else np.zeros((0, ls + lc + ld + lsrc))
        return np.array(inv), np.array(lin), np.array(lg), A

    def compute_cn_lr(self, normal_cov):
        ls, lc, ld = len(self.sequence_edges), len(self.concordant_edges), len(self.discordant_edges)
        inv, lin, lg, A = self.cn_problem(normal_cov)
        if A.shape[0] > 0:
            x = solve_cn(inv, lin, lg, A)
            for k in range(ls):
                self.sequence_edges[k][-1] = x[k] * 2
                self.max_cn = max(self.max_cn, x[k] * 2)
            for k in range(lc):
                self.concordant_edges[k][-1] = x[ls + k] * 2
                self.max_cn = max(self.max_cn, x[ls + k] * 2)
            for k in range(ld):
                e = self.discordant_edges[k]
                v = x[ls + lc + k]
                if not (e[0] == e[3] and e[1] == e[4] and e[2] == e[5]):   # bg:585-592
                    v = v * 2
                e[-1] = v
                self.max_cn = max(self.max_cn, v)
            for k in range(len(self.source_edges)):
                self.source_edges[k][-1] = x[ls + lc + ld + k] * 2
                self.max_cn = max(self.max_cn, x[ls + lc + ld + k] * 2)
        else:                                                               # bg:597-605
            assert lc == 0 and ld == 0 and len(self.source_edges) == 0
            for e in self.sequence_edges:
                e[-1] = e[6] * 2.0 / (normal_cov * e[7])
                self.max_cn = max(self.max_cn, e[-1])
        self.max_cn += 1.0


# ---- multiplicities handed to the cycle step (bg:17-80, bg:609-693) ----
def _cluster_test(rc, part, max_multiplicity=5):
    """bg:17-71: (valid, index of the last multiplicity-1 entry, score) of the run rc[part[0]..part[1]] (ascending)."""
    a, b = part
    if a == b:                                               # bg:18-19
        return True, a, 0.0
    p = rc[a:b + 1]
    if p[-1] < p[0] * 2.0:                                   # bg:21-22
        return True, b, 0.0
    k0 = 0
    while k0 < len(p) and p[k0] < p[0] * 2.0:                # bg:23-25
        k0 += 1
    if p[-1] / np.average(p[:k0]) >= max_multiplicity + 0.5:  # bg:26-28
        return False, None, None
    score, best, dev_best = -10.0, k0, 1.0
    k = k0
    while k >= 1:                                            # bg:32 (base group shrinks from the right)
        mean = np.average(p[:k])
        m = 2
        if not p[k] / mean < m - 0.5:                        # bg:40-41
            while p[k] / mean >= m + 0.5:                    # bg:42-43
                m += 1
            gap = math.log2(p[k]) - math.log2(p[k - 1])      # bg:44
            spans = {}
            left = k
            for i in range(k, len(p)):                       # bg:47-53
                if p[i] / mean >= m + 0.5:
                    gap += math.log2(p[i]) - math.log2(p[i - 1])
                    spans[m] = [left, i - 1]
                    left = i
                    while p[i] / mean >= m + 0.5:
                        m += 1
            spans[m] = [left, len(p) - 1]                    # bg:54
            fits = m <= max_multiplicity                     # bg:55-56
            for mm in range(2, m + 1):                       # bg:57-63
                if mm in spans and spans[mm][1] - spans[mm][0] >= k:
                    fits = False
            if fits:
                dev = sum([abs(mm - np.average(p[spans[mm][0]: spans[mm][1] + 1] / mean))
                           for mm in range(2, m + 1) if mm in spans])       # bg:64
                if gap - dev > score:                        # bg:65-68
                    score, dev_best, best = gap - dev, dev, k
        k -= 1
    if dev_best < 1.0:                                       # bg:69-72
        return True, best + a - 1, score
    return False, None, None


def _contiguous_partitions(k, start, end):
    """bg:74-80: every split of start..end into k + 1 non-empty contiguous parts, first part shortest first."""
    if k == 0:
        yield [[start, end]]
        return
    for first_len in range(1, end - start - k + 2):
        for rest in _contiguous_partitions(k - 1, start + first_len, end):
            yield [[start, start + first_len - 1]] + rest


def infer_discordant_edge_multiplicities(discordant_edges, max_multiplicity=5):
    """bg:630-693 on the long-read supports ``e[9]`` of the discordant edges."""
    rc = [e[9] for e in discordant_edges]
    if not rc:
        return []
    idx = np.argsort(rc)                                      # bg:640
    rc = sorted(rc)
    if math.log2(rc[-1]) - math.log2(rc[0]) < 1.0:            # bg:642-643
        return [1 for _ in idx]
    n_parts, chosen, chosen_bases, found = 1, [], [], False
    while not found:                                          # bg:651-677: fewest parts with a valid split
        top = -10.0
        for parts in _contiguous_partitions(n_parts - 1, 0, len(rc) - 1):
            total, bases, ok = 0.0, [], True
            for pi, part in enumerate(parts):
                valid, base_end, sc = _cluster_test(rc, part, max_multiplicity)
                if not valid:
                    ok = False
                    break
                total += sc
                bases.append([part[0], base_end])
                if pi > 0:                                    # bg:667-668
                    total += math.log2(rc[part[0]]) - math.log2(rc[parts[pi - 1][1]])
            if ok:
                found = True
                if total > top:
                    top, chosen, chosen_bases = total, parts, bases
        n_parts += 1
    out_sorted = []
    for part, (b0, b1) in zip(chosen, chosen_bases):          # bg:678-692
        out_sorted += [1] * (b1 - b0 + 1)
        nxt = b1 + 1
        if nxt > part[1]:
            continue
        mean = np.average(rc[b0: b1 + 1])
        m = 2
        while rc[nxt] / mean >= m + 0.5:
            m += 1
        for i in range(nxt, part[1] + 1):
            while rc[i] / mean >= m + 0.5:
                m += 1
            out_sorted.append(m)
    where = list(idx)
    return [out_sorted[where.index(i)] for i in range(len(rc))]            # bg:693


def infer_max_seq_multiplicity(sequence_edges, gain=5.0, size_cutoff=10000, multiplicity=2):
    """bg:609-627."""
    cns = [e[-1] for e in sequence_edges if e[7] >= size_cutoff and e[-1] >= gain]
    lens = [e[7] for e in sequence_edges if e[7] >= size_cutoff and e[-1] >= gain]
    if cns:
        return int(round(max(cns) / np.average(cns, weights=lens))) + 1
    return multiplicity


def solve_cn(inv, lin, lg, A, max_iter=200):
    """minimise Σ inv/x + lin·x − lg·log x  s.t.  A x = 0, x > 0  (objective bg:546-556), from x = 1.

    Infeasible-start Newton on the full KKT system (the Hessian is diagonal and may have zero entries —
    concordant edges without read support — so no Schur complement; A may be rank deficient, in which case
    the step comes from least squares), float64, iterated until the Newton step is below 1e-13 relative.  This is NOT cvxopt's
    algorithm; it returns the optimum cvxopt.solvers.cp approximates.
    """
    n, p = len(lin), A.shape[0]
    x = np.ones(n)
    nu = np.zeros(p)

    def resid(x, nu):
        g = lin - lg / x - inv / (x * x)
        return np.concatenate([g + A.T @ nu, A @ x])
    K = np.zeros((n + p, n + p))
    K[:n, n:] = A.T
    K[n:, :n] = A
    r = resid(x, nu)
    for _ in range(max_iter):
        K[np.arange(n), np.arange(n)] = lg / (x * x) + 2.0 * inv / (x ** 3)
        # Exact LU solve when the KKT matrix is regular; least squares only when A is rank deficient.  (lstsq alone truncates
        # small singular values of this badly scaled matrix — entries span 1e-4 .. 1e10 at 2 M reads — and the inexact Newton
        # direction then stalls the line search at an infeasible point: seen at full config-3 size, tools/validate_full_size.py.)
        try:
            step = np.linalg.solve(K, -r)
            if not np.all(np.isfinite(step)) or np.linalg.norm(K @ step + r) > 1e-8 * max(1.0, np.linalg.norm(r)):
                raise np.linalg.LinAlgError            # (numerically) singular: dependent balance rows
        except np.linalg.LinAlgError:
            step = np.linalg.lstsq(K, -r, rcond=None)[0]
        dx, dnu = step[:n], step[n:]
        t = 1.0
        neg = dx < 0
        if neg.any():
            t = min(1.0, 0.99 * float(np.min(-x[neg] / dx[neg])))
        r0 = np.linalg.norm(r)
        ok = False
        while t > 1e-10:
            rn_ = resid(x + t * dx, nu + t * dnu)
            if np.linalg.norm(rn_) <= (1 - 0.01 * t) * r0:
                ok = True
                break
            t *= 0.5
        tiny = float(np.max(np.abs(dx) / x)) < 1e-13
        if not ok:
            if tiny:
                x = x + dx
            break
        x, nu, r = x + t * dx, nu + t * dnu, rn_
        if tiny:
            break
    return x


def graph_text(g):
    """bg:805-822 — the *_graph.txt text."""
    out = ["SequenceEdge: StartPosition, EndPosition, PredictedCN, AverageCoverage, Size, NumberOfLongReads\n"]
    for e in g.sequence_edges:
        out.append("sequence\t%s:%s-\t%s:%s+\t%f\t%f\t%d\t%d\n" % (e[0], e[1], e[0], e[2], e[-1], e[6] * 1.0 / e[7], e[7], e[5]))
    out.append("BreakpointEdge: StartPosition->EndPosition, PredictedCN, NumberOfLongReads\n")
    for e in g.source_edges:
        out.append("source\t%s:%s%s->%s:%s%s\t%f\t-1\n" % (e[0], e[1], e[2], e[3], e[4], e[5], e[-1]))
    for e in g.concordant_edges:
        out.append("concordant\t%s:%s%s->%s:%s%s\t%f\t%d\n" % (e[0], e[1], e[2], e[3], e[4], e[5], e[-1], e[8]))
    for e in g.discordant_edges:
        out.append("discordant\t%s:%s%s->%s:%s%s\t%f\t%d\n" % (e[0], e[1], e[2], e[3], e[4], e[5], e[-1], e[9]))
    return "".join(out)


def breakpoint_info_text(g, bp_stats):
    """bg:845-854 — the *_breakpoints.txt text."""
    out = ["chr1\tpos1\tchr2\tpos2\torientation\tlr_support\tlr_info=[avg1, avg2, std1, std2, mapq1, mapq2]\n"]
    for k, e in enumerate(g.discordant_edges):
        out.append("%s\t%s\t%s\t%s\t%s%s\t%d\t%s\n" % (e[3], e[4], e[0], e[1], e[5], e[2], e[9], bp_stats[k]))
    return "".join(out)


# ======================================================================================
# infer_breakpoint_graph.py — the graph builder
# ======================================================================================
class OracleGraphBuild:
    """Restatement of ``bam_to_breakpoint_nanopore`` (ibg:20-1056) on HostRecords.

    Per-instance state (the reference's class-level mutables, ibg:22-61, are a process-wide singleton).
    """
    max_seq_len = 2000000
    cn_gain = 5.0
    min_bp_match_cutoff_ = 100
    interval_delta = 100000
    max_breakpoint_distance_cutoff = 2000
    min_del_len = 600

    def __init__(self, records, seedfile):                       # ibg:64-72
        self.rec = records
        self.min_bp_cov_factor = 1.0
        self.min_cluster_cutoff = 3
        self.read_length = {}
        self.chimeric_alignments = {}
        self.chimeric_alignments_seg = {}
        self.large_indel_alignments = {}
        self.nm_stats = [0.0, 0.0, 0]
        self.amplicon_intervals = []
        self.amplicon_interval_connections = {}
        self.cns_intervals = []
        self.cns_intervals_by_chr = {}
        self.log2_cn = []
        self.cns_index = {}          # chr -> (starts, ends_exclusive, idx) arrays: stands in for the IntervalTree
        self.normal_cov = 0.0
        self.ccid2id = {}
        self.new_bp_list = []
        self.new_bp_stats = []
        self.new_bp_ccids = []
        self.source_edges = []
        self.lr_graph = []
        with open(seedfile) as fp:
            for line in fp:
                s = line.strip().split()
                self.amplicon_intervals.append([s[0], int(s[1]), int(s[2]), -1])

    # ---- A1/A2: read_cns (ibg:75-136)
    def read_cns(self, cns):
        self.cns_intervals, self.log2_cn = [], []
        tree = {}
        idx = 0
        with open(cns) as fp:
            for line in fp:
                s = line.strip().split()
                if s[0] == "chromosome":
                    continue
                self.cns_intervals.append([s[0], int(s[1]), int(s[2]) - 1])
                if s[0] not in tree:
                    tree[s[0]] = []
                    self.cns_intervals_by_chr[s[0]] = []
                    idx = 0                                  # resets only on first sight of a chromosome (Q8)
                tree[s[0]].append((int(s[1]), int(s[2]), idx))
                idx += 1
                if cns.endswith(".cns"):
                    self.cns_intervals_by_chr[s[0]].append([s[0], int(s[1]), int(s[2]) - 1, 2 * (2 ** float(s[4]))])
                    self.log2_cn.append(float(s[4]))
                elif cns.endswith(".bed"):
                    self.cns_intervals_by_chr[s[0]].append([s[0], int(s[1]), int(s[2]) - 1, float(s[3])])
                    self.log2_cn.append(np.log2(float(s[3]) / 2.0))
        self.cns_index = {c: tuple(np.array(col) for col in zip(*v)) for c, v in tree.items()}
        order = np.argsort(self.log2_cn)
        im = int(len(order) / 2.4)
        ip = im + 1
        picked = [self.cns_intervals[order[ip]], self.cns_intervals[order[im]]]
        total = sum(p[2] - p[1] + 1 for p in picked)
        i = 1
        while total < 10000000:                               # IndexError when the file is too short (ibg:118-124)
            for p in (self.cns_intervals[order[ip + i]], self.cns_intervals[order[im - i]]):
                picked.append(p)
                total += p[2] - p[1] + 1
            i += 1
        nnc = 0
        for p in picked:
            nnc += self.rec.count_coverage_sum(p[0], p[1], p[2] + 1)
        self.normal_cov = nnc * 1.0 / total
        self.min_cluster_cutoff = max(self.min_cluster_cutoff, self.min_bp_cov_factor * self.normal_cov)

    def pos2cni(self, chrom, pos):                            # ibg:177-178 (IntervalTree point query, half-open)
        st, en, ix = self.cns_index[chrom]
        return ix[(st <= pos) & (pos < en)].tolist()

    # ---- A3: fetch (ibg:139-174)
    def fetch(self):
        r = self.rec
        for i in range(r.n):
            if r.tid[i] < 0:
                continue
            rn = r.names[r.name_id[i]]
            if r.flag[i] < 256 and rn not in self.read_length:
                self.read_length[rn] = int(r.qlen[i])
            sa = r.sa_str[i]
            if sa is not None:
                lst = self.chimeric_alignments.setdefault(rn, [])
                for ent in sa[:-1].split(";"):
                    if ent not in lst:
                        lst.append(ent)
            elif r.mapq[i] == 60:
                e = int(r.nm[i]) / int(r.qlen[i])
                self.nm_stats[0] += e
                self.nm_stats[1] += e * e
                self.nm_stats[2] += 1
        self.nm_stats[0] /= self.nm_stats[2]
        self.nm_stats[1] = math.sqrt(self.nm_stats[1] / self.nm_stats[2] - self.nm_stats[0] ** 2)
        orphans = []
        for rn in self.chimeric_alignments:
            if rn not in self.read_length:
                orphans.append(rn)
                continue
            self.chimeric_alignments[rn] = alignment_from_satags(self.chimeric_alignments[rn], self.read_length[rn])
        for rn in orphans:
            del self.chimeric_alignments[rn]

    # ---- A4: hash_alignment_to_seg (ibg:181-210)
    def hash_alignment_to_seg(self):
        for rn, ca in self.chimeric_alignments.items():
            for seg in ca[1]:
                if seg[0] not in self.cns_index:
                    seg.append(set([-1]))
                    continue
                lo = self.pos2cni(seg[0], min(seg[1], seg[2]))
                hi = self.pos2cni(seg[0], max(seg[1], seg[2]))
                assert len(lo) <= 1 and len(hi) <= 1
                cniset = set([lo[0] if lo else -1, hi[0] if hi else -1])
                if len(cniset) > 1 and -1 in cniset:
                    cniset.remove(-1)
                seg.append(cniset)
                per_chr = self.chimeric_alignments_seg.setdefault(seg[0], {})
                for cni in cniset:
                    if cni != -1:
                        per_chr.setdefault(cni, []).append(rn)

    # ---- A5: find_amplicon_intervals (ibg:213-323)
    def find_amplicon_intervals(self):
        by = self.cns_intervals_by_chr
        for iv in self.amplicon_intervals:                    # snap seeds to CN-segment bounds ± delta (ibg:216-225)
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
        n0 = len(self.amplicon_intervals)                     # range(len()) is evaluated once in the reference
        for ai in range(n0):
            if self.amplicon_intervals[ai][3] == -1:
                self.find_interval_i(ai, ccid)
                ccid += 1
        ivs = self.amplicon_intervals
        order = sorted(range(len(ivs)), key=lambda i: (CHR_IDX[ivs[i][0]], ivs[i][1]))
        srt = [ivs[i] for i in order]
        # runs of adjacent / overlapping intervals (ibg:243-254)
        runs = []
        first = 0
        for k in range(len(srt) - 1):
            if not (interval_adjacent(srt[k + 1], srt[k]) or interval_overlap(srt[k], srt[k + 1])):
                if k > first:
                    runs.append([first, k])
                first = k + 1
        if len(srt) > 0 and first < len(srt) - 1:
            runs.append([first, len(srt) - 1])
        conn = self.amplicon_interval_connections
        for a, b in runs[::-1]:                               # ibg:255-298
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
                    if gone == cmap[key][0]:
                        cmap[key] = (keep, cmap[key][1])
                    if gone == cmap[key][1]:
                        cmap[key] = (cmap[key][0], keep)
                    if cmap[key][1] < cmap[key][0]:
                        cmap[key] = (cmap[key][1], cmap[key][0])
            for key in cmap:
                new = cmap[key]
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
        ind = {order[i]: i for i in range(len(order))}
        cmap = {key: (min(ind[key[0]], ind[key[1]]), max(ind[key[0]], ind[key[1]])) for key in conn}
        self.amplicon_interval_connections = {cmap[key]: conn[key] for key in conn}
        # re-label connected components by BFS over the connections (ibg:305-319)
        seen = [0] * len(self.amplicon_intervals)
        for ai in range(len(self.amplicon_intervals)):
            label = self.amplicon_intervals[ai][3]
            if seen[ai]:
                continue
            queue = [ai]
            while queue:
                cur = queue.pop(0)
                seen[cur] = 1
                if self.amplicon_intervals[cur][3] != label:
                    self.amplicon_intervals[cur][3] = label
                for (p, q) in self.amplicon_interval_connections:
                    if p == cur and not seen[q]:
                        queue.append(q)
                    elif q == cur and not seen[p]:
                        queue.append(p)

    def addbp(self, bp_, bpr_, bp_stats_, ccid):              # ibg:326-340
        for k, bp in enumerate(self.new_bp_list):
            if bp[0] == bp_[0] and bp[3] == bp_[3] and bp[2] == bp_[2] and bp[5] == bp_[5] and \
                    abs(bp[1] - bp_[1]) < 200 and abs(bp[4] - bp_[4]) < 200:
                bp[-1] |= set(bpr_)
                return k
        self.new_bp_list.append(bp_ + [bpr_])
        self.new_bp_ccids.append(ccid)
        self.new_bp_stats.append(bp_stats_)
        return len(self.new_bp_list) - 1

    def _accept(self, num_subcluster, support):
        """Support test shared by ibg:450-451, ibg:704, ibg:788."""
        n = len(set(support))
        return (num_subcluster == 0 and n >= self.min_cluster_cutoff) or \
            n >= max(self.normal_cov * self.min_bp_cov_factor, 3.0)

    def find_interval_i(self, ai, ccid):                      # ibg:343-673
        by = self.cns_intervals_by_chr
        half = int(self.max_seq_len / 2)
        queue = [ai]
        while queue:
            cur = queue.pop(0)
            chrom, s, e = self.amplicon_intervals[cur][:3]
            if self.amplicon_intervals[cur][3] == -1:
                self.amplicon_intervals[cur][3] = ccid
            try:
                si = self.pos2cni(chrom, s)[0]
                ei = self.pos2cni(chrom, e)[0]
            except Exception:
                continue
            # CN segments reached from this interval through a chimeric read (ibg:369-384)
            reach = {}
            for i in range(si, ei + 1):
                if i in self.chimeric_alignments_seg[chrom]:
                    for rn in self.chimeric_alignments_seg[chrom][i]:
                        for seg in self.chimeric_alignments[rn][1]:
                            for j in seg[-1]:
                                if (seg[0] != chrom or (j <= si or j >= ei)) and j != -1:        # Q9
                                    # (the reference's try/except only creates entries for j != -1)
                                    if seg[0] in reach and j in reach[seg[0]]:
                                        reach[seg[0]][j].add(rn)
                                    else:
                                        if seg[0] not in reach:
                                            reach[seg[0]] = {}
                                        reach[seg[0]][j] = set([rn])
            for c in list(reach):                             # ibg:386-398
                for j in [j for j in reach[c] if len(reach[c][j]) < self.min_cluster_cutoff]:
                    del reach[c][j]
                if not reach[c]:
                    del reach[c]

            refined, refined_bps = [], []
            for c in reach:
                bins = sorted(reach[c])
                groups = []                                   # ibg:404-419
                names = set()
                first = 0
                for k in range(len(bins) - 1):
                    nil = by[c][bins[k + 1]][1]
                    lir = by[c][bins[k]][2]
                    names |= reach[c][bins[k]]
                    if bins[k + 1] - bins[k] > 2 or nil - lir > self.max_seq_len:
                        groups.append([c, bins[first], bins[k], names])
                        first = k + 1
                        names = set()
                names |= reach[c][bins[-1]]
                groups.append([c, bins[first], bins[-1], names])

                for grp in groups:                            # ibg:422-623
                    ns = by[grp[0]][grp[1]][1]
                    ne = by[grp[0]][grp[2]][2]
                    cands = []
                    for rn in grp[3]:                         # set-of-str iteration order (Q21)
                        cands += alignment2bp(rn, self.chimeric_alignments[rn], self.min_bp_match_cutoff_, 20,
                                              [grp[0], ns, ne], self.amplicon_intervals[cur])
                    found = []
                    for cl in cluster_bp_list(cands, self.min_cluster_cutoff, self.max_breakpoint_distance_cutoff):
                        if len(cl) < self.min_cluster_cutoff:
                            continue
                        rest = cl
                        while len(rest) >= self.min_cluster_cutoff:
                            bp, bpr, st, rest = bpc2bp(rest, self.min_bp_match_cutoff_)
                            if self._accept(0, bpr):          # num_subcluster never advances here (Q4)
                                k = self.addbp(bp, set(bpr), st, ccid)
                                if k not in found:
                                    found.append(k)
                    inside, outside = [], []
                    if found:
                        tgt = [grp[0], ns, ne]
                        for k in found:
                            bp = self.new_bp_list[k][:6]
                            try:                              # ibg:466-491 (bare except swallows failed point queries)
                                e1 = [bp[0], bp[1], bp[1]]
                                e2 = [bp[3], bp[4], bp[4]]
                                if interval_overlap(e1, self.amplicon_intervals[cur]) and interval_overlap(e2, tgt):
                                    inside.append([self.pos2cni(bp[3], bp[4])[0], bp[4], k])
                                elif interval_overlap(e2, self.amplicon_intervals[cur]) and interval_overlap(e1, tgt):
                                    inside.append([self.pos2cni(bp[0], bp[1])[0], bp[1], k])
                                else:
                                    o1 = interval_overlap(e1, tgt)
                                    o2 = interval_overlap(e2, tgt)
                                    if o1 and o2:
                                        inside.append([self.pos2cni(bp[0], bp[1])[0], bp[1], k])
                                        inside.append([self.pos2cni(bp[3], bp[4])[0], bp[4], k])
                                    elif o1:
                                        inside.append([self.pos2cni(bp[0], bp[1])[0], bp[1], k])
                                        outside.append([bp[3], self.pos2cni(bp[3], bp[4])[0], bp[4], k])
                                    elif o2:
                                        outside.append([bp[0], self.pos2cni(bp[0], bp[1])[0], bp[1], k])
                                        inside.append([self.pos2cni(bp[3], bp[4])[0], bp[4], k])
                                    else:
                                        outside.append([bp[0], self.pos2cni(bp[0], bp[1])[0], bp[1], k])
                                        outside.append([bp[3], self.pos2cni(bp[3], bp[4])[0], bp[4], k])
                            except Exception:
                                pass
                        inside.sort(key=lambda t: (t[0], t[1]))
                        outside.sort(key=lambda t: (CHR_IDX[t[0]], t[1], t[2]))
                        D = self.interval_delta
                        segs = by[c]
                        # --- breakpoints inside the target group (ibg:494-561)
                        first = 0
                        for k in range(len(inside) - 1):
                            nil, ncn = segs[inside[k + 1][0]][1], segs[inside[k + 1][0]][3]
                            lir, lcn = segs[inside[k][0]][2], segs[inside[k][0]][3]
                            amp = ncn >= self.cn_gain or lcn >= self.cn_gain
                            if inside[k + 1][0] - inside[k][0] > 2 or nil - lir > self.max_seq_len / 2 or \
                                    inside[k + 1][1] - inside[k][1] > self.max_seq_len or \
                                    (not amp and nil - lir > 2 * D) or \
                                    (not amp and inside[k + 1][1] - inside[k][1] > 3 * D):
                                if not segs[inside[first][0]][3] >= self.cn_gain:
                                    l = max(inside[first][1] - D, segs[0][1])
                                else:
                                    l = max(segs[inside[first][0]][1] - D, segs[0][1])
                                if not segs[inside[k][0]][3] >= self.cn_gain:
                                    r = min(inside[k][1] + D, segs[-1][2])
                                else:
                                    r = min(lir + D, segs[-1][2])
                                if segs[inside[first][0]][3] and inside[first][1] - half > l:     # Q3: truthiness
                                    l = inside[first][1] - half
                                if inside[k][1] + half < r:
                                    r = inside[k][1] + half
                                if not self.pos2cni(c, l):
                                    l = segs[inside[first][0]][1]
                                if not self.pos2cni(c, r):
                                    r = lir
                                refined.append([c, l, r, -1])
                                refined_bps.append([inside[j][2] for j in range(first, k + 1)])
                                first = k + 1
                        if inside:
                            if not segs[inside[first][0]][3] >= self.cn_gain:
                                l = max(inside[first][1] - D, segs[0][1])
                            else:
                                l = max(segs[inside[first][0]][1] - D, segs[0][1])
                            if not segs[inside[-1][0]][3] >= self.cn_gain:
                                r = min(inside[-1][1] + D, segs[-1][2])
                            else:
                                r = min(segs[inside[-1][0]][2] + D, segs[-1][2])
                            if inside[first][1] - half > l:
                                l = inside[first][1] - half > l                                      # Q2: stores a bool
                            if inside[-1][1] + half < r:
                                r = inside[-1][1] + half
                            if not self.pos2cni(c, l):
                                l = segs[inside[first][0]][1]
                            if not self.pos2cni(c, r):
                                r = segs[inside[-1][0]][2]
                            refined.append([c, l, r, -1])
                            refined_bps.append([inside[j][2] for j in range(first, len(inside))])
                        # --- breakpoint ends outside both intervals (ibg:562-623)
                        first = 0
                        for k in range(len(outside) - 1):
                            a, b = outside[k], outside[k + 1]
                            nil, ncn = by[b[0]][b[1]][1], by[b[0]][b[1]][3]
                            lir, lcn = by[a[0]][a[1]][2], by[a[0]][a[1]][3]
                            amp = ncn >= self.cn_gain or lcn >= self.cn_gain
                            if b[0] != a[0] or b[1] - a[1] > 2 or nil - lir > self.max_seq_len / 2 or \
                                    b[2] - a[2] > self.max_seq_len or (not amp and nil - lir > 2 * D) or \
                                    (not amp and b[2] - a[2] > 3 * D):
                                f = outside[first]
                                if not by[f[0]][f[1]][3] >= self.cn_gain:
                                    l = max(f[2] - D, by[f[0]][0][1])
                                else:
                                    l = max(by[f[0]][f[1]][1] - D, by[f[0]][0][1])
                                if not by[a[0]][a[1]][3] >= self.cn_gain:
                                    r = min(a[2] + D, by[a[0]][-1][2])
                                else:
                                    r = min(lir + D, by[a[0]][-1][2])
                                if f[2] - half > l:
                                    l = f[2] - half
                                if a[2] + half < r:
                                    r = a[2] + half
                                if not self.pos2cni(f[0], l):
                                    l = by[f[0]][f[1]][1]
                                if not self.pos2cni(a[0], r):
                                    r = lir
                                refined.append([f[0], l, r, -1])
                                refined_bps.append([])
                                first = k + 1
                        if outside:
                            f, z = outside[first], outside[-1]
                            if not by[f[0]][f[1]][3] >= self.cn_gain:
                                l = max(f[2] - D, by[f[0]][0][1])
                            else:
                                l = max(by[f[0]][f[1]][1] - D, by[f[0]][0][1])
                            if not by[z[0]][z[1]][3] >= self.cn_gain:
                                r = min(z[2] + D, by[z[0]][-1][2])
                            else:
                                r = min(by[z[0]][z[1]][2] + D, by[z[0]][-1][2])
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
            for ni in range(len(refined)):                    # ibg:627-673
                hit, parts = interval_exclusive(refined[ni], self.amplicon_intervals)
                if not parts:
                    for k in refined_bps[ni]:
                        bp = self.new_bp_list[k][:6]
                        for o in hit:
                            key = (min(cur, o), max(cur, o))
                            if (o != cur and interval_overlap([bp[0], bp[1], bp[1]], self.amplicon_intervals[o])) or \
                                    interval_overlap([bp[3], bp[4], bp[4]], self.amplicon_intervals[o]):   # Q5
                                conn.setdefault(key, set()).add(k)
                    for o in hit:
                        if o != cur and self.amplicon_intervals[o][3] < 0:
                            queue.append(o)
                else:
                    for part in parts:
                        nai = len(self.amplicon_intervals)
                        self.amplicon_intervals.append(part)
                        conn[(cur, nai)] = set()
                        if not hit:
                            for k in refined_bps[ni]:
                                conn[(cur, nai)].add(k)
                        else:
                            for k in refined_bps[ni]:
                                bp = self.new_bp_list[k][:6]
                                for o in hit:
                                    key = (min(cur, o), max(cur, o))
                                    if interval_overlap([bp[0], bp[1], bp[1]], self.amplicon_intervals[o]) or \
                                            interval_overlap([bp[3], bp[4], bp[4]], self.amplicon_intervals[o]):
                                        conn.setdefault(key, set()).add(k)
                                    else:
                                        conn[(cur, nai)].add(k)
                        queue.append(nai)

    # ---- shared tail of find_breakpoints / find_smalldel_breakpoints (ibg:691-718, ibg:775-802)
    def _cluster_and_add(self, cands):
        for cl in cluster_bp_list(cands, self.min_cluster_cutoff, self.max_breakpoint_distance_cutoff):
            if len(cl) < self.min_cluster_cutoff:
                continue
            sub = 0
            rest = cl
            while len(rest) >= self.min_cluster_cutoff:
                bp, bpr, st, rest = bpc2bp(rest, self.min_bp_match_cutoff_)
                if self._accept(sub, bpr):
                    io1 = interval_overlap_l([bp[0], bp[1], bp[1]], self.amplicon_intervals)
                    io2 = interval_overlap_l([bp[3], bp[4], bp[4]], self.amplicon_intervals)
                    if io1 >= 0 and io2 >= 0:
                        assert self.amplicon_intervals[io1][3] == self.amplicon_intervals[io2][3]
                        k = self.addbp(bp, set(bpr), st, self.amplicon_intervals[io1][3])
                        self.amplicon_interval_connections.setdefault((min(io1, io2), max(io1, io2)), set()).add(k)
                sub += 1

    # ---- A6: find_smalldel_breakpoints (ibg:721-802, live branch 749-762)
    def find_smalldel_breakpoints(self):
        r = self.rec
        for iv in self.amplicon_intervals:
            for i in r.region(iv[0], iv[1], iv[2] + 1):
                if r.mapq[i] < 20:
                    continue
                rn = r.names[r.name_id[i]]
                bl = r.blocks(i)
                for k in range(len(bl) - 1):
                    if abs(bl[k + 1][0] - bl[k][1]) > self.min_del_len:
                        self.large_indel_alignments.setdefault(rn, []).append(
                            [iv[0], bl[k + 1][0], bl[k][1], bl[0][0], bl[-1][1], int(r.mapq[i])])
        cands = []
        for rn, gaps in self.large_indel_alignments.items():
            for k, gp in enumerate(gaps):
                a, b = gp[1], gp[2]
                if b > a:                                      # aliasing "swap" sets both to the same value (Q7)
                    b = a
                cands.append([gp[0], a, "-", gp[0], b, "+", (rn, k, k), 0, 0, -1, -1])
        self._cluster_and_add(cands)

    # ---- A7: find_breakpoints (ibg:676-718)
    def find_breakpoints(self):
        cands = []
        for rn, ca in self.chimeric_alignments.items():
            cands += alignment2bp_l(rn, ca, self.min_bp_match_cutoff_, 20, 100, self.amplicon_intervals)
        self._cluster_and_add(cands)

    # ---- A9: build_graph (ibg:864-1016)
    def build_graph(self):
        cuts = {}
        for k, bp in enumerate(self.new_bp_list):
            for ai, seg in enumerate(self.amplicon_intervals):
                for (ci, pi, oi) in ((0, 1, 2), (3, 4, 5)):
                    if bp[ci] == seg[0] and seg[1] < bp[pi] < seg[2]:        # strictly inside (Q13)
                        if bp[oi] == "+":
                            cuts.setdefault(ai, []).append((bp[pi], bp[pi] + 1, k, pi, "+"))
                        if bp[oi] == "-":
                            cuts.setdefault(ai, []).append((bp[pi] - 1, bp[pi], k, pi, "-"))
        nxt = 1
        for seg in self.amplicon_intervals:                   # ibg:918-922 (Q18)
            if seg[3] not in self.ccid2id:
                self.ccid2id[seg[3]] = nxt
                nxt += 1
        for _ in range(len(self.ccid2id)):
            self.lr_graph.append(OracleBreakpointGraph())
        for ai in cuts:
            cuts[ai].sort(key=lambda t: t[0])
            seg = self.amplicon_intervals[ai]
            g = self.lr_graph[self.ccid2id[seg[3]] - 1]
            c = seg[0]
            for j, cut in enumerate(cuts[ai]):
                if j == 0:
                    left = seg[1]
                elif cut[0] > cuts[ai][j - 1][0]:
                    left = cuts[ai][j - 1][1]
                else:
                    continue
                g.add_node((c, left, "-"))
                g.add_node((c, cut[0], "+"))
                g.add_node((c, cut[1], "-"))
                g.add_sequence_edge(c, left, cut[0])
                g.add_concordant_edge(c, cut[0], "+", c, cut[1], "-")
            g.add_node((c, cuts[ai][-1][1], "-"))
            g.add_node((c, seg[2], "+"))
            g.add_sequence_edge(c, cuts[ai][-1][1], seg[2])
        for ai, seg in enumerate(self.amplicon_intervals):
            if ai not in cuts:
                g = self.lr_graph[self.ccid2id[seg[3]] - 1]
                g.add_node((seg[0], seg[1], "-"))
                g.add_node((seg[0], seg[2], "+"))
                g.add_sequence_edge(seg[0], seg[1], seg[2])
        for g in self.lr_graph:
            g.sort_edges()
        for seg in self.amplicon_intervals:
            g = self.lr_graph[self.ccid2id[seg[3]] - 1]
            g.amplicon_intervals.append([seg[0], seg[1], seg[2]])
            g.add_endnode((seg[0], seg[1], "-"))
            g.add_endnode((seg[0], seg[2], "+"))
        for k, bp in enumerate(self.new_bp_list):
            io1 = interval_overlap_l([bp[0], bp[1], bp[1]], self.amplicon_intervals)
            io2 = interval_overlap_l([bp[3], bp[4], bp[4]], self.amplicon_intervals)
            assert self.amplicon_intervals[io1][3] == self.amplicon_intervals[io2][3]
            cc = self.amplicon_intervals[io1][3]
            if cc != self.new_bp_ccids[k]:
                self.new_bp_ccids[k] = cc
            self.lr_graph[self.ccid2id[cc] - 1].add_discordant_edge(bp[0], bp[1], bp[2], bp[3], bp[4], bp[5],
                                                                     lr_count=len(bp[-1]), reads=bp[-1])

    # ---- A10: assign_cov (ibg:1019-1056)
    def assign_cov(self):
        r = self.rec
        for g in self.lr_graph:
            for e in g.sequence_edges:
                if e[5] == -1:
                    e[5] = sum(1 for i in r.region(e[0], e[1], e[2] + 1) if r.infer_read_length(i))
                    e[6] = r.count_coverage_sum(e[0], e[1], e[2] + 1)
        cut = self.min_bp_match_cutoff_
        names = lambda c, p: set(r.names[r.name_id[i]] for i in r.region(c, p, p + 1))
        for g in self.lr_graph:
            for e in g.concordant_edges:
                rls = names(e[0], e[1])
                rrs = names(e[3], e[4])
                rls1 = names(e[0], e[1] - cut - 1)
                rrs1 = names(e[3], e[4] + cut)
                rbps = set()
                for node in ((e[0], e[1], e[2]), (e[3], e[4], e[5])):
                    for k in g.nodes[node][2]:
                        for t in g.discordant_edges[k][10]:
                            rbps.add(t[0])
                e[9] = rls | rrs
                e[8] = len((rls & rrs & rls1 & rrs1) - rbps)


def reconstruct_graph(records, seedfile, cn_seg, output_prefix=None, min_bp_support=1.0, output_bp=False):
    """ibg:1333-1395 without logging.  Returns (builder, {file name: text}); writes files when a prefix is given."""
    b = OracleGraphBuild(records, seedfile)
    b.min_bp_cov_factor = min_bp_support
    b.read_cns(cn_seg)
    b.fetch()
    b.hash_alignment_to_seg()
    b.find_amplicon_intervals()
    b.find_smalldel_breakpoints()
    b.find_breakpoints()
    b.build_graph()
    files = {}
    if output_bp:
        for gi, g in enumerate(b.lr_graph):
            stats = []
            for e in g.discordant_edges:
                for k, bp in enumerate(b.new_bp_list):
                    if e[:6] == bp[:6]:
                        stats.append(b.new_bp_stats[k])
                        break
            files["_amplicon%d_breakpoints.txt" % (gi + 1)] = breakpoint_info_text(g, stats)
    else:
        b.assign_cov()
        for g in b.lr_graph:
            g.compute_cn_lr(b.normal_cov)
        for gi, g in enumerate(b.lr_graph):
            files["_amplicon%d_graph.txt" % (gi + 1)] = graph_text(g)
    if output_prefix is not None:
        for k, v in files.items():
            with open(output_prefix + k, "w") as fp:
                fp.write(v)
    return b, files
