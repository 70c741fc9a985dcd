"""Subpath constraints: reads -> walks in the breakpoint graph.

The step right after the graph build: ``compute_path_constraints`` (/root/reference/src/infer_breakpoint_graph.py:1059-1323)
and the helpers it uses from /root/reference/src/path_constraints.py (``pc``).  The cycle decomposition calls it on the
object the graph build returns (cycle_decomposition.py:2067), so it has to live on our class as well.

A path is a list alternating edges ``('s'|'c'|'d', index)`` and nodes ``(chr, pos, orientation)``.
Reads without breakpoints ("concordant reads") are the bulk of the work — one ``alignment_to_path`` per alignment
record of every amplicon interval — and are handled for all records at once: the walk of such a record depends only on
the first and last sequence edge it keeps after the overlap trimming, so records are classified by that pair with
binary searches over the sorted sequence edges and every distinct pair is walked once.
"""
from __future__ import annotations


import numpy as np

EDGE_SLOT = {'s': 0, 'c': 1, 'd': 2}


def _overlap(a, b):
    return a[0] == b[0] and int(a[1]) <= int(b[2]) and int(b[1]) <= int(a[2])


def valid_path(g, path) -> bool:
    """pc:10-45 — alternating edges/nodes, sequence edges at both ends, every node incident to both neighbours."""
    if len(path) <= 3 or len(path) % 2 == 0:
        return False
    if path[0][0] != 's' or path[-1][0] != 's':
        return False
    for i, item in enumerate(path):
        if i % 2 == 0:
            if len(item) != 2:
                return False
            continue
        if len(item) != 3:
            return False
        e1, e2 = path[i - 1], path[i + 1]
        try:
            if (e1[0] == 's') == (e2[0] == 's'):
                return False
            if e1[1] not in g.nodes[item][EDGE_SLOT[e1[0]]] or e2[1] not in g.nodes[item][EDGE_SLOT[e2[0]]]:
                return False
        except Exception:
            return False
    return True


def traverse_through_sequence_edge(g, start_node, end_node):
    """pc:304-342 — walk sequence/concordant edges from start_node until end_node (or the end of the interval)."""
    assert start_node[2] != end_node[2]
    path = [start_node]
    node = start_node
    while True:
        seqi = g.nodes[node][0][0]
        e = g.sequence_edges[seqi]
        far = (e[0], e[2], '+') if node[2] == '-' else (e[0], e[1], '-')
        path.append(('s', seqi))
        path.append(far)
        if far == end_node:
            return path
        try:
            ci = g.nodes[far][1][0]
        except Exception:
            return path                      # alignments spanning two amplicon intervals are cut here
        path.append(('c', ci))
        ce = g.concordant_edges[ci]
        node = (ce[0], ce[1], ce[2])
        if node == far:
            node = (ce[3], ce[4], ce[5])
        path.append(node)


def alignment_to_path(g, rint, min_overlap=500):
    """pc:48-88 — path of a single non-chimeric alignment [chr, start, end]."""
    lst = [k for k, e in enumerate(g.sequence_edges) if _overlap(rint, e)]
    if not lst:
        return []
    lst.sort(key=lambda k: g.sequence_edges[k][1])
    ov = lambda k: min(g.sequence_edges[k][2], rint[2]) - max(g.sequence_edges[k][1], rint[1])
    if len(lst) > 1 and ov(lst[0]) < min_overlap:
        del lst[0]
    while len(lst) > 1 and g.sequence_edges[lst[0]][7] < min_overlap:
        del lst[0]
    if len(lst) > 1 and ov(lst[-1]) < min_overlap:
        del lst[-1]
    while len(lst) > 1 and g.sequence_edges[lst[-1]][7] < min_overlap:
        del lst[-1]
    if len(lst) <= 2:
        return []
    a, b = g.sequence_edges[lst[0]], g.sequence_edges[lst[-1]]
    return traverse_through_sequence_edge(g, (a[0], a[1], '-'), (b[0], b[2], '+'))[1:-1]


def _edges_under(g, al):
    """Sequence edges under one local alignment, ordered along the read direction, with the read orientation."""
    fwd = al[-1] == '+'
    probe = al if fwd else [al[0], al[2], al[1]]
    lst = [[k, '+' if fwd else '-'] for k, e in enumerate(g.sequence_edges) if _overlap(probe, e)]
    lst.sort(key=lambda t: g.sequence_edges[t[0]][1], reverse=not fwd)
    return lst, fwd


def _concordant_between(g, left_edge, right_edge):
    """Index of the concordant edge joining two adjacent sequence edges (left_edge ends where right_edge starts)."""
    for ci, ce in enumerate(g.concordant_edges):
        if ce[0] == left_edge[0] and left_edge[2] == ce[1] and right_edge[1] == ce[4]:
            return ci
    return None


def chimeric_alignment_to_path_l(g, rints, ai, bp_node, min_overlap=500):
    """pc:91-181 — from alignment ``ai`` up to ``bp_node`` (result ends with that node)."""
    al = rints[ai]
    lst, fwd = _edges_under(g, al)
    if not lst:
        return []
    E = g.sequence_edges
    lo_al, hi_al = (al[1], al[2]) if fwd else (al[2], al[1])
    if len(lst) > 1 and min(E[lst[0][0]][2], hi_al) - max(E[lst[0][0]][1], lo_al) < min_overlap:
        del lst[0]
    while lst and E[lst[0][0]][7] < min_overlap:
        del lst[0]
    while lst:
        e = E[lst[-1][0]]
        tail = (e[0], e[2], '+') if fwd else (e[0], e[1], '-')
        if tail == bp_node:
            break
        del lst[-1]
    if not lst:
        return []
    path = []
    for si, (k, o) in enumerate(lst):
        e = E[k]
        path.append(('s', k))
        path.append((e[0], e[2], '+') if fwd else (e[0], e[1], '-'))
        if si < len(lst) - 1:
            nx = E[lst[si + 1][0]]
            if fwd and e[2] + 1 == nx[1]:
                ci = _concordant_between(g, e, nx)
                if ci is not None:
                    path.append(('c', ci))
                    path.append((e[0], nx[1], '-'))
            if not fwd and e[1] - 1 == nx[2]:
                ci = _concordant_between(g, nx, e)
                if ci is not None:
                    path.append(('c', ci))
                    path.append((e[0], nx[2], '+'))
    return path


def chimeric_alignment_to_path_r(g, rints, ai, bp_node, min_overlap=500):
    """pc:184-277 — from ``bp_node`` into alignment ``ai`` (result starts with that node)."""
    ar = rints[ai]
    lst, fwd = _edges_under(g, ar)
    if not lst:
        return []
    E = g.sequence_edges
    lo_al, hi_al = (ar[1], ar[2]) if fwd else (ar[2], ar[1])
    if min(E[lst[-1][0]][2], hi_al) - max(E[lst[-1][0]][1], lo_al) < 500:
        del lst[-1]
    if not lst:
        return []
    while lst and E[lst[-1][0]][7] < 500:
        del lst[-1]
    while lst:
        e = E[lst[0][0]]
        head = (e[0], e[1], '-') if fwd else (e[0], e[2], '+')
        if head == bp_node:
            break
        del lst[0]
    if not lst:
        return []
    path = []
    for si, (k, o) in enumerate(lst):
        e = E[k]
        path.append((e[0], e[1], '-') if fwd else (e[0], e[2], '+'))
        path.append(('s', k))
        if si < len(lst) - 1:
            nx = E[lst[si + 1][0]]
            if fwd and e[2] + 1 == nx[1]:
                ci = _concordant_between(g, e, nx)
                if ci is not None:
                    path.append((e[0], e[2], '+'))
                    path.append(('c', ci))
            if not fwd and e[1] - 1 == nx[2]:
                ci = _concordant_between(g, nx, e)
                if ci is not None:
                    path.append((e[0], e[1], '-'))
                    path.append(('c', ci))
    return path


def _disc_nodes(g, di):
    d = g.discordant_edges[di]
    return (d[0], d[1], d[2]), (d[3], d[4], d[5])


def chimeric_alignment_to_path_i(g, rints, ai1, ai2, di):
    """pc:280-301 — two alignments joined by discordant edge ``di``."""
    n1, n2 = _disc_nodes(g, di)
    if ai1 > ai2:
        return chimeric_alignment_to_path_l(g, rints, ai2, n2) + [('d', di)] + chimeric_alignment_to_path_r(g, rints, ai1, n1)
    return chimeric_alignment_to_path_l(g, rints, ai1, n1) + [('d', di)] + chimeric_alignment_to_path_r(g, rints, ai2, n2)


def chimeric_alignment_to_path(g, rints, ai_list, bp_list):
    """pc:345-375 — a chain of alignments joined by several discordant edges."""
    path = []
    last = ()
    for i, di in enumerate(bp_list):
        n1, n2 = _disc_nodes(g, di)
        a, b = ai_list[i][0], ai_list[i][1]
        enter, leave, first_ai, last_ai = (n2, n1, b, a) if a > b else (n1, n2, a, b)
        if i == 0:
            path = chimeric_alignment_to_path_l(g, rints, first_ai, enter) + [('d', di)]
            last = leave
        else:
            path += traverse_through_sequence_edge(g, last, enter)
            path.append(('d', di))
            last = leave
            if i == len(bp_list) - 1:
                path += chimeric_alignment_to_path_r(g, rints, last_ai, leave)
    return path


# ----------------------------------------------------------------------------------------------
# all non-chimeric alignment records of an interval at once
# ----------------------------------------------------------------------------------------------
def classify_alignments(g, chrom: str, start: np.ndarray, end: np.ndarray, min_overlap=500):
    """Vectorised front half of ``alignment_to_path`` for many alignments [chrom, start, end] on one chromosome.

    Returns (lo, hi): indices (into the chromosome-sorted edge list returned as third value) of the first / last
    sequence edge each alignment keeps after trimming, or lo > hi when the alignment yields no path.
    """
    idx = [k for k, e in enumerate(g.sequence_edges) if e[0] == chrom]
    idx.sort(key=lambda k: g.sequence_edges[k][1])
    n = len(start)
    if not idx:
        return np.ones(n, dtype=np.int64), np.zeros(n, dtype=np.int64), idx
    st = np.array([g.sequence_edges[k][1] for k in idx], dtype=np.int64)
    en = np.array([g.sequence_edges[k][2] for k in idx], dtype=np.int64)
    size = np.array([g.sequence_edges[k][7] for k in idx], dtype=np.int64)
    # edges are disjoint and sorted: the overlapped ones (start <= e[2] and e[1] <= end) form a contiguous range
    lo = np.searchsorted(en, start, side="left")
    hi = np.searchsorted(st, end, side="right") - 1
    m = len(idx)
    big = size >= min_overlap
    # next / previous edge of sufficient size, for the "while size < min_overlap: drop" loops
    nxt_big = np.full(m + 1, m, dtype=np.int64)
    for k in range(m - 1, -1, -1):
        nxt_big[k] = k if big[k] else nxt_big[k + 1]
    prv_big = np.full(m + 1, -1, dtype=np.int64)
    for k in range(m):
        prv_big[k + 1] = k if big[k] else prv_big[k]
    ok = lo <= hi
    loc, hic = np.clip(lo, 0, m - 1), np.clip(hi, 0, m - 1)
    # front: drop the first edge when the overlap with it is short (only while more than one edge is left) ...
    short = ok & (hi > lo) & (np.minimum(en[loc], end) - np.maximum(st[loc], start) < min_overlap)
    lo = np.where(short, lo + 1, lo)
    # ... then drop small edges while more than one is left
    lo = np.where(ok, np.minimum(nxt_big[np.clip(lo, 0, m)], hi), lo)
    loc = np.clip(lo, 0, m - 1)
    short = ok & (hi > lo) & (np.minimum(en[hic], end) - np.maximum(st[hic], start) < min_overlap)
    hi = np.where(short, hi - 1, hi)
    hi = np.where(ok, np.maximum(prv_big[np.clip(hi + 1, 0, m)], lo), hi)
    keep = ok & (hi - lo + 1 > 2)
    return np.where(keep, lo, 1), np.where(keep, hi, 0), idx
