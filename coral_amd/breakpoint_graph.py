"""BreakpointGraph container, long-read CN assignment and the ``*_graph.txt`` / ``*_breakpoints.txt`` writers.

Mirrors the surface of the reference's ``breakpoint_graph.py`` that the graph-build path uses
(/root/reference/src/breakpoint_graph.py:83-207 container, :348-363 sort_edges, :495-606 compute_cn_lr,
:805-822 and :845-854 writers) so that downstream consumers (cycle decomposition, path constraints, plot)
see the same fields:

    sequence_edges[i]   = [chr, l, r, sr_count, sr_flag, lr_count, lr_nc, size, cn]
    concordant_edges[i] = [c1, p1, o1, c2, p2, o2, sr_count, sr_flag, lr_count, reads:set[str], cn]
    discordant_edges[i] = [c1, p1, o1, c2, p2, o2, sr_count, sr_flag, sr_cn, lr_count, reads:set[(name,i,j)], cn]
    nodes               = {(chr, pos, o): [[seq], [conc], [disc], [src]]}   (insertion ordered)

The CN step does not use cvxopt: ``solve_cn_lr`` minimises the reference's objective (bg:546-556) under
the same balance constraints (bg:531-543) by Newton's method on the KKT system to machine precision.
The edge-list functions below work on ANY object with these attributes, so a maintainer can keep the
reference's own BreakpointGraph class (with its cycle-step helpers) as the container: see INTEGRATION.md.
"""
from __future__ import annotations

import logging
import math
import warnings

import os

import numpy as np

from .global_names import chr_idx


class BreakpointGraph:
    def __init__(self):
        self.amplicon_intervals = []
        self.sequence_edges = []
        self.concordant_edges = []
        self.discordant_edges = []
        self.source_edges = []
        self.nodes = dict()
        self.endnodes = dict()
        self.max_cn = 0.0

    # -- nodes ---------------------------------------------------------------------------------
    def add_node(self, node_):
        if type(node_) != tuple or len(node_) != 3:
            raise Exception("Breakpoint node must be of form (chr, pos, orientation).")
        self.nodes[node_] = [[], [], [], []]          # re-adding resets the adjacency, as bg:122-124 does

    def add_endnode(self, node_):
        if type(node_) != tuple or len(node_) != 3:
            raise Exception("Breakpoint node must be of form (chr, pos, orientation).")
        if node_ not in self.endnodes:
            self.endnodes[node_] = []
        else:
            warnings.warn("Node corresponding to interval end already exists.")

    # -- edges ---------------------------------------------------------------------------------
    def add_sequence_edge(self, chr, l, r, sr_count=-1, sr_flag='d', lr_count=-1, lr_nc=0, cn=0.0):
        if (chr, l, '-') not in self.nodes or (chr, r, '+') not in self.nodes:
            raise Exception("Breakpoint node must be added first.")
        k = len(self.sequence_edges)
        self.nodes[(chr, l, '-')][0].append(k)
        self.nodes[(chr, r, '+')][0].append(k)
        self.sequence_edges.append([chr, l, r, sr_count, sr_flag, lr_count, lr_nc, r - l + 1, cn])

    def add_concordant_edge(self, chr1, pos1, o1, chr2, pos2, o2, sr_count=-1, sr_flag='d', lr_count=-1,
                            reads=None, cn=0.0):
        if chr1 != chr2 or pos2 != pos1 + 1 or o1 != '+' or o2 != '-':
            raise Exception("Invalid concordant edge.")
        if (chr1, pos1, o1) not in self.nodes or (chr2, pos2, o2) not in self.nodes:
            raise Exception("Breakpoint node must be added first.")
        k = len(self.concordant_edges)
        self.nodes[(chr1, pos1, o1)][1].append(k)
        self.nodes[(chr2, pos2, o2)][1].append(k)
        self.concordant_edges.append([chr1, pos1, o1, chr2, pos2, o2, sr_count, sr_flag, lr_count,
                                      set() if reads is None else reads, cn])

    def add_discordant_edge(self, chr1, pos1, o1, chr2, pos2, o2, sr_count=-1, sr_flag='d', sr_cn=0.0, lr_count=-1,
                            reads=None, cn=0.0):
        if (chr1, pos1, o1) not in self.nodes or (chr2, pos2, o2) not in self.nodes:
            raise Exception("Breakpoint node must be added first.")
        k = len(self.discordant_edges)
        self.nodes[(chr1, pos1, o1)][2].append(k)
        self.nodes[(chr2, pos2, o2)][2].append(k)
        for nd in ((chr1, pos1, o1), (chr2, pos2, o2)):
            if nd in self.endnodes:
                self.endnodes[nd].append(k)
        self.discordant_edges.append([chr1, pos1, o1, chr2, pos2, o2, sr_count, sr_flag, sr_cn, lr_count,
                                      set() if reads is None else reads, cn])

    def add_source_edge(self, chr1, pos1, o1, sr_count=0, sr_flag='d', sr_cn=0.0, lr_cn=0.0, cn=0.0):
        if (chr1, pos1, o1) not in self.nodes:
            raise Exception("Breakpoint node must be added first.")
        self.nodes[(chr1, pos1, o1)][3].append(len(self.source_edges))
        self.source_edges.append(['source', -1, '-', chr1, pos1, o1, sr_count, sr_flag, sr_cn, lr_cn, cn])

    def sort_edges(self):
        self.sequence_edges.sort(key=lambda e: (chr_idx[e[0]], e[1]))
        self.concordant_edges.sort(key=lambda e: (chr_idx[e[0]], e[1]))
        for k, e in enumerate(self.sequence_edges):
            self.nodes[(e[0], e[1], '-')][0] = [k]
            self.nodes[(e[0], e[2], '+')][0] = [k]
        for k, e in enumerate(self.concordant_edges):
            self.nodes[(e[0], e[1], e[2])][1] = [k]
            self.nodes[(e[3], e[4], e[5])][1] = [k]

    def compute_cn_lr(self, normal_cov_lr):
        compute_cn_lr(self, normal_cov_lr)

    # -- what the (unchanged) cycle step calls on every graph it receives ------------------------------
    def infer_max_seq_multiplicity(self, gain=5.0, size_cutoff=10000, multiplicity=2):
        """Largest multiplicity a sequence edge may take in a cycle / path (bg:609-627): over the edges of at least
        ``size_cutoff`` bp and CN >= ``gain``, round(max CN / length-weighted mean CN) + 1; ``multiplicity`` without any."""
        return infer_max_seq_multiplicity(self.sequence_edges, gain, size_cutoff, multiplicity)

    def infer_discordant_edge_multiplicities(self, max_multiplicity=5):
        """One multiplicity per discordant edge from the long-read supports (bg:630-693; called at cd:146, :623, :1029)."""
        return discordant_edge_multiplicities([e[9] for e in self.discordant_edges], max_multiplicity)

    # -- container maintenance of the reference class (bg:142-164, :210-253) ---------------------------
    def del_endnode(self, node_):
        if node_ in self.endnodes:
            del self.endnodes[node_]
        else:
            warnings.warn("Node corresponding to interval end not exists.")

    def del_discordant_endnodes(self):
        for nd in [nd for nd, edges in self.endnodes.items() if len(edges) > 0]:
            del self.endnodes[nd]

    def _drop_and_renumber(self, adjacency_lists, gone, index_map):
        # the reference's loop (bg:217-228, :248-253): `del lst[i]` inside `for i in range(len(lst))`, kept literally so the
        # behaviour (including its IndexError when an entry is deleted before the end of a list) is the same
        for lst in adjacency_lists:
            for i in range(len(lst)):
                if lst[i] in gone:
                    del lst[i]
                else:
                    lst[i] = index_map[lst[i]]

    def del_discordant_edges(self, del_list, bpi_map):
        gone = sorted(del_list, reverse=True)
        for k in gone:
            del self.discordant_edges[k]
        self._drop_and_renumber(self.endnodes.values(), gone, bpi_map)
        self._drop_and_renumber([adj[2] for adj in self.nodes.values()], gone, bpi_map)

    def del_source_edges(self, del_list, srci_map):
        gone = sorted(del_list, reverse=True)
        for k in gone:
            del self.source_edges[k]
        self._drop_and_renumber([adj[3] for adj in self.nodes.values()], gone, srci_map)

    # -- walks along the sequence edges used by the graph readers (bg:696-765) -------------------------
    def _walk(self, chr, pos, cutoff, here, ahead, step):
        """Distance walked from ``pos`` over consecutive sequence edges until a node with breakpoint edges or ``cutoff``.
        ``here(p)`` = node whose discordant list stops the walk, ``ahead(p)`` = node whose sequence edge is crossed next."""
        dist, p = -1, pos
        while ahead(p) in self.nodes:
            if p != pos and len(self.nodes[here(p)][2]) > 0:
                break
            if dist >= cutoff:
                break
            seglen = self.sequence_edges[self.nodes[ahead(p)][0][0]][7]
            dist = max(dist, 0) + seglen
            p += step * seglen
        return dist

    def nextminus(self, chr, pos, min_bp_match_cutoff_=100):
        return self._walk(chr, pos, min_bp_match_cutoff_, lambda p: (chr, p, '-'), lambda p: (chr, p, '-'), +1)

    def lastminus(self, chr, pos, min_bp_match_cutoff_=100):
        return self._walk(chr, pos, min_bp_match_cutoff_, lambda p: (chr, p, '-'), lambda p: (chr, p - 1, '+'), -1)

    def nextplus(self, chr, pos, min_bp_match_cutoff_=100):
        return self._walk(chr, pos, min_bp_match_cutoff_, lambda p: (chr, p, '+'), lambda p: (chr, p + 1, '-'), +1)

    def lastplus(self, chr, pos, min_bp_match_cutoff_=100):
        return self._walk(chr, pos, min_bp_match_cutoff_, lambda p: (chr, p, '+'), lambda p: (chr, p, '+'), -1)


# ----------------------------------------------------------------------------------------------
# multiplicities for the cycle step (bg:17-80, :609-693)
# ----------------------------------------------------------------------------------------------
def infer_max_seq_multiplicity(sequence_edges, gain=5.0, size_cutoff=10000, multiplicity=2):
    big = [(e[-1], e[7]) for e in sequence_edges if e[7] >= size_cutoff and e[-1] >= gain]
    if not big:
        return multiplicity
    cn, size = [b[0] for b in big], [b[1] for b in big]
    return int(round(max(cn) / np.average(cn, weights=size))) + 1


def _level_above(ratio, level):
    """Smallest multiplicity m >= level with ratio < m + 0.5 (the reference avoids int(round()) on .5 ties, bg:682-683)."""
    while ratio >= level + 0.5:
        level += 1
    return level


def _score_run(rc, lo, hi, max_multiplicity):
    """One run rc[lo..hi] of the ascending supports: can it be explained as a base group of multiplicity 1 followed by
    groups of multiplicity 2, 3, ...?  Returns (valid, index of the last base entry, score) — ``test_clustering`` (bg:17-71).
    Score = Σ log2 jumps between consecutive groups − Σ |m − mean(group / base mean)|; a run needs Σ deviations < 1."""
    if lo == hi:
        return True, lo, 0.0
    p = rc[lo:hi + 1]
    if p[-1] < p[0] * 2.0:
        return True, hi, 0.0
    n_base = next(k for k in range(len(p)) if not p[k] < p[0] * 2.0)
    if p[-1] / np.average(p[:n_base]) >= max_multiplicity + 0.5:
        return False, None, None
    best_score, best_base, best_dev = -10.0, n_base, 1.0
    for nb in range(n_base, 0, -1):                      # shrink the base group from the right
        base = np.average(p[:nb])
        if p[nb] / base < 1.5:
            continue
        m = _level_above(p[nb] / base, 2)
        jumps = math.log2(p[nb]) - math.log2(p[nb - 1])
        groups, start = {}, nb
        for i in range(nb, len(p)):
            if p[i] / base >= m + 0.5:
                jumps += math.log2(p[i]) - math.log2(p[i - 1])
                groups[m] = (start, i - 1)
                start = i
                m = _level_above(p[i] / base, m)
        groups[m] = (start, len(p) - 1)
        if m > max_multiplicity:
            continue
        if any(b - a >= nb for a, b in groups.values()):     # no group may outnumber the base group
            continue
        dev = sum([abs(mm - np.average(p[groups[mm][0]: groups[mm][1] + 1] / base)) for mm in range(2, m + 1) if mm in groups])
        if jumps - dev > best_score:
            best_score, best_dev, best_base = jumps - dev, dev, nb
    if best_dev < 1.0:
        return True, best_base + lo - 1, best_score
    return False, None, None


def discordant_edge_multiplicities(supports, max_multiplicity=5):
    """Multiplicity of every discordant edge from its long-read support (bg:630-693).

    Supports within a factor of two of each other all get 1.  Otherwise the ascending supports are cut into the FEWEST
    contiguous runs that are each explainable by ``_score_run``; among the cuttings with that many runs the one with the
    largest Σ run scores + Σ log2 gaps between neighbouring runs wins (first one in lexicographic order of the cut
    positions on ties), and inside every run the entries after the base group get the multiplicity their ratio to the
    base mean rounds to (never decreasing along the run)."""
    from itertools import combinations
    n = len(supports)
    if n == 0:
        return []
    order = np.argsort(supports)
    rc = sorted(supports)
    if math.log2(rc[-1]) - math.log2(rc[0]) < 1.0:
        return [1] * n
    best = None
    for n_runs in range(1, n + 1):
        best_total = -10.0
        for cuts in combinations(range(1, n), n_runs - 1):
            bounds = [0] + list(cuts) + [n]
            runs = [(bounds[k], bounds[k + 1] - 1) for k in range(n_runs)]
            total, bases = 0.0, []
            for k, (lo, hi) in enumerate(runs):
                ok, base_end, score = _score_run(rc, lo, hi, max_multiplicity)
                if not ok:
                    break
                total += score
                bases.append(base_end)
                if k > 0:
                    total += math.log2(rc[lo]) - math.log2(rc[runs[k - 1][1]])
            else:
                if best is None:
                    best = ([], [])                       # a valid cutting exists at this run count
                if total > best_total:
                    best_total, best = total, (runs, bases)
        if best is not None:
            break
    runs, bases = best
    in_sorted = []
    for (lo, hi), base_end in zip(runs, bases):
        in_sorted += [1] * (base_end - lo + 1)
        if base_end + 1 > hi:
            continue
        base = np.average(rc[lo: base_end + 1])
        m = _level_above(rc[base_end + 1] / base, 2)
        for i in range(base_end + 1, hi + 1):
            m = _level_above(rc[i] / base, m)
            in_sorted.append(m)
    rank = {int(src): k for k, src in enumerate(order)}        # position of edge i in the ascending order
    return [in_sorted[rank[i]] for i in range(n)]


# ----------------------------------------------------------------------------------------------
# CN assignment (replaces cvxopt.solvers.cp at bg:558-563)
# ----------------------------------------------------------------------------------------------
def cn_problem(g, normal_cov):
    """Weights (bg:514-525) and the dense balance matrix (bg:531-543; variable order seq, conc, disc, src;
    one row per non-end node in ``nodes`` insertion order; entries are assigned, not accumulated)."""
    ls, lc, ld, lsrc = len(g.sequence_edges), len(g.concordant_edges), len(g.discordant_edges), len(g.source_edges)
    n = ls + lc + ld + lsrc
    w_lin = np.empty(n)
    w_log = np.empty(n)
    w_inv = np.zeros(n)
    for k, e in enumerate(g.sequence_edges):
        w_lin[k] = 0.5 * normal_cov * e[7]
        w_log[k] = -0.5
        w_inv[k] = 0.5 * e[6] ** 2 / (normal_cov * e[7])
    for k, e in enumerate(g.concordant_edges):
        w_lin[ls + k] = normal_cov
        w_log[ls + k] = e[8] * 1.0
    for k, e in enumerate(g.discordant_edges):
        w_lin[ls + lc + k] = normal_cov
        w_log[ls + lc + k] = e[9] * 1.0
    for k, e in enumerate(g.source_edges):
        w_lin[ls + lc + ld + k] = 0.5 * normal_cov
        w_log[ls + lc + ld + k] = -0.5
        w_inv[ls + lc + ld + k] = 0.5 * e[-1] ** 2 / normal_cov
    interior = [nd for nd in g.nodes if nd not in g.endnodes]
    A = np.zeros((len(interior), n))
    for row, nd in enumerate(interior):
        adj = g.nodes[nd]
        A[row, adj[0]] = 1
        A[row, [ls + k for k in adj[1]]] = -1
        A[row, [ls + lc + k for k in adj[2]]] = -1
        A[row, [ls + lc + ld + k for k in adj[3]]] = -1
    return w_inv, w_lin, w_log, A


def _independent_rows(A, tol=1e-9):
    """Indices (ascending) of a maximal linearly independent subset of the rows of A (entries are 0 / ±1): the rows are taken in
    order and a row is kept when it is not in the span of the rows kept so far (Gram-Schmidt against an orthonormal basis of that
    span, re-orthogonalised once; a dependent 0 / ±1 row leaves a residual at rounding level, an independent one a residual of
    order one).  Redundant balance rows are consistent (right-hand side 0), so which independent subset is kept changes
    nothing: the kept rows span the same constraint space.  numpy only (a scipy import would cost the first build 85 ms)."""
    m, n = A.shape
    if m == 0:
        return []
    if os.environ.get("CORAL_CN_SOLVER", "native") != "python":          # the same selection in libcoral_hip (coral_independent_rows)
        from . import _lib
        Ad = np.ascontiguousarray(A, dtype=np.float64)
        keep = np.zeros(m, dtype=np.uint8)
        if _lib.lib().coral_independent_rows(m, n, Ad.ctypes.data, keep.ctypes.data, float(tol)) < 0:
            raise _lib.CoralHipError("coral_independent_rows: bad arguments")
        return np.nonzero(keep)[0].tolist()
    Q = np.zeros((min(m, n), n))
    keep = []
    for i in range(m):
        v = A[i].astype(np.float64)
        norm = float(np.sqrt(v @ v))
        if norm == 0.0:
            continue
        k = len(keep)
        if k == Q.shape[0]:
            break
        if k:
            B = Q[:k]
            v = v - B.T @ (B @ v)
            v = v - B.T @ (B @ v)
        res = float(np.sqrt(v @ v))
        if res > tol * max(1.0, norm) and res > 1e-7 * norm:
            Q[k] = v / res
            keep.append(i)
    return keep


def _newton_step(h, A, r, n, p):
    """Solve [[diag(h), Aᵀ], [A, 0]] [dx; dnu] = -r by eliminating the variables with h > 0 (dx_i = -(r_i + (Aᵀdnu)_i) / h_i):
    what remains is a system in dnu and the few variables with h == 0 (concordant edges without read support), less than
    half the size of the KKT matrix.  Returns None when that system is singular."""
    pos = h > 0
    zero = np.nonzero(~pos)[0]
    rd, rp = r[:n], r[n:]
    Ap = A[:, pos]
    hinv = 1.0 / h[pos]
    S = (Ap * hinv) @ Ap.T
    nz = len(zero)
    if nz:
        Az = A[:, zero]
        M = np.zeros((p + nz, p + nz))
        M[:p, :p] = S
        M[:p, p:] = -Az
        M[p:, :p] = Az.T
        rhs = np.concatenate([rp - Ap @ (hinv * rd[pos]), -rd[zero]])
    else:
        M, rhs = S, rp - Ap @ (hinv * rd[pos])
    try:
        sol = np.linalg.solve(M, rhs)
    except np.linalg.LinAlgError:
        return None
    if not np.all(np.isfinite(sol)):
        return None
    dnu = sol[:p]
    dx = np.empty(n)
    dx[pos] = -(rd[pos] + Ap.T @ dnu) * hinv
    if nz:
        dx[zero] = sol[p:]
    return np.concatenate([dx, dnu])


def solve_cn_lr(w_inv, w_lin, w_log, A, max_iter=200):
    """argmin  Σ w_inv/x + w_lin·x − w_log·log x   s.t.  A x = 0,  x > 0,   started at x = 1 (bg:546-548).

    Infeasible-start Newton on the KKT system [[H, Aᵀ], [A, 0]] (H diagonal, possibly with zero entries for
    concordant edges without read support, so no Schur complement), with a backtracking search on the KKT
    residual that keeps x strictly positive.  Iterates until the Newton step is below 1e-13 relative (or the
    residual can no longer be reduced in float64).  Linearly dependent balance rows are removed first so the KKT
    matrix is non-singular and a plain LU solve can be used.
    """
    n = len(w_lin)
    if A.shape[0]:
        A = A[_independent_rows(A)]
    p = A.shape[0]
    native = _solve_native(w_inv, w_lin, w_log, A, max_iter) if (n and os.environ.get("CORAL_CN_SOLVER", "native") != "python") else None
    if native is not None:
        return _checked(native[0], native[1], w_inv, w_lin, w_log, A)
    x = np.ones(n)
    nu = np.zeros(p)
    K = None                                     # full KKT matrix, built only if the reduced solve below is not applicable

    def kkt_residual(x, nu):
        return np.concatenate([w_lin - w_log / x - w_inv / (x * x) + A.T @ nu, A @ x])

    r = kkt_residual(x, nu)
    for _ in range(max_iter):
        h = w_log / (x * x) + 2.0 * w_inv / (x * x * x)
        step = _newton_step(h, A, r, n, p)
        if step is None:                         # (not seen in practice) the reduced system is singular: full KKT matrix
            if K is None:
                K = np.zeros((n + p, n + p))
                K[:n, n:] = A.T
                K[n:, :n] = A
            K[np.arange(n), np.arange(n)] = h
            try:
                step = np.linalg.solve(K, -r)
            except np.linalg.LinAlgError:
                step = np.linalg.lstsq(K, -r, rcond=None)[0]
        dx, dnu = step[:n], step[n:]
        t = 1.0
        shrink = dx < 0
        if shrink.any():
            t = min(1.0, 0.99 * float(np.min(-x[shrink] / dx[shrink])))
        r0 = np.linalg.norm(r)
        improved = False
        while t > 1e-10:
            r_new = kkt_residual(x + t * dx, nu + t * dnu)
            if np.linalg.norm(r_new) <= (1.0 - 0.01 * t) * r0:
                improved = True
                break
            t *= 0.5
        small = float(np.max(np.abs(dx) / x)) < 1e-13
        if not improved:
            if small:                       # at the float64 floor: take the (tiny) full step and stop
                x = x + dx
            break
        x = x + t * dx
        nu = nu + t * dnu
        r = r_new
        if small:
            break
    return _checked(x, nu, w_inv, w_lin, w_log, A)


def _solve_native(w_inv, w_lin, w_log, A, max_iter):
    """The same iteration in libcoral_hip (coral_cn_solve: sparse columns, tens of microseconds per Newton step); None when the
    reduced Newton system turned out singular there — the general path below handles that case."""
    import ctypes as C
    from . import _lib
    n, p = len(w_lin), A.shape[0]
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    wi, wl, wg, Ad = f(w_inv), f(w_lin), f(w_log), f(A)
    x, nu = np.empty(n), np.zeros(max(p, 1))
    it = C.c_int32(0)
    rc = _lib.lib().coral_cn_solve(n, p, wi.ctypes.data, wl.ctypes.data, wg.ctypes.data, Ad.ctypes.data if p else None, int(max_iter),
                                   x.ctypes.data, nu.ctypes.data, C.byref(it))
    if rc == 1:
        return None
    _lib.check(rc, "coral_cn_solve")
    return x, nu[:p]


def _checked(x, nu, w_inv, w_lin, w_log, A):
    """x, after the product's own check of the point it returns (nu: the multipliers the solver ended with)."""
    n, p = len(w_lin), A.shape[0]

    def kkt_residual(x, nu):
        return np.concatenate([w_lin - w_log / x - w_inv / (x * x) + A.T @ nu, A @ x])
    # A stalled solve must not pass silently: relative KKT residual of the returned point (the exact optimum sits at ~1e-13,
    # cvxopt's own stopping rule at ~1e-7).  A variable without log / inverse term (a concordant edge nobody supports) may be
    # driven to the boundary x -> 0, where the condition is complementary slackness: gradient >= 0 and gradient * x -> 0.
    res = kkt_residual(x, nu)
    g = res[:n]
    free = (w_log == 0) & (w_inv == 0)
    stat = np.where(free & (g > 0), np.minimum(np.abs(g), np.abs(g) * x), np.abs(g))
    scale = max(1.0, float(np.max(np.abs(w_lin) + np.abs(w_log) / x + np.abs(w_inv) / (x * x)))) if n else 1.0
    solve_cn_lr.last_residual = float(np.max(stat) / scale) if n else 0.0
    if p:
        solve_cn_lr.last_residual = max(solve_cn_lr.last_residual, float(np.max(np.abs(res[n:])) / max(1.0, float(np.max(x)))))
    if not solve_cn_lr.last_residual < 1e-8:
        # e.g. a sequence edge without a single aligned base makes the objective unbounded (x -> 0): the reference's cvxopt run
        # ends with status "unknown" there and keeps its last iterate (bg:565-568); say so instead of passing silently
        logging.warning("CN assignment did not converge (relative KKT residual %.3g); the CN column is the last iterate."
                        % solve_cn_lr.last_residual)
    return x


solve_cn_lr.last_residual = 0.0


def compute_cn_lr(g, normal_cov_lr):
    """Fill the CN field of every edge of ``g`` and ``g.max_cn`` (bg:495-606)."""
    ls, lc, ld = len(g.sequence_edges), len(g.concordant_edges), len(g.discordant_edges)
    w_inv, w_lin, w_log, A = cn_problem(g, normal_cov_lr)
    if A.shape[0] > 0:
        x = solve_cn_lr(w_inv, w_lin, w_log, A)
        doubled = [float(v) * 2 for v in x]
        for k in range(ls):
            g.sequence_edges[k][-1] = doubled[k]
        for k in range(lc):
            g.concordant_edges[k][-1] = doubled[ls + k]
        for k in range(ld):
            e = g.discordant_edges[k]
            self_loop = e[0] == e[3] and e[1] == e[4] and e[2] == e[5]
            e[-1] = float(x[ls + lc + k]) if self_loop else doubled[ls + lc + k]      # bg:585-592
        for k in range(len(g.source_edges)):
            g.source_edges[k][-1] = doubled[ls + lc + ld + k]
        for lst in (g.sequence_edges, g.concordant_edges, g.discordant_edges, g.source_edges):
            for e in lst:
                if e[-1] > g.max_cn:
                    g.max_cn = e[-1]
    else:
        assert lc == 0 and ld == 0 and len(g.source_edges) == 0
        for e in g.sequence_edges:
            e[-1] = e[6] * 2.0 / (normal_cov_lr * e[7])
            if e[-1] > g.max_cn:
                g.max_cn = e[-1]
    g.max_cn += 1.0


# ----------------------------------------------------------------------------------------------
# writers
# ----------------------------------------------------------------------------------------------
def graph_text(g) -> str:
    rows = ["SequenceEdge: StartPosition, EndPosition, PredictedCN, AverageCoverage, Size, NumberOfLongReads\n"]
    for e in g.sequence_edges:
        rows.append("sequence\t%s:%s-\t%s:%s+\t%f\t%f\t%d\t%d\n" % (e[0], e[1], e[0], e[2], e[-1], e[6] * 1.0 / e[7], e[7], e[5]))
    rows.append("BreakpointEdge: StartPosition->EndPosition, PredictedCN, NumberOfLongReads\n")
    for e in g.source_edges:
        rows.append("source\t%s:%s%s->%s:%s%s\t%f\t-1\n" % (e[0], e[1], e[2], e[3], e[4], e[5], e[-1]))
    for e in g.concordant_edges:
        rows.append("concordant\t%s:%s%s->%s:%s%s\t%f\t%d\n" % (e[0], e[1], e[2], e[3], e[4], e[5], e[-1], e[8]))
    for e in g.discordant_edges:
        rows.append("discordant\t%s:%s%s->%s:%s%s\t%f\t%d\n" % (e[0], e[1], e[2], e[3], e[4], e[5], e[-1], e[9]))
    return "".join(rows)


def output_breakpoint_graph_lr(g, ogfile):
    with open(ogfile, 'w') as fp:
        fp.write(graph_text(g))


def breakpoint_info_text(g, bp_stats) -> str:
    rows = ["chr1\tpos1\tchr2\tpos2\torientation\tlr_support\tlr_info=[avg1, avg2, std1, std2, mapq1, mapq2]\n"]
    for k, e in enumerate(g.discordant_edges):
        rows.append("%s\t%s\t%s\t%s\t%s%s\t%d\t%s\n" % (e[3], e[4], e[0], e[1], e[5], e[2], e[9], bp_stats[k]))
    return "".join(rows)


def output_breakpoint_info_lr(g, obpfile, bp_stats):
    with open(obpfile, 'w') as fp:
        fp.write(breakpoint_info_text(g, bp_stats))
