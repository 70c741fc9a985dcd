"""Coverage track of the ``plot`` mode on the MI355X path.

The reference draws, for every amplified interval of a ``*_graph.txt``, one grey rectangle per window of 150 / 1 000 /
10 000 bp whose height is ``sum(count_coverage(chrom, w, w + window)) / window``
(/root/reference/src/plot_amplicons.py:376-411) — thousands of pysam calls that each walk the CIGARs of the overlapping
reads.  Here all windows of a plot are ONE ``coral_segment_coverage`` launch over the HBM-resident records (the kernel of
the graph build's A2 / A10 steps: per-record sums from the fused CIGAR scan, CIGAR re-walk only for records straddling a
window border).  Only the numbers are produced; drawing stays with the reference's matplotlib code.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import kernels


def parse_graph_intervals(graph_fn: str) -> Dict[str, List[List[int]]]:
    """Amplified intervals of a breakpoint-graph file: consecutive sequence edges merged (plot:108-121, :167-184)."""
    by_chr: Dict[str, list] = {}
    with open(graph_fn) as fp:
        for line in fp:
            s = line.strip().split("\t")
            if s[0] == "sequence":
                chrom = s[1].split(":")[0]
                by_chr.setdefault(chrom, []).append((int(s[1].split(":")[1][:-1]), int(s[2].split(":")[1][:-1])))
    out: Dict[str, List[List[int]]] = {}
    for chrom, edges in by_chr.items():
        lstart, lend = -2, -2
        out[chrom] = []
        for start, end in edges:
            if start != lend + 1:
                if lstart >= 0:
                    out[chrom].append([lstart, lend])
                lstart, lend = start, end
            else:
                lend = end
        out[chrom].append([lstart, lend])
    return out


def sort_chrom_names(chromlist):
    """bu:419-427."""
    def sort_key(x):
        val = x[3:] if x.startswith("chr") else x
        return int(val) if val.isnumeric() else ord(val)
    return sorted(chromlist, key=sort_key)


def track_windows(intervals: Dict[str, Sequence[Sequence[int]]], plot_bounds: Optional[Tuple[str, int, int]] = None):
    """(chrom, start, stop) of every rectangle, in drawing order (plot:376-411), as arrays per run of equal window size."""
    out = []
    for chrom in sort_chrom_names(intervals.keys()):
        for a, b in intervals[chrom]:
            if plot_bounds:
                if chrom != plot_bounds[0] or not (b >= plot_bounds[1] and a <= plot_bounds[2]):
                    continue
            window = 150
            length = (plot_bounds[2] - plot_bounds[1]) if plot_bounds else (b - a)
            if length >= 1000000:
                window = 10000
            elif length >= 100000:
                window = 1000
            starts = np.arange(a, b, window, dtype=np.int64)
            tail = b - ((b - a + 1) % window)
            if tail < b:
                starts = np.concatenate([starts, [tail]])
            out.append((chrom, window, starts))
    return out


def coverage_track(dr, intervals, plot_bounds=None, scan=None):
    """[(chrom, start, stop, bases)] for every window of the plot; height of the rectangle = bases / (stop - start).

    ``dr``: DeviceRecords; ``scan``: a previous ``kernels.cigar_scan(dr)`` result to reuse (else it is run once)."""
    runs = track_windows(intervals, plot_bounds)
    if not runs:
        return []
    tid_of = {c: k for k, c in enumerate(dr.header_chroms)}
    segs = np.concatenate([np.stack([np.full(len(st), tid_of[c], dtype=np.int64), st, st + w], axis=1) for c, w, st in runs])
    if scan is None:
        scan = kernels.cigar_scan(dr)
    _, n_bases = kernels.segment_coverage(dr, scan, segs)
    out, k = [], 0
    for c, w, st in runs:
        for s in st.tolist():
            out.append((c, s, s + w, int(n_bases[k])))
            k += 1
    return out
