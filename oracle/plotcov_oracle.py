"""CPU restatement of the coverage track of the reference's ``plot`` mode — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/plot_amplicons.py:108-132 (sequence edges of a ``*_graph.txt``), :167-184 (amplified
intervals), :376-411 (the windows and their coverage) and breakpoint_utilities.py:419-427 (chromosome order), with plain
Python loops over ``oracle.hostrecords.HostRecords``.  Pinned by tests/golden/plotcov_*.json (real reference run).
"""
from __future__ import annotations


def intervals_from_graph_text(text):
    by_chr = {}
    for line in text.split("\n"):
        s = line.strip().split("\t")
        if s[0] == "sequence":
            c = s[1].split(":")[0]
            by_chr.setdefault(c, []).append([int(s[1].split(":")[1][:-1]), int(s[2].split(":")[1][:-1])])
    out = {}
    for c, edges in by_chr.items():
        lstart, lend = -2, -2
        out[c] = []
        for start, end in edges:
            if start != lend + 1:
                if lstart >= 0:
                    out[c].append([lstart, lend])
                lstart, lend = start, end
            else:
                lend = end
        out[c].append([lstart, lend])
    return out


def sort_chrom_names(chroms):
    def key(x):
        v = x[3:] if x.startswith("chr") else x
        return int(v) if v.isnumeric() else ord(v)
    return sorted(chroms, key=key)


def windows(intervals, plot_bounds=None):
    """[(chrom, start, stop)] in the order plot:376-411 queries them."""
    out = []
    for chrom in sort_chrom_names(intervals.keys()):
        for a, b in intervals[chrom]:
            if plot_bounds:
                if chrom != plot_bounds[0]:
                    continue
                if not (b >= plot_bounds[1] and a <= plot_bounds[2]):
                    continue
            size = 150
            length = (plot_bounds[2] - plot_bounds[1]) if plot_bounds else (b - a)
            if length >= 1000000:
                size = 10000
            elif length >= 100000:
                size = 1000
            for w in range(a, b, size):
                out.append((chrom, w, w + size))
            w = b - ((b - a + 1) % size)
            if w < b:
                out.append((chrom, w, w + size))
    return out


def coverage_track(host, graph_text, plot_bounds=None):
    """[(chrom, start, stop, aligned A/C/G/T bases in [start, stop))] — the rectangle heights are bases / (stop - start)."""
    return [(c, a, b, host.count_coverage_sum(c, a, b)) for c, a, b in windows(intervals_from_graph_text(graph_text), plot_bounds)]
