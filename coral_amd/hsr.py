"""``hsr`` mode: candidate integration points of ecDNA into chromosomes.

Same contract as the reference's ``hsr.locate_hsrs(args)`` (/root/reference/src/hsr.py:54-224): same arguments
(``lr_bam, cycles, cn_seg, output_prefix, normal_cov, bp_match_cutoff, bp_match_cutoff_clustering``), same text on stdout,
same ``integration_sites_<prefix>.png`` in the working directory, same errors (KeyError for a breakpoint on a chromosome
without copy-number rows, SystemExit for a cycles file that is neither ``*_cycles.txt`` nor ``*.bed``).

The work is the graph build's own: the chimeric table of ALL reads comes from ``coral_sa_table`` (K3, the whole-BAM
``fetch`` of hsr.py:21-51), the junction candidates are computed for all SA rows at once on arrays, and clustering +
exact breakpoints are one ``coral_call_breakpoints`` call (hsr.py:149-170).
"""
from __future__ import annotations

import sys
from typing import List

import numpy as np

from . import global_names
from .bpcluster import call_breakpoints
from .chimeric import Candidates, ChimericTable, build_chimeric_table, first_interval_overlap, rows_overlap

_ORI = "+-"

# /root/reference/src/global_names.py:20-25 (hg38 chromosome sizes; public assembly constants)
chr_sizes = {'chr1': 248956422, 'chr2': 242193529, 'chr3': 198295559, 'chr4': 190214555, 'chr5': 181538259,
             'chr6': 170805979, 'chr7': 159345973, 'chr8': 145138636, 'chr9': 138394717, 'chr10': 133797422,
             'chr11': 135086622, 'chr12': 133275309, 'chr13': 114364328, 'chr14': 107043718, 'chr15': 101991189,
             'chr16': 90338345, 'chr17': 83257441, 'chr18': 80373285, 'chr19': 58617616, 'chr20': 64444167,
             'chr21': 46709983, 'chr22': 50818468, 'chrX': 156040895, 'chrY': 57227415}


class HsrResult:
    """What the reference keeps in local variables: candidates, clusters, refined breakpoints, plotted points."""

    def __init__(self):
        self.candidates: List[list] = []
        self.cluster_sizes: List[int] = []
        self.calls: List[list] = []            # [bp (9 fields), support tuples, stats] per accepted bpc2bp call
        self.bp_refined: List[list] = []
        self.bp_stats: List[list] = []
        self.points: List[list] = []


def _merge_walk(segments, walk):
    """Segments of one cycle / path in walk order, neighbours that continue each other on the same strand fused into one
    interval, and the last one folded into the first when the walk closes on itself (cycle2bed.py:29-48)."""
    out = []
    for token in walk:
        sid, strand = token[:-1], token[-1]
        if int(sid) <= 0:                        # segment 0 marks the ends of a linear path
            continue
        chrom, start, end = segments[sid]
        last = out[-1] if out else None
        if last and last[3] == strand == '+' and last[0] == chrom and last[2] + 1 == start:
            last[2] = end
        elif last and last[3] == strand == '-' and last[0] == chrom and last[1] - 1 == end:
            last[1] = start
        else:
            out.append([chrom, start, end, strand])
    head, tail = out[0], out[-1]
    if head[3] == '+' and tail[0] == head[0]:
        if tail[3] == '+' and tail[2] + 1 == head[1]:
            head[1] = tail[1]
            out.pop()
    head, tail = out[0], out[-1]
    if head[3] == '+' and tail[0] == head[0]:
        if tail[3] == '-' and tail[1] - 1 == head[2]:
            head[2] = tail[2]
            out.pop()
    return out


def convert_cycles_to_bed(cycle_fn, output_fn):
    """AmpliconSuite ``*_cycles.txt`` -> bed, what cycle2bed.convert_cycles_to_bed does with its default arguments (the call
    of hsr.py:65): ``Segment`` lines define numbered intervals, every ``Cycle=..;Copy_count=..;Segments=..`` line a walk."""
    segments, walks = {}, {}
    with open(cycle_fn) as fp:
        for line in fp:
            t = line.strip().split()
            if not t:
                continue
            if t[0] == "Segment":
                segments[t[1]] = (t[2], int(t[3]), int(t[4]))
            if t[0].startswith("Cycle"):
                fields = dict(kv.split('=', 1) for kv in t[0].split(';') if '=' in kv)
                walk = fields.get("Segments", "0+,0-").split(',')
                cyclic = walk[0] != "0+" or walk[-1] != "0-"
                walks[int(fields.get("Cycle", 1))] = (cyclic, float(fields.get("Copy_count", 1.0)), _merge_walk(segments, walk))
    print("Creating bed-converted cycles file: " + output_fn)
    with open(output_fn, 'w') as fp:
        fp.write("#chr\tstart\tend\torientation\tcycle_id\tiscyclic\tweight\n")
        for cid in range(1, len(walks) + 1):
            cyclic, weight, intervals = walks[cid]
            for chrom, start, end, strand in intervals:
                fp.write("%s\t%d\t%d\t%s\t%d\t%s\t%f\n" % (chrom, start, end, strand, cid, cyclic, weight))


def junction_candidates(T: ChimericTable, ecdna, chroms, chr_rank) -> Candidates:
    """hsr.py:116-147 for every chimeric read at once.

    A read takes part when, for some ecDNA interval, its FIRST piece overlapping that interval lies inside it.  Junctions
    are (a) adjacent pieces with MAPQ >= 20 of which exactly one overlaps an ecDNA interval, then (b) pieces ri - 1 and
    ri + 1 around a piece of MAPQ < 10 when neither adjacent pair was taken, both have MAPQ >= 20 and piece ri - 1 is
    off the ecDNA (the reference's test on piece ri + 1 is vacuous, see oracle/hsr_oracle.py).  Candidates come out in
    the reference's order: read by read, all of (a) then all of (b).
    """
    n_rows = T.n_rows
    if n_rows == 0 or not ecdna:
        return Candidates()
    tid_of = {c: k for k, c in enumerate(chroms)}
    ivs = [(tid_of.get(c, -1), s, e) for c, s, e in ecdna]
    rows = np.arange(n_rows)
    ec = first_interval_overlap(T, ivs)                      # interval_overlap_l(rr_int[k], ecdna_intervals)
    n_reads = T.n_reads
    on_cycle = np.zeros(n_reads, dtype=bool)
    for (t, s, e) in ivs:
        m = rows_overlap(T, rows, t, s, e)                   # interval_overlap(interval, rr_int[k]) is the same expression
        hit = np.nonzero(m)[0]
        if len(hit) == 0:
            continue
        r_of = T.read[hit]
        firsts = hit[np.concatenate([[True], r_of[1:] != r_of[:-1]])]          # first overlapping piece of each read
        inside = (T.tid[firsts] == t) & (T.ra[firsts] >= s) & (T.rb[firsts] <= e)   # interval_include(rr_int[i], interval)
        on_cycle[T.read[firsts[inside]]] = True
    row_on = on_cycle[T.read]
    same_next = np.zeros(n_rows, dtype=bool)
    same_next[:-1] = T.read[1:] == T.read[:-1]
    same_next2 = np.zeros(n_rows, dtype=bool)
    same_next2[:-2] = T.read[2:] == T.read[:-2]
    hi20 = T.mapq >= 20
    nxt = np.minimum(rows + 1, n_rows - 1)
    nxt2 = np.minimum(rows + 2, n_rows - 1)
    # (a) pieces k, k + 1
    adj = row_on & same_next & hi20 & hi20[nxt] & ((ec == -1) != (ec[nxt] == -1))
    # (b) pieces k, k + 2 around the low-MAPQ piece k + 1
    skip = row_on & same_next2 & ~adj & ~adj[nxt] & (T.mapq[nxt] < 10) & hi20 & hi20[nxt2] & (ec == -1)
    first_row = T.off[T.read]
    a_rows, s_rows = np.nonzero(adj)[0], np.nonzero(skip)[0]
    left = np.concatenate([a_rows, s_rows])
    right = np.concatenate([a_rows + 1, s_rows + 2])
    kind = np.concatenate([np.zeros(len(a_rows), dtype=np.int64), np.ones(len(s_rows), dtype=np.int64)])
    order = np.lexsort((left, kind, T.read[left]))            # read, then (a) before (b), then piece index
    left, right = left[order], right[order]
    if len(left) == 0:
        return Candidates()
    # interval2bp(R1, R2, (r, i, j), gap)  (bu:289-295)
    k1, k2 = chr_rank[T.tid[left]], chr_rank[T.tid[right]]
    bad = np.nonzero((k1 < 0) | (k2 < 0))[0]
    if len(bad):                                                               # global_names.chr_idx[...] at bu:293: R2 is looked up first
        b = bad[0]
        raise KeyError(chroms[T.tid[right][b]] if k2[b] < 0 else chroms[T.tid[left][b]])
    plain = (k2 < k1) | ((k2 == k1) & (T.ra[right] < T.rb[left]))
    i_idx, j_idx = left - first_row[left], right - first_row[right]
    flip = 1 - T.strand[right]
    c = Candidates(
        c1=np.where(plain, T.tid[left], T.tid[right]), p1=np.where(plain, T.rb[left], T.ra[right]),
        o1=np.where(plain, T.strand[left], flip),
        c2=np.where(plain, T.tid[right], T.tid[left]), p2=np.where(plain, T.ra[right], T.rb[left]),
        o2=np.where(plain, flip, T.strand[left]),
        read=T.name_id[T.read[left]], i=np.where(plain, i_idx, j_idx), j=np.where(plain, j_idx, i_idx),
        gap=T.qs[right] - T.qe[left], swapped=np.where(plain, 0, 1), mqa=T.mapq[left], mqb=T.mapq[right])
    return c


def _candidate_lists(c: Candidates, chroms, names) -> List[list]:
    out = []
    for k in range(len(c)):
        out.append([chroms[c.c1[k]], int(c.p1[k]), _ORI[c.o1[k]], chroms[c.c2[k]], int(c.p2[k]), _ORI[c.o2[k]],
                    (names[c.read[k]], int(c.i[k]), int(c.j[k])), int(c.gap[k]), int(c.swapped[k]), int(c.mqa[k]), int(c.mqb[k])])
    return out


def _ecdna_intervals(args):
    """Intervals of the ecDNA from a cycles bed (or an AmpliconSuite ``*_cycles.txt``, converted first) — hsr.py:59-79."""
    path = args.cycles
    if path.endswith("_cycles.txt"):
        sep = "" if args.output_prefix.endswith("/") else "_"
        bed = "%s%sconverted_cycles.bed" % (args.output_prefix, sep)
        convert_cycles_to_bed(path, bed)
        path = bed
    elif not path.endswith(".bed"):
        sys.stderr.write(args.cycles + "\n")
        sys.stderr.write("Cycles file must be either a valid *_cycles.txt file or a converted .bed file!\n")
        sys.exit(1)
    with open(path) as fp:
        rows = [ln.split() for ln in fp if not ln.startswith("#")]
    return [[r[0], int(r[1]), int(r[2])] for r in rows if len(r) >= 3]


def _copy_numbers(cn_seg):
    """{chromosome: [[start, end, cn], ...]} from a CNVkit ``.cns`` (cn = 2 * 2**log2) or a ``.bed`` (4th column) — hsr.py:84-108."""
    per_chrom = {}
    with open(cn_seg) as fp:
        for ln in fp:
            if ln.startswith('chromosome'):
                continue
            f = ln.strip().split()
            if cn_seg.endswith(".cns"):
                cn = 2 * (2 ** float(f[4]))
            elif cn_seg.endswith(".bed"):
                cn = float(f[3])
            else:
                sys.stderr.write(cn_seg + "\n")
                sys.stderr.write("Invalid cn_seg file format!\n")
            per_chrom.setdefault(f[0], []).append([int(f[1]), int(f[2]), cn])
    return per_chrom


def _genome_axis():
    """Start of every chromosome and the tick positions on a 0-100 axis over the concatenated genome (hsr.py:173-183)."""
    total = sum(chr_sizes.values())
    start, ticks, borders, run = {}, [], [], 0
    for c, size in chr_sizes.items():
        start[c] = run * 100.0 / total
        run += size
        ticks.append((run - 0.5 * size) * 100.0 / total)
        if run < total:
            borders.append(run * 100.0 / total)
    return total, start, ticks, borders


def locate_hsrs(args, records=None, device="cuda:0"):
    """hsr.locate_hsrs(args) (hsr.py:54-224).  ``records``: already decoded ``DeviceRecords`` (tests, pipelines)."""
    import matplotlib as mpl
    mpl.use('Agg')
    import matplotlib.pyplot as plt
    from pylab import rcParams
    rcParams['figure.figsize'] = [20, 8]
    rcParams['pdf.fonttype'] = 42
    mpl.rc('xtick', labelsize=25)
    mpl.rc('ytick', labelsize=25)

    res = HsrResult()
    cutoff = args.bp_match_cutoff
    ecdna = _ecdna_intervals(args)
    padded = [(c, s - cutoff, e + cutoff) for c, s, e in ecdna]
    print("ecDNA intervals:")
    for iv in ecdna:
        print(iv)
    cns = _copy_numbers(args.cn_seg)

    if records is None:
        from .bam import load_bam
        from .records import DeviceRecords
        records = DeviceRecords(load_bam(args.lr_bam, getattr(args, "device", device)), getattr(args, "device", device))
    dr = records
    T = build_chimeric_table(dr)                              # whole-BAM fetch (hsr.py:21-51) through coral_sa_table
    print("Fetched %d chimeric alignments." % T.n_reads)
    chroms, names = dr.header_chroms, dr.names
    chr_rank = np.array([global_names.chr_idx.get(c, -1) for c in chroms], dtype=np.int64)
    cands = junction_candidates(T, ecdna, chroms, chr_rank)
    res.candidates = _candidate_lists(cands, chroms, names)

    # clusters -> exact breakpoints -> merged list (hsr.py:149-170); a sub-cluster counts when its distinct support reaches
    # half the normal coverage, which is also the minimum cluster size
    need = float(args.normal_cov) * 0.5
    res.cluster_sizes, calls = call_breakpoints(cands, need, args.bp_match_cutoff_clustering, cutoff, need, False)
    for head, p1, p2, sup, st in calls:
        bp = [chroms[cands.c1[head]], p1, _ORI[cands.o1[head]], chroms[cands.c2[head]], p2, _ORI[cands.o2[head]],
              (names[cands.read[head]], int(cands.i[head]), int(cands.j[head])), int(cands.gap[head]), int(cands.swapped[head])]
        support = [(names[cands.read[k]], int(cands.i[k]), int(cands.j[k])) for k in sup.tolist()]
        res.calls.append([bp, support, st])
        same = [old for old in res.bp_refined
                if (old[0], old[2], old[3], old[5]) == (bp[0], bp[2], bp[3], bp[5])
                and abs(bp[1] - old[1]) <= cutoff and abs(bp[4] - old[4]) < cutoff]
        if same:
            same[0][-1] |= set(support)                       # merged into the FIRST close breakpoint (hsr.py:160-165)
        else:
            res.bp_refined.append(bp + [support])
            res.bp_stats.append(st)
    print("Found %d breakpoints connecting ecDNA and chromosomes." % len(res.bp_refined))

    total, start_of, ticks, borders = _genome_axis()
    for x in borders:
        plt.plot([x, x], [-1, 1000000], 'k--', linewidth=2)

    def touches_ecdna(c, p):                                  # interval_overlap_l([c, p, p], padded intervals) >= 0
        return any(c == ic and lo <= p <= hi for ic, lo, hi in padded)

    limit = float(args.normal_cov) * 2.5
    for bp in res.bp_refined:
        first, second = touches_ecdna(bp[0], bp[1]), touches_ecdna(bp[3], bp[4])
        if first == second:
            continue                                          # both ends on the ecDNA, or neither
        c, p = (bp[3], bp[4]) if first else (bp[0], bp[1])    # the chromosomal end
        if c not in start_of:
            continue
        cn = next((seg[2] for seg in cns[c] if seg[0] < p < seg[1]), 0.0)     # KeyError without CN rows, as the reference
        support = len(bp[-1])
        if cn <= 5.0 and support <= limit:
            print("Breakpoint", bp[:6], "Support = ", support)
            res.points.append([start_of[c] + p * 100.0 / total, support])
            plt.plot(res.points[-1][0], support, 'bo')

    plt.xlim([0, 100])
    plt.ylim([1, 500])
    plt.yscale('log')
    plt.xticks(ticks, list(range(1, 23)) + ['X', 'Y'])
    plt.title(args.output_prefix + " integration loci", fontsize=25)
    plt.ylabel('Long read support', fontsize=25)
    plt.tight_layout()
    image = "integration_sites_" + args.output_prefix + ".png"
    plt.savefig(image)
    plt.close()
    print('\nCreated ' + image)
    return res
