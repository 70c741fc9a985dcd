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


def convert_cycles_to_bed(cycle_fn, output_fn):
    """AmpliconSuite ``*_cycles.txt`` -> bed (cycle2bed.py:7-73 with its default arguments, as hsr.py:65 calls it)."""
    all_segs, cycles = {}, {}
    with open(cycle_fn) as fp:
        for line in fp:
            t = line.strip().split()
            if not t:
                continue
            if t[0] == "Segment":
                all_segs[t[1]] = [t[2], int(t[3]), int(t[4])]
            if t[0][:5] == "Cycle":
                cycle_id, weight, segs = 1, 1.0, ['0+', '0-']
                for s in t[0].split(';'):
                    s = s.split('=')
                    if s[0] == "Cycle":
                        cycle_id = s[1]
                    if s[0] == "Copy_count":
                        weight = float(s[1])
                    if s[0] == "Segments":
                        segs = s[1].split(',')
                iscyclic = (segs[0] != "0+" or segs[-1] != "0-")
                cycle = []
                for seg in segs:
                    sid, sdir = seg[:-1], seg[-1]
                    if int(sid) > 0:
                        cur = all_segs[sid]
                        if cycle and cycle[-1][-1] == '+' and sdir == '+' and cycle[-1][0] == cur[0] and cycle[-1][2] + 1 == cur[1]:
                            cycle[-1][2] = cur[2]
                        elif cycle and cycle[-1][-1] == '-' and sdir == '-' and cycle[-1][0] == cur[0] and cycle[-1][1] - 1 == cur[2]:
                            cycle[-1][1] = cur[1]
                        else:
                            cycle.append(cur + [sdir])
                if cycle[-1][-1] == '+' and cycle[0][-1] == '+' and cycle[-1][0] == cycle[0][0] and cycle[-1][2] + 1 == cycle[0][1]:
                    cycle[0][1] = cycle[-1][1]
                    del cycle[-1]
                if cycle[-1][-1] == '-' and cycle[0][-1] == '+' and cycle[-1][0] == cycle[0][0] and cycle[-1][1] - 1 == cycle[0][2]:
                    cycle[0][2] = cycle[-1][2]
                    del cycle[-1]
                cycles[int(cycle_id)] = [iscyclic, weight, cycle]
    print("Creating bed-converted cycles file: " + output_fn)
    with open(output_fn, 'w') as fp:
        fp.write("#chr\tstart\tend\torientation\tcycle_id\tiscyclic\tweight\n")
        for i in range(1, len(cycles) + 1):
            for seg in cycles[i][2]:
                fp.write("%s\t%d\t%d\t%s\t%d\t%s\t%f\n" % (seg[0], seg[1], seg[2], seg[3], i, cycles[i][0], cycles[i][1]))


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
    ecdna, ecdna_ext = [], []
    cycle_fn = args.cycles
    if args.cycles.endswith("_cycles.txt"):
        init_char = "" if args.output_prefix.endswith("/") else "_"
        conv = args.output_prefix + init_char + "converted_" + "cycles.bed"
        convert_cycles_to_bed(args.cycles, conv)
        cycle_fn = conv
    elif not args.cycles.endswith(".bed"):
        sys.stderr.write(args.cycles + "\n")
        sys.stderr.write("Cycles file must be either a valid *_cycles.txt file or a converted .bed file!\n")
        sys.exit(1)
    with open(cycle_fn, 'r') as fp:
        for line in fp:
            if line.startswith("#"):
                continue
            s = line.strip().split()
            ecdna.append([s[0], int(s[1]), int(s[2])])
            ecdna_ext.append([s[0], int(s[1]) - args.bp_match_cutoff, int(s[2]) + args.bp_match_cutoff])
    print("ecDNA intervals:")
    for ival in ecdna:
        print(ival)

    cns_dict = {}
    with open(args.cn_seg, 'r') as fp:
        for line in fp:
            s = line.strip().split()
            if line.startswith('chromosome'):
                continue
            if args.cn_seg.endswith(".cns"):
                cn = 2 * (2 ** float(s[4]))
            elif args.cn_seg.endswith(".bed"):
                cn = float(s[3])
            else:
                sys.stderr.write(args.cn_seg + "\n")
                sys.stderr.write("Invalid cn_seg file format!\n")
            cns_dict.setdefault(s[0], []).append([int(s[1]), int(s[2]), cn])

    if records is None:
        from .bam import decode_bam
        from .records import DeviceRecords
        records = DeviceRecords(decode_bam(args.lr_bam), getattr(args, "device", device))
    dr = records
    T = build_chimeric_table(dr)                              # whole-BAM fetch (hsr.py:21-51) through coral_sa_table
    print("Fetched %d chimeric alignments." % T.n_reads)
    chroms, names = dr.header_chroms, dr.names
    chr_rank = np.array([global_names.chr_idx.get(c, -1) for c in chroms], dtype=np.int64)
    cands = junction_candidates(T, ecdna, chroms, chr_rank)
    res.candidates = _candidate_lists(cands, chroms, names)

    thr = float(args.normal_cov) * 0.5
    sizes, calls = call_breakpoints(cands, thr, args.bp_match_cutoff_clustering, args.bp_match_cutoff, thr, False)
    res.cluster_sizes = sizes
    for head, p1, p2, sup, st in calls:
        bp = [chroms[cands.c1[head]], p1, _ORI[cands.o1[head]], chroms[cands.c2[head]], p2, _ORI[cands.o2[head]],
              (names[cands.read[head]], int(cands.i[head]), int(cands.j[head])), int(cands.gap[head]), int(cands.swapped[head])]
        bpr = [(names[cands.read[k]], int(cands.i[k]), int(cands.j[k])) for k in sup.tolist()]
        res.calls.append([bp, bpr, st])
        hit = -1
        for k, old in enumerate(res.bp_refined):
            if bp[0] == old[0] and bp[3] == old[3] and bp[2] == old[2] and bp[5] == old[5] and \
                    abs(bp[1] - old[1]) <= args.bp_match_cutoff and abs(bp[4] - old[4]) < args.bp_match_cutoff:
                old[-1] |= set(bpr)
                hit = k
                break
        if hit < 0:
            res.bp_refined.append(bp + [bpr])
            res.bp_stats.append(st)
    print("Found %d breakpoints connecting ecDNA and chromosomes." % len(res.bp_refined))

    sum_sizes = sum(chr_sizes.values())
    agg_size = 0
    xtick_pos, starting_pos = [], {}
    for c in chr_sizes.keys():
        agg_size += chr_sizes[c]
        if agg_size < sum_sizes:
            plt.plot([agg_size * 100.0 / sum_sizes, agg_size * 100.0 / sum_sizes], [-1, 1000000], 'k--', linewidth=2)
        xtick_pos.append((agg_size - 0.5 * chr_sizes[c]) * 100.0 / sum_sizes)
        starting_pos[c] = (agg_size - chr_sizes[c]) * 100.0 / sum_sizes

    def on_ecdna(c, p):
        return any(c == iv[0] and p <= iv[2] and iv[1] <= p for iv in ecdna_ext)       # interval_overlap_l([c, p, p], ...) >= 0

    for bp in res.bp_refined:
        on1, on2 = on_ecdna(bp[0], bp[1]), on_ecdna(bp[3], bp[4])
        if on1 and not on2:
            c, p = bp[3], bp[4]
        elif on2 and not on1:
            c, p = bp[0], bp[1]
        else:
            continue
        if c in starting_pos.keys():
            cn = 0.0
            for seg in cns_dict[c]:                           # KeyError for a chromosome without CN rows, as the reference
                if p > seg[0] and p < seg[1]:
                    cn = seg[2]
                    break
            if cn <= 5.0 and len(bp[-1]) <= float(args.normal_cov) * 2.5:
                print("Breakpoint", bp[:6], "Support = ", len(bp[-1]))
                xpos = starting_pos[c] + p * 100.0 / sum_sizes
                ypos = len(bp[-1])
                res.points.append([xpos, ypos])
                plt.plot(xpos, ypos, 'bo')

    plt.xlim([0, 100])
    plt.ylim([1, 500])
    plt.yscale('log')
    plt.xticks(xtick_pos, list(range(1, 23)) + ['X', 'Y'])
    plt.title(args.output_prefix + " integration loci", fontsize=25)
    plt.ylabel('Long read support', fontsize=25)
    plt.tight_layout()
    out_img_name = "integration_sites_" + args.output_prefix
    plt.savefig(out_img_name + '.png')
    plt.close()
    print('\nCreated ' + out_img_name + '.png')
    return res
