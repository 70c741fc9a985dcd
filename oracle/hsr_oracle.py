"""CPU restatement of the reference's ``hsr`` mode — TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/hsr.py:21-224 on decoded records (``oracle.hostrecords.HostRecords``) with plain Python
loops; only ``tests/`` may import it.  Pinned by the goldens ``tests/golden/hsr_*.json``, which
oracle/refharness/run_reference_hsr.py produced by running the real ``hsr.locate_hsrs`` (fake pysam, real matplotlib).
The plot itself is not restated: ``points`` are the (x, y) pairs the reference hands to ``plt.plot(x, y, 'bo')``.
"""
from __future__ import annotations

from oracle import coral_oracle as O

CHR_SIZES = {'chr1': 248956422, 'chr2': 242193529, 'chr3': 198295559, 'chr4': 190214555, 'chr5': 181538259,
             'chr6': 170805979, 'chr7': 159345973, 'chr8': 145138636, 'chr9': 138394717, 'chr10': 133797422,
             'chr11': 135086622, 'chr12': 133275309, 'chr13': 114364328, 'chr14': 107043718, 'chr15': 101991189,
             'chr16': 90338345, 'chr17': 83257441, 'chr18': 80373285, 'chr19': 58617616, 'chr20': 64444167,
             'chr21': 46709983, 'chr22': 50818468, 'chrX': 156040895, 'chrY': 57227415}        # gn:20-25 (published hg38 sizes)


def fetch(host):
    """hsr.py:21-51 — like the graph build's fetch, without the NM statistics."""
    read_length, chim = {}, {}
    for i in range(host.n):
        if host.tid[i] < 0:
            continue
        rn = host.names[host.name_id[i]]
        if host.flag[i] < 256 and rn not in read_length:
            read_length[rn] = int(host.qlen[i])
        sa = host.sa_str[i]
        if sa is not None:
            lst = chim.setdefault(rn, [])
            for ent in sa[:-1].split(";"):
                if ent not in lst:
                    lst.append(ent)
    orphans = []
    for rn in chim:
        if rn not in read_length:
            orphans.append(rn)
            continue
        chim[rn] = O.alignment_from_satags(chim[rn], read_length[rn])
    for rn in orphans:
        del chim[rn]
    return read_length, chim


def read_cns(cn_seg):
    """hsr.py:84-108."""
    cns = {}
    with open(cn_seg) as fp:
        for line in fp:
            s = line.strip().split()
            if line.startswith('chromosome'):
                continue
            if cn_seg.endswith(".cns"):
                cn = 2 * (2 ** float(s[4]))
            elif cn_seg.endswith(".bed"):
                cn = float(s[3])
            cns.setdefault(s[0], []).append([int(s[1]), int(s[2]), cn])
    return cns


def candidates(chim, ecdna):
    """hsr.py:116-147 — junctions between a piece outside every ecDNA interval and a piece overlapping one."""
    bp_list = []
    for r, ca in chim.items():
        r_int, rr_int, q_ = ca[0], ca[1], ca[2]
        on_cycle = False
        for iv in ecdna:
            i = O.interval_overlap_l(iv, rr_int)
            if i >= 0 and O.interval_include(rr_int[i], iv):
                on_cycle = True
                break
        if not on_cycle:
            continue
        assigned = [0] * (len(rr_int) - 1)
        for ri in range(len(rr_int) - 1):
            if q_[ri] >= 20 and q_[ri + 1] >= 20:
                a = O.interval_overlap_l(rr_int[ri], ecdna)
                b = O.interval_overlap_l(rr_int[ri + 1], ecdna)
                if (a == -1 and b >= 0) or (a >= 0 and b == -1):
                    bp_list.append(O.interval2bp(rr_int[ri], rr_int[ri + 1], (r, ri, ri + 1),
                                                 int(r_int[ri + 1][0]) - int(r_int[ri][1])) + [q_[ri], q_[ri + 1]])
                    assigned[ri] = 1
        for ri in range(1, len(rr_int) - 1):
            # hsr.py:140 / :146 test ``interval_overlap(rr_int[ri + 1], ecdna_intervals) >= 0`` — interval_overlap (not _l) of
            # an interval with the LIST of intervals compares a chromosome name with a list, is False, and False >= 0 holds:
            # the last piece is not required to lie on the ecDNA.  Both branches of the reference are this same test.
            if assigned[ri - 1] == 0 and assigned[ri] == 0 and q_[ri] < 10 and q_[ri - 1] >= 20 and q_[ri + 1] >= 20 and \
                    O.interval_overlap_l(rr_int[ri - 1], ecdna) == -1:
                bp_list.append(O.interval2bp(rr_int[ri - 1], rr_int[ri + 1], (r, ri - 1, ri + 1),
                                             int(r_int[ri + 1][0]) - int(r_int[ri - 1][1])) + [q_[ri - 1], q_[ri + 1]])
    return bp_list


def refine(bp_list, normal_cov, cutoff, cutoff_clustering):
    """hsr.py:149-170."""
    thr = float(normal_cov) * 0.5
    refined, stats, calls = [], [], []
    clusters = O.cluster_bp_list(bp_list, thr, cutoff_clustering)
    for c in clusters:
        if len(c) >= thr:
            rest = c
            while len(rest) >= thr:
                bp, bpr, st, rest = O.bpc2bp(rest, cutoff)
                calls.append([bp, bpr, st, len(rest)])
                if len(set(bpr)) >= thr:
                    hit = -1
                    for k, old in enumerate(refined):
                        if bp[0] == old[0] and bp[3] == old[3] and bp[2] == old[2] and bp[5] == old[5] and \
                                abs(bp[1] - old[1]) <= cutoff and abs(bp[4] - old[4]) < cutoff:
                            refined[k][-1] |= set(bpr)
                            hit = k
                            break
                    if hit < 0:
                        refined.append(bp + [bpr])
                        stats.append(st)
    return [len(c) for c in clusters], calls, refined, stats


def integration_points(refined, ecdna_ext, cns, normal_cov):
    """hsr.py:185-209 — the printed lines and the plotted (x, y) of breakpoints with exactly one end on the (padded) ecDNA,
    a copy number <= 5 at the other end and a support <= 2.5 x normal coverage.  ``cns[chrom]`` raises KeyError for a
    chromosome without CN rows, as the reference does."""
    total = sum(CHR_SIZES.values())
    start, agg = {}, 0
    for c, sz in CHR_SIZES.items():
        agg += sz
        start[c] = (agg - sz) * 100.0 / total
    lines, points = [], []
    for bp in refined:
        on1 = O.interval_overlap_l([bp[0], bp[1], bp[1]], ecdna_ext) >= 0
        on2 = O.interval_overlap_l([bp[3], bp[4], bp[4]], ecdna_ext) >= 0
        if on1 and not on2:
            c, p = bp[3], bp[4]
        elif on2 and not on1:
            c, p = bp[0], bp[1]
        else:
            continue
        if c in start:
            cn = 0.0
            for seg in cns[c]:
                if p > seg[0] and p < seg[1]:
                    cn = seg[2]
                    break
            if cn <= 5.0 and len(bp[-1]) <= float(normal_cov) * 2.5:
                lines.append("Breakpoint %s Support =  %d" % (bp[:6], len(bp[-1])))
                points.append([start[c] + p * 100.0 / total, len(bp[-1])])
    return lines, points


def locate_hsrs(host, ecdna, cn_seg, normal_cov, bp_match_cutoff=100, bp_match_cutoff_clustering=2000):
    """Everything hsr.locate_hsrs computes, as a dict (stdout text, candidates, bpc2bp calls, plotted points)."""
    ecdna_ext = [[c, s - bp_match_cutoff, e + bp_match_cutoff] for c, s, e in ecdna]
    out = ["ecDNA intervals:"] + [str(iv) for iv in ecdna]
    cns = read_cns(cn_seg)
    _, chim = fetch(host)
    out.append("Fetched %d chimeric alignments." % len(chim))
    bp_list = candidates(chim, ecdna)
    clusters, calls, refined, stats = refine(bp_list, normal_cov, bp_match_cutoff, bp_match_cutoff_clustering)
    out.append("Found %d breakpoints connecting ecDNA and chromosomes." % len(refined))
    lines, points = integration_points(refined, ecdna_ext, cns, normal_cov)
    return dict(stdout_lines=out + lines, candidates=bp_list, clusters=clusters, calls=calls, refined=refined, stats=stats,
                points=points)
