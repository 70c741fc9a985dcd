"""Generate the golden fixtures in this directory from the REAL reference (build container only).

    python tests/golden/make_golden.py            # needs /root/reference, writes tests/golden/*.json

Two kinds of fixture:
  * ``unit_vectors.json`` — known-answer vectors for the pure functions of
    /root/reference/src/cigar_parsing.py and /root/reference/src/breakpoint_utilities.py, obtained by
    importing those modules unmodified and calling them on seeded random inputs;
  * ``e2e_<config>[_variant].json`` — phase-by-phase snapshots (A2..A11 of SURVEY.md §8(a)) and the
    emitted ``*_graph.txt`` / ``*_breakpoints.txt`` text of the reference's graph build run on the
    synthetic records of ``coral_amd.synth.named_config(<config>)`` behind the fake pysam / cvxopt of
    oracle/refharness (see there for what that does and does not pin).

The fixtures are data (inputs are re-generated from the seeded generator and verified against the
stored sha256); nothing of the reference's source text is stored.
"""
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/src"

E2E_CASES = [
    ("tiny", []),
    ("tiny", ["--output_bp"]),
    ("tiny", ["--min_bp_support", "30.0"]),
    ("tiny_edge", []),
    ("small", []),
    ("ultra", []),
    ("tiny", ["--cn_format", "cns"]),           # the .cns flavour of the CN segment file (ibg:94-95)
    ("cfg3_12k", []),                            # BASELINE config 3's layout at 12 000 reads: same graph shape as the full run
    ("cfg3_2amp", []),                           # config 3's layout, two amplicons (two ccids -> two graph files)
]


# the hsr mode (oracle/refharness/run_reference_hsr.py): (data set, --normal_cov)
HSR_CASES = [("tiny", "4"), ("small", "8"), ("ultra", "10"), ("hsr_edge", "4"), ("tiny_edge", "2")]


# coverage track of the plot mode (oracle/refharness/run_reference_plotcov.py): (data set, region)
PLOTCOV_CASES = [("tiny", None), ("tiny", "chr8:72000000-72050000"), ("tiny_edge", "chr8:72090000-72125001"), ("ultra", None)]


def jsonable(o):
    if isinstance(o, (set, frozenset)):
        return {"__set__": sorted((jsonable(x) for x in o), key=lambda v: json.dumps(v))}
    if isinstance(o, tuple):
        return {"__tuple__": [jsonable(x) for x in o]}
    if isinstance(o, list):
        return [jsonable(x) for x in o]
    if isinstance(o, dict):
        return {"__dict__": [[jsonable(k), jsonable(v)] for k, v in o.items()]}
    return o


def unit_vectors():
    sys.path.insert(0, REF_SRC)
    import cigar_parsing as cp
    import breakpoint_utilities as bu
    rnd = random.Random(20241024)
    out = {}

    # ---- cigar2pos* (cp:17-215) through the dispatch dict (cp:219-229)
    v = []
    for pat in cp.cigar2pos_ops:
        for strand in "+-":
            for _ in range(6):
                rl = rnd.randint(2000, 60000)
                nums = [rnd.randint(1, rl // 3) for _ in pat]
                cigar = "".join("%d%s" % (n, c) for n, c in zip(nums, pat))
                v.append(dict(cigar=cigar, strand=strand, read_length=rl,
                              out=list(cp.cigar2pos_ops[pat](cigar, strand, rl))))
    out["cigar2pos"] = v

    # ---- alignment_from_satags (cp:232-269)
    chroms = ["chr7", "chr8", "chr12", "chrX"]
    v = []

    def rand_sa(rl, allow_bad=False):
        pat = rnd.choice(list(cp.cigar2pos_ops))
        if allow_bad and rnd.random() < 0.15:
            pat = rnd.choice(["M", "MD", "MI"])
        nums = [rnd.randint(1, rl // 3) for _ in pat]
        cigar = "".join("%d%s" % (n, c) for n, c in zip(nums, pat))
        return "%s,%d,%s,%s,%d,%d" % (rnd.choice(chroms), rnd.randint(1, 10 ** 8), rnd.choice("+-"), cigar,
                                      rnd.randint(0, 60), rnd.randint(0, 500))
    for k in range(40):
        rl = rnd.randint(3000, 50000)
        sa = [rand_sa(rl, allow_bad=(k % 5 == 4)) for _ in range(rnd.randint(2, 5))]
        v.append(dict(sa_list=sa, read_length=rl, out=jsonable(cp.alignment_from_satags(list(sa), rl))))
    out["alignment_from_satags"] = v

    # ---- interval predicates (bu:11-67)
    v = []
    for _ in range(60):
        a = ["chr8", rnd.randint(0, 1000), 0]; a[2] = a[1] + rnd.randint(-50, 400)
        b = [rnd.choice(["chr8", "chr8", "chr7"]), rnd.randint(0, 1000), 0]; b[2] = b[1] + rnd.randint(-50, 400)
        v.append(dict(a=a, b=b, overlap=bu.interval_overlap(a, b), include=bu.interval_include(a, b),
                      adjacent=bu.interval_adjacent(a, b)))
    out["interval_predicates"] = v
    v = []
    for _ in range(40):
        a = ["chr8", rnd.randint(0, 2000), 0, -1]; a[2] = a[1] + rnd.randint(10, 1500)
        L = []
        for _ in range(rnd.randint(0, 5)):
            s = rnd.randint(0, 3000)
            L.append([rnd.choice(["chr8", "chr8", "chr12"]), s, s + rnd.randint(5, 900), rnd.randint(0, 3)])
        ov, rem = bu.interval_exclusive(a, L)
        v.append(dict(a=a, L=L, overlap_ints=sorted(ov), remaining=rem))
    out["interval_exclusive"] = v

    # ---- interval2bp (bu:289-295)
    def rand_rint():
        c = rnd.choice(chroms[:3]); s = rnd.randint(10 ** 6, 10 ** 7); e = s + rnd.randint(500, 20000)
        return [c, s, e, "+"] if rnd.random() < 0.5 else [c, e, s, "-"]
    v = []
    for _ in range(60):
        r1, r2 = rand_rint(), rand_rint()
        gap = rnd.randint(-200, 300)
        v.append(dict(R1=r1, R2=r2, r=["rd", 1, 2], rgap=gap, out=jsonable(bu.interval2bp(r1, r2, ("rd", 1, 2), gap))))
    out["interval2bp"] = v

    # ---- alignment2bp (bu:70-96) and alignment2bp_l (bu:129-186)
    used = []

    def rand_chimeric(intervals):
        n = rnd.randint(2, 5)
        qint, rint, qual = [], [], []
        del used[:]
        q = 0
        for _ in range(n):
            ln = rnd.randint(400, 6000)
            q0 = q + rnd.randint(-150, 60)
            qint.append([max(0, q0), max(0, q0) + ln])
            q = qint[-1][1] + 1
            iv = rnd.choice(intervals)
            if rnd.random() < 0.8:
                s = rnd.randint(iv[1], max(iv[1], iv[2] - ln - 1)); e = s + ln + rnd.randint(-30, 30)
            else:
                s = rnd.randint(10 ** 6, 10 ** 7); e = s + ln
            rint.append([iv[0], s, e, "+"] if rnd.random() < 0.5 else [iv[0], e, s, "-"])
            qual.append(rnd.choice([60, 60, 60, 60, 60, 30, 15, 5, 0]))
            used.append(iv)
        return (qint, rint, qual)
    v, v2 = [], []
    for k in range(240):
        ivs = [["chr8", 2 * 10 ** 6, 2 * 10 ** 6 + 300000, 0], ["chr8", 4 * 10 ** 6, 4 * 10 ** 6 + 200000, 0],
               ["chr12", 5 * 10 ** 6, 5 * 10 ** 6 + 250000, 1]]
        ca = rand_chimeric(ivs)
        i1, i2 = rnd.choice(ivs), rnd.choice(ivs)
        if k % 3:
            j = rnd.randrange(len(used) - 1)
            i1, i2 = (used[j], used[j + 1]) if k % 2 else (used[j + 1], used[min(j + 2, len(used) - 1)])
        v.append(dict(ca=jsonable(ca), i1=i1, i2=i2,
                      out=jsonable(bu.alignment2bp("rd%d" % k, ca, 100, 20, i1[:3], i2))))
        v2.append(dict(ca=jsonable(ca), intervals=ivs,
                       out=jsonable(bu.alignment2bp_l("rd%d" % k, ca, 100, 20, 100, ivs))))
    out["alignment2bp"] = v
    out["alignment2bp_l"] = v2

    # ---- cluster_bp_list (bu:252-286), bpc2bp (bu:299-388), bp_match (bu:391-416)
    def rand_bp_list(n):
        centres = [(rnd.choice(chroms[:2]), rnd.randint(10 ** 6, 10 ** 7), rnd.choice("+-"),
                    rnd.choice(chroms[:2]), rnd.randint(10 ** 6, 10 ** 7), rnd.choice("+-")) for _ in range(4)]
        L = []
        for i in range(n):
            c = rnd.choice(centres)
            sp = rnd.choice([3, 3, 40, 1500, 2500])
            L.append([c[0], c[1] + rnd.randint(-sp, sp), c[2], c[3], c[4] + rnd.randint(-sp, sp), c[5],
                      ("rd%d" % i, rnd.randint(0, 2), rnd.randint(1, 3)), rnd.randint(-90, 400), rnd.randint(0, 1),
                      rnd.choice([60, 60, 25]), rnd.choice([60, 60, 30])])
        return L
    v, v2 = [], []
    for _ in range(25):
        L = rand_bp_list(rnd.randint(1, 40))
        cl = bu.cluster_bp_list(L, 3, 2000)
        v.append(dict(bp_list=jsonable(L), min_cluster_size=3, cutoff=2000, out=jsonable(cl)))
        for c in cl:
            if len(c) >= 2:
                bp, bpr, st, rem = bu.bpc2bp(c, 100)
                v2.append(dict(cluster=jsonable(c), cutoff=100, bp=jsonable(bp), bpr=jsonable(bpr), stats=st,
                               rest=jsonable(rem)))
    out["cluster_bp_list"] = v
    out["bpc2bp"] = v2
    v = []
    for _ in range(120):
        c = ("chr8", rnd.randint(10 ** 6, 10 ** 6 + 500), rnd.choice("+-"), "chr8", rnd.randint(10 ** 6, 10 ** 6 + 500),
             rnd.choice("+-"))
        b1 = list(c)
        b2 = [c[0], c[1] + rnd.randint(-400, 400), rnd.choice([c[2], c[2], "+"]), c[3], c[4] + rnd.randint(-400, 400), c[5]]
        rgap = rnd.choice([0, -5, 30, 150, 600]) * rnd.choice([1.0, 1.2])
        v.append(dict(bp1=b1, bp2=b2, rgap=rgap, cutoff=[100, 100], out=bool(bu.bp_match(b1, b2, rgap, [100, 100]))))
    out["bp_match"] = v
    return out


def graph_method_vectors():
    """graph_methods.json: known answers of the BreakpointGraph methods the cycle step relies on
    (/root/reference/src/breakpoint_graph.py:609-765), from the reference class itself (imported behind the cvxopt stub:
    these methods do not touch the solver)."""
    sys.path.insert(0, ROOT)
    from oracle.refharness import run_reference as rr
    rr._install_stubs()
    import breakpoint_graph as bg
    rnd = random.Random(20241025)
    out = {}
    v = []
    for k in range(400):
        n = rnd.randint(1, 9)
        style = k % 4
        if style == 0:
            rc = [rnd.randint(1, 60) for _ in range(n)]
        elif style == 1:
            base = rnd.randint(3, 40)
            rc = [max(1, int(base * rnd.choice([1, 1, 1, 2, 2, 3, 4, 5]) * rnd.uniform(0.8, 1.2))) for _ in range(n)]
        elif style == 2:
            rc = [rnd.randint(1, 400) for _ in range(n)]
        else:
            base = rnd.randint(20, 200)
            rc = [max(1, int(base * rnd.choice([1, 1, 2, 3, 6, 7]) * rnd.uniform(0.95, 1.05))) for _ in range(n)]
        g = bg.BreakpointGraph()
        g.discordant_edges = [["chr8", 1, "+", "chr8", 2, "-", -1, "d", 0.0, c, set(), 0.0] for c in rc]
        try:
            res = g.infer_discordant_edge_multiplicities()
        except Exception as exc:                     # noqa: BLE001 — the exception type is the known answer
            res = {"raises": type(exc).__name__}
        v.append(dict(lr_counts=rc, out=res))
    out["discordant_edge_multiplicities"] = v
    v = []
    for k in range(80):
        n = rnd.randint(0, 8)
        edges = []
        for _ in range(n):
            size = rnd.choice([500, 9999, 10000, 25000, 180000, 1200000])
            cn = round(rnd.choice([1.9, 2.0, 4.99, 5.0, 12.5, 40.0, 83.3]) * rnd.uniform(0.9, 1.1), 4)
            edges.append(["chr8", 1000, 1000 + size - 1, -1, "d", 10, 1000, size, cn])
        g = bg.BreakpointGraph()
        g.sequence_edges = edges
        kw = [{}, {"gain": 4.0}, {"size_cutoff": 500, "multiplicity": 3}][k % 3]
        v.append(dict(sequence_edges=edges, kwargs=kw, out=g.infer_max_seq_multiplicity(**kw)))
    out["max_seq_multiplicity"] = v
    # walks along consecutive sequence edges (bg:696-765) on chains with breakpoint edges at random nodes
    v = []
    for k in range(60):
        g = bg.BreakpointGraph()
        pos = 10000
        cuts = [pos]
        for _ in range(rnd.randint(1, 6)):
            pos += rnd.choice([1, 30, 60, 99, 100, 101, 400])
            cuts.append(pos)
        segs = [(cuts[i], cuts[i + 1] - 1) for i in range(len(cuts) - 1)]
        for (l, r) in segs:
            g.add_node(("chr8", l, "-"))
            g.add_node(("chr8", r, "+"))
            g.add_sequence_edge("chr8", l, r)
        for i in range(len(segs) - 1):
            g.add_concordant_edge("chr8", segs[i][1], "+", "chr8", segs[i + 1][0], "-")
        marks = []
        for nd in list(g.nodes):
            if rnd.random() < 0.25:
                g.add_discordant_edge(nd[0], nd[1], nd[2], nd[0], nd[1], nd[2])
                marks.append(list(nd))
        queries = []
        for (l, r) in segs:
            for cutoff in (100, 0):
                queries.append(dict(pos=[l, r], cutoff=cutoff,
                                    nextminus=g.nextminus("chr8", l, cutoff), lastminus=g.lastminus("chr8", l, cutoff),
                                    nextplus=g.nextplus("chr8", r, cutoff), lastplus=g.lastplus("chr8", r, cutoff)))
        v.append(dict(segments=[list(x) for x in segs], discordant_nodes=marks, queries=queries))
    out["walks"] = v
    return out


def main():
    assert os.path.isdir(REF_SRC), "the reference is only available in the build container"
    with open(os.path.join(HERE, "unit_vectors.json"), "w") as fp:
        json.dump(unit_vectors(), fp, separators=(",", ":"))
    graph_methods_only()
    env = dict(os.environ, PYTHONHASHSEED="0")
    for cfg, extra in E2E_CASES:
        tag = cfg + "".join("_" + x.strip("-").replace(".", "p") for x in extra)
        out = os.path.join(HERE, "e2e_%s.json" % tag)
        cmd = [sys.executable, "-m", "oracle.refharness.run_reference", cfg, out] + extra
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, cwd=ROOT, env=env, check=True, stdout=subprocess.DEVNULL)
    hsr_only()
    print("done")


def graph_methods_only():
    with open(os.path.join(HERE, "graph_methods.json"), "w") as fp:
        json.dump(graph_method_vectors(), fp, separators=(",", ":"))


def hsr_only():
    env = dict(os.environ, PYTHONHASHSEED="0")
    for cfg, cov in HSR_CASES:
        out = os.path.join(HERE, "hsr_%s_%s.json" % (cfg, cov))
        cmd = [sys.executable, "-m", "oracle.refharness.run_reference_hsr", cfg, cov, out]
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, cwd=ROOT, env=env, check=True, stdout=subprocess.DEVNULL)


CYCLES_EXAMPLE = """List of cycle segments
Interval\t1\tchr8\t127000000\t129000000
Interval\t2\tchr12\t50000000\t51000000
Segment\t1\tchr8\t127000000\t127499999
Segment\t2\tchr8\t127500000\t127999999
Segment\t3\tchr8\t128400000\t128899999
Segment\t4\tchr12\t50100000\t50299999
Segment\t5\tchr12\t50300000\t50499999
Segment\t6\tchr8\t126000000\t126999999
Cycle=1;Copy_count=12.5;Segments=1+,2+,4-,3+
Cycle=2;Copy_count=3.25;Segments=0+,5-,4-,3+,0-
Cycle=3;Copy_count=2.0;Segments=2+,3-,6+,1+
Cycle=4;Copy_count=1.0;Segments=2-,1-,5+
Cycle=5;Copy_count=7.0;Segments=1+,4+,2-
"""


def cycles_only():
    """cycles_example.json: a hand-written AmpliconSuite cycles file and the bed the reference's cycle2bed makes of it
    (hsr.py:61-66 converts *_cycles.txt inputs this way)."""
    import contextlib
    import io
    import tempfile
    sys.path.insert(0, REF_SRC)
    import cycle2bed
    d = tempfile.mkdtemp()
    a, b = os.path.join(d, "x_cycles.txt"), os.path.join(d, "x.bed")
    with open(a, "w") as fp:
        fp.write(CYCLES_EXAMPLE)
    with contextlib.redirect_stdout(io.StringIO()):
        cycle2bed.convert_cycles_to_bed(a, b)
    with open(os.path.join(HERE, "cycles_example.json"), "w") as fp:
        json.dump({"cycles_txt": CYCLES_EXAMPLE, "bed": open(b).read(),
                   "made_by": "cycle2bed.convert_cycles_to_bed(cycle_fn, output_fn) of the reference, default arguments"}, fp, indent=1)


def plotcov_only():
    for cfg, region in PLOTCOV_CASES:
        out = os.path.join(HERE, "plotcov_%s%s.json" % (cfg, "_region" if region else ""))
        cmd = [sys.executable, "-m", "oracle.refharness.run_reference_plotcov", cfg, out] + ([region] if region else [])
        print(" ".join(cmd), flush=True)
        subprocess.run(cmd, cwd=ROOT, check=True, stdout=subprocess.DEVNULL)


if __name__ == "__main__":
    if sys.argv[1:] == ["hsr"]:
        hsr_only()
    elif sys.argv[1:] == ["plotcov"]:
        plotcov_only()
    elif sys.argv[1:] == ["cycles"]:
        cycles_only()
    elif sys.argv[1:] == ["graph_methods"]:
        graph_methods_only()
    else:
        main()
        plotcov_only()
        cycles_only()
