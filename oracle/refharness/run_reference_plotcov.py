"""Coverage track of the reference's ``plot`` mode (/root/reference/src/plot_amplicons.py:376-411) -> golden vectors.

TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).  SURVEY.md §8(f) item 4.  The real
``graph_vis.plot_graph`` runs on the ``*_graph.txt`` text stored in the e2e golden of the same data set, behind the fake
pysam; every ``count_coverage(chrom, start, stop)`` call it makes for the track is recorded (in order) together with the
height of the silver ``Rectangle`` drawn for it (``Rectangle`` is wrapped in the module's namespace).  Genes are hidden (no
annotation files are read).  The fixture stores the calls run-length grouped: [chrom, first start, window, [base totals]].

Usage:  python -m oracle.refharness.run_reference_plotcov <config> <out_json> [chrom:start-end]
"""
from __future__ import annotations

import json
import os
import sys
import tempfile

from oracle.refharness.run_reference import REF_SRC, CONDA_SITE, records_digest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(HERE)), "tests", "golden")


def main():
    config, out_json = sys.argv[1], sys.argv[2]
    region = sys.argv[3] if len(sys.argv) > 3 else None
    from coral_amd import synth
    from oracle.refharness import fake_pysam
    cfg, rec = synth.dataset(config, "cpu")
    with open(os.path.join(GOLDEN, "e2e_%s.json" % config)) as fp:
        e2e = json.load(fp)
    assert e2e["records_sha256"] == records_digest(rec)
    sys.modules["pysam"] = fake_pysam
    sys.path.insert(0, REF_SRC)
    sys.path.append(CONDA_SITE)
    tmp = tempfile.mkdtemp(prefix="coral_plot_")
    bam = os.path.join(tmp, "synthetic.bam")
    fake_pysam.register(bam, fake_pysam.records_to_host(rec))
    graph_fn = os.path.join(tmp, "amplicon1_graph.txt")
    name = sorted(k for k in e2e["files"] if k.endswith("_graph.txt"))[0]
    with open(graph_fn, "w") as fp:
        fp.write(e2e["files"][name])
    import plot_amplicons as pa                  # the reference, unmodified
    rects = []
    real_rect = pa.Rectangle

    def rect_wrap(xy, w, h, **k):
        if k.get("color") == "silver":
            rects.append([float(xy[0]), float(w), float(h)])
        return real_rect(xy, w, h, **k)

    pa.Rectangle = rect_wrap
    calls = []
    real_cc = fake_pysam.AlignmentFile.count_coverage

    def cc_wrap(self, contig, start=None, stop=None, **k):
        out = real_cc(self, contig, start, stop, **k)
        calls.append([contig, int(start), int(stop), int(sum(sum(a) for a in out))])
        return out

    fake_pysam.AlignmentFile.count_coverage = cc_wrap
    g = pa.graph_vis()
    g.open_bam(bam)
    g.parse_graph_file(graph_fn)
    if region:
        pchrom = region.split(':')[0]
        pb1, pb2 = region.split(':')[1].rsplit('-')
        g.plot_bounds = (pchrom, int(pb1), int(pb2))
    g.graph_amplified_intervals()
    g.plot_graph("golden", os.path.join(tmp, "golden_graph"), hide_genes=True)
    assert len(calls) == len(rects)
    for (c, a, b, tot), (x, w, h) in zip(calls, rects):
        assert h == tot * 1.0 / (b - a)                      # plot:399-400: the height IS the windowed coverage
    tracks = []
    for c, a, b, tot in calls:
        t = tracks[-1] if tracks else None
        if t and t[0] == c and t[2] == b - a and t[1] + t[2] * len(t[3]) == a:
            t[3].append(tot)
        else:
            tracks.append([c, a, b - a, [tot]])
    snap = {"config": config, "records_sha256": e2e["records_sha256"], "graph_file": name, "region": region,
            "intervals_from_graph": {c: v for c, v in g.intervals_from_graph.items()}, "n_windows": len(calls), "tracks": tracks}
    with open(out_json, "w") as fp:
        json.dump(snap, fp, separators=(",", ":"))
    print("intervals", snap["intervals_from_graph"], "windows", len(calls), "runs", [(t[0], t[1], t[2], len(t[3])) for t in tracks])


if __name__ == "__main__":
    main()
