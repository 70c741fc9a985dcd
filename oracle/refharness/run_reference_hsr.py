"""Run the *real* reference ``hsr`` mode (/root/reference/src/hsr.py, unmodified) on synthetic records -> golden vectors.

TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).  SURVEY.md §8(f) item 3.  pysam is replaced by the
in-memory stand-in of fake_pysam.py; matplotlib is the real one (Agg), with ``pyplot.plot`` wrapped so that the plotted
integration points become part of the golden; ``cluster_bp_list`` / ``bpc2bp`` are wrapped inside the hsr module's
namespace to record the candidate list and every refined breakpoint (the reference returns nothing).

Usage:  PYTHONHASHSEED=0 python -m oracle.refharness.run_reference_hsr <config> <normal_cov> <out_json>
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys
import tempfile
import types

from oracle.refharness.run_reference import REF_SRC, CONDA_SITE, _js, records_digest


def hsr_inputs(config):
    """(cfg, rec, ecDNA intervals): the first half of the segments of the first amplified circle play the ecDNA, so the
    junctions of that circle into its other segments (and every other chimeric read touching them) are the
    "integration" breakpoints the mode looks for."""
    from coral_amd import synth
    cfg, rec = synth.dataset(config, "cpu")
    chroms = rec.header_chroms
    segs = cfg.circles[0]
    half = segs[:max(1, len(segs) // 2)]
    return cfg, rec, [[chroms[sg.tid], int(sg.start), int(sg.end)] for sg in half]


def main():
    config, normal_cov, out_json = sys.argv[1], sys.argv[2], sys.argv[3]
    assert os.environ.get("PYTHONHASHSEED") == "0"
    from coral_amd import synth
    from oracle.refharness import fake_pysam
    cfg, rec, ecdna = hsr_inputs(config)
    sys.modules["pysam"] = fake_pysam
    sys.path.insert(0, REF_SRC)
    tmp = tempfile.mkdtemp(prefix="coral_hsr_")
    bam = os.path.join(tmp, "synthetic.bam")
    fake_pysam.register(bam, fake_pysam.records_to_host(rec))
    cn = os.path.join(tmp, "cn.bed")
    synth.write_cn_bed(cfg, cn)
    cyc = os.path.join(tmp, "ecdna.bed")
    with open(cyc, "w") as fp:
        fp.write("#chr\tstart\tend\torientation\tcycle_id\tiscyclic\tweight\n")
        for c, s, e in ecdna:
            fp.write("%s\t%d\t%d\t+\t1\tTrue\t1.000000\n" % (c, s, e))
    os.chdir(tmp)                               # the reference writes integration_sites_<prefix>.png into the cwd
    import hsr                                   # the reference, unmodified
    import matplotlib.pyplot as plt
    rec_calls = {"candidates": None, "clusters": None, "bpc2bp": [], "points": []}
    real_cluster, real_bpc2bp, real_plot = hsr.cluster_bp_list, hsr.bpc2bp, plt.plot

    def cluster_wrap(bp_list, *a):
        rec_calls["candidates"] = _js(bp_list)
        out = real_cluster(bp_list, *a)
        rec_calls["clusters"] = [len(c) for c in out]
        return out

    def bpc2bp_wrap(cl, *a):
        out = real_bpc2bp(cl, *a)
        rec_calls["bpc2bp"].append(_js([out[0], out[1], out[2], len(out[3])]))
        return out

    def plot_wrap(*a, **k):
        if len(a) == 3 and a[2] == 'bo':
            rec_calls["points"].append([float(a[0]), float(a[1])])
        return real_plot(*a, **k)

    hsr.cluster_bp_list, hsr.bpc2bp, plt.plot = cluster_wrap, bpc2bp_wrap, plot_wrap
    args = types.SimpleNamespace(lr_bam=bam, cycles=cyc, cn_seg=cn, output_prefix="golden", normal_cov=normal_cov,
                                 bp_match_cutoff=100, bp_match_cutoff_clustering=2000)
    buf = io.StringIO()
    raised = None
    with contextlib.redirect_stdout(buf):
        try:
            hsr.locate_hsrs(args)
        except Exception as exc:                # noqa: BLE001 — the error IS the golden (e.g. KeyError for a contig without CN rows)
            raised = [type(exc).__name__, [str(a) for a in exc.args]]
    snap = {"raises": raised,"config": config, "records_sha256": records_digest(rec), "normal_cov": normal_cov, "ecdna": ecdna,
            "stdout": buf.getvalue(), "png_written": os.path.exists(os.path.join(tmp, "integration_sites_golden.png")), **rec_calls}
    with open(out_json, "w") as fp:
        json.dump(snap, fp, separators=(",", ":"))
    sys.stdout.write(buf.getvalue())
    sys.stdout.write("candidates %d  clusters %s  refined calls %d  points %d\n" % (
        len(rec_calls["candidates"] or []), rec_calls["clusters"], len(rec_calls["bpc2bp"]), len(rec_calls["points"])))


if __name__ == "__main__":
    main()
