"""Run the *real* reference graph-build path on synthetic records and dump golden vectors.

TEST INFRASTRUCTURE ONLY — runs in the build container (needs /root/reference); its outputs are the
small fixtures under tests/golden/.  One case per process (the reference keeps all state in class
attributes, /root/reference/src/infer_breakpoint_graph.py:22-61) with PYTHONHASHSEED=0
(set-of-str iteration order decides the discordant-edge order).

Usage:  PYTHONHASHSEED=0 python -m oracle.refharness.run_reference <config> <out_json> [--output_bp]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import tempfile
import types

import numpy as np

REF_SRC = "/root/reference/src"
CONDA_SITE = "/opt/conda/lib/python3.9/site-packages"   # pure-python intervaltree 3.1.0 lives here


def _install_stubs():
    from oracle.refharness import fake_pysam, fake_cvxopt
    sys.modules["pysam"] = fake_pysam
    cv = types.ModuleType("cvxopt")
    for k in ("matrix", "mul", "log", "spdiag", "solvers", "modeling"):
        setattr(cv, k, getattr(fake_cvxopt, k))
    sys.modules["cvxopt"] = cv
    sys.modules["cvxopt.modeling"] = fake_cvxopt.modeling
    sys.modules["cvxopt.solvers"] = fake_cvxopt.solvers
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    if CONDA_SITE not in sys.path:
        sys.path.append(CONDA_SITE)


def _js(o):
    """JSON-able deep copy: sets -> sorted lists (tagged), tuples -> lists, numpy scalars -> python."""
    if isinstance(o, (set, frozenset)):
        return {"__set__": sorted((_js(x) for x in o), key=lambda v: json.dumps(v))}
    if isinstance(o, dict):
        return {"__dict__": [[_js(k), _js(v)] for k, v in o.items()]}
    if isinstance(o, (list, tuple)):
        return [_js(x) for x in o]
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    if isinstance(o, (bool, int, float, str)) or o is None:
        return o
    if hasattr(o, "data") and hasattr(o, "begin"):     # intervaltree Interval
        return [o.begin, o.end, o.data]
    return repr(o)


def records_digest(rec) -> str:
    h = hashlib.sha256()
    for k in ("tid", "pos", "end", "flag", "mapq", "qlen", "has_seq", "nm", "name_id", "n_cigar", "cigar_off",
              "cigar", "sa_off", "sa", "sa_nm", "nonacgt_rec", "nonacgt_pos", "name_gid"):
        h.update(getattr(rec, k).cpu().numpy().tobytes())
    return h.hexdigest()


def graph_snapshot(g):
    return dict(sequence_edges=_js(g.sequence_edges), concordant_edges=_js(g.concordant_edges),
                discordant_edges=_js(g.discordant_edges), source_edges=_js(g.source_edges),
                nodes=[[list(k), _js(v)] for k, v in g.nodes.items()],
                endnodes=[[list(k), _js(v)] for k, v in g.endnodes.items()],
                amplicon_intervals=_js(g.amplicon_intervals), max_cn=g.max_cn)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("out_json")
    ap.add_argument("--output_bp", action="store_true")
    ap.add_argument("--min_bp_support", type=float, default=1.0)
    ap.add_argument("--cn_format", default="bed", choices=["bed", "cns"],
                    help="flavour of the CN segment file (ibg:94-98): .bed (CN in column 4) or .cns (log2 ratio in column 5)")
    a = ap.parse_args()
    assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0"

    from coral_amd import synth
    from oracle.refharness import fake_pysam
    cfg, rec = synth.dataset(a.config, "cpu")
    _install_stubs()
    tmp = tempfile.mkdtemp(prefix="coral_ref_")
    bam = os.path.join(tmp, "synthetic.bam")
    fake_pysam.register(bam, fake_pysam.records_to_host(rec))
    cn = os.path.join(tmp, "cn." + a.cn_format)
    seedf = os.path.join(tmp, "seeds.bed")
    (synth.write_cn_bed if a.cn_format == "bed" else synth.write_cn_cns)(cfg, cn)
    synth.write_seed_bed(cfg, seedf)
    prefix = os.path.join(tmp, "out")

    import infer_breakpoint_graph as ibg        # the reference, unmodified
    import breakpoint_graph as bg
    import global_names
    import logging, time
    global_names.TSTART = time.time()
    logging.basicConfig(filename=os.path.join(tmp, "ref.log"), filemode="w", level=logging.DEBUG)

    snap = {"config": a.config, "records_sha256": records_digest(rec), "n_records": rec.n,
            "python_hash_seed": 0, "min_bp_support": a.min_bp_support, "cn_format": a.cn_format}
    # same call sequence as reconstruct_graph (/root/reference/src/infer_breakpoint_graph.py:1349-1394)
    b = ibg.bam_to_breakpoint_nanopore(bam, seedf)
    b.min_bp_cov_factor = a.min_bp_support
    b.read_cns(cn)
    snap["A2"] = dict(normal_cov=b.normal_cov, min_cluster_cutoff=b.min_cluster_cutoff,
                      n_cns=len(b.cns_intervals))
    b.fetch()
    snap["A3"] = dict(n_read_length=len(b.read_length), nm_stats=_js(b.nm_stats),
                      chimeric_alignments=_js(b.chimeric_alignments))
    b.hash_alignment_to_seg()
    snap["A4"] = dict(chimeric_alignments=_js(b.chimeric_alignments),
                      chimeric_alignments_seg=_js(b.chimeric_alignments_seg))
    b.find_amplicon_intervals()
    snap["A5"] = dict(amplicon_intervals=_js(b.amplicon_intervals),
                      amplicon_interval_connections=_js(b.amplicon_interval_connections),
                      new_bp_list=_js(b.new_bp_list), new_bp_stats=_js(b.new_bp_stats), new_bp_ccids=_js(b.new_bp_ccids))
    b.find_smalldel_breakpoints()
    snap["A6"] = dict(large_indel_alignments=_js(b.large_indel_alignments), new_bp_list=_js(b.new_bp_list),
                      new_bp_ccids=_js(b.new_bp_ccids),
                      amplicon_interval_connections=_js(b.amplicon_interval_connections))
    b.find_breakpoints()
    snap["A7"] = dict(new_bp_list=_js(b.new_bp_list), new_bp_stats=_js(b.new_bp_stats), new_bp_ccids=_js(b.new_bp_ccids),
                      amplicon_interval_connections=_js(b.amplicon_interval_connections))
    b.build_graph()
    snap["A9"] = dict(ccid2id=_js(b.ccid2id), graphs=[graph_snapshot(g) for g in b.lr_graph])
    files = {}
    if a.output_bp:
        for gi in range(len(b.lr_graph)):
            bp_stats_i = []
            for de in b.lr_graph[gi].discordant_edges:
                for bpi in range(len(b.new_bp_list)):
                    bp_ = b.new_bp_list[bpi]
                    if de[:6] == bp_[:6]:
                        bp_stats_i.append(b.new_bp_stats[bpi])
                        break
            fn = prefix + "_amplicon" + str(gi + 1) + "_breakpoints.txt"
            bg.output_breakpoint_info_lr(b.lr_graph[gi], fn, bp_stats_i)
            files[os.path.basename(fn)] = open(fn).read()
    else:
        b.assign_cov()
        snap["A10"] = dict(graphs=[graph_snapshot(g) for g in b.lr_graph])
        for gi in range(len(b.lr_graph)):
            b.lr_graph[gi].compute_cn_lr(b.normal_cov)
        snap["A11"] = dict(graphs=[graph_snapshot(g) for g in b.lr_graph],
                           cn_solver="oracle/refharness/fake_cvxopt.py (NOT cvxopt; CN parity vs cvxopt unpinned)")
        # what the cycle step asks of every graph it is handed (cd:146, :623, :1029; companion bg:609-627)
        snap["A11x"] = dict(discordant_edge_multiplicities=[_js(g.infer_discordant_edge_multiplicities()) for g in b.lr_graph],
                            max_seq_multiplicity=[_js(g.infer_max_seq_multiplicity()) for g in b.lr_graph])
        for gi in range(len(b.lr_graph)):
            fn = prefix + "_amplicon" + str(gi + 1) + "_graph.txt"
            bg.output_breakpoint_graph_lr(b.lr_graph[gi], fn)
            files[os.path.basename(fn)] = open(fn).read()
    snap["files"] = files
    if not a.output_bp:
        # the step right after the graph build (ibg:1059-1323, called from cd:2067)
        b.compute_path_constraints()
        snap["F2"] = dict(path_constraints=_js(b.path_constraints))
    with open(a.out_json, "w") as fp:
        json.dump(snap, fp, indent=None, separators=(",", ":"))
    for k, v in files.items():
        sys.stdout.write("==== %s\n%s" % (k, v))


if __name__ == "__main__":
    main()
