"""What the unchanged cycle step touches on the objects the graph build hands over.

  * the BreakpointGraph methods of /root/reference/src/breakpoint_graph.py:609-765 (oracle restatement AND the product class)
    against known answers produced by the reference class itself (tests/golden/graph_methods.json, made by
    tests/golden/make_golden.py graph_methods);
  * a static list of every attribute / method `cycle_decomposition.py` and `path_constraints.py` read on `bb`
    (bam_to_breakpoint_nanopore) and `g` (BreakpointGraph) — cd:46-67, 92-107, 146, 623, 1029, 1500-1555, 1945, 2067;
    pc:48-375; SURVEY.md §8(b) — asserted on the product objects of a real build.
"""
import json
import os

import pytest

from coral_amd import breakpoint_graph as P
from oracle import coral_oracle as O


@pytest.fixture(scope="module")
def vec(golden_dir):
    with open(os.path.join(golden_dir, "graph_methods.json")) as fp:
        return json.load(fp)


def _disc_edges(counts):
    return [["chr8", 1, "+", "chr8", 2, "-", -1, "d", 0.0, c, set(), 0.0] for c in counts]


def _run(fn):
    try:
        return fn()
    except Exception as exc:          # noqa: BLE001 — the golden records the exception type
        return {"raises": type(exc).__name__}


def test_discordant_edge_multiplicities_product(vec):
    n_multi = 0
    for v in vec["discordant_edge_multiplicities"]:
        g = P.BreakpointGraph()
        g.discordant_edges = _disc_edges(v["lr_counts"])
        got = _run(g.infer_discordant_edge_multiplicities)
        assert got == v["out"], v
        n_multi += isinstance(got, list) and max(got, default=1) > 1
    assert n_multi > 50                     # the clustering branch (bg:645-693) is exercised, not only the all-ones exit


def test_discordant_edge_multiplicities_oracle(vec):
    for v in vec["discordant_edge_multiplicities"]:
        assert _run(lambda: O.infer_discordant_edge_multiplicities(_disc_edges(v["lr_counts"]))) == v["out"], v


def test_max_seq_multiplicity(vec):
    seen = set()
    for v in vec["max_seq_multiplicity"]:
        g = P.BreakpointGraph()
        g.sequence_edges = [list(e) for e in v["sequence_edges"]]
        assert g.infer_max_seq_multiplicity(**v["kwargs"]) == v["out"], v
        assert O.infer_max_seq_multiplicity(v["sequence_edges"], **v["kwargs"]) == v["out"], v
        seen.add(v["out"])
    assert len(seen) >= 3


def test_sequence_walks(vec):
    for v in vec["walks"]:
        g = P.BreakpointGraph()
        segs = v["segments"]
        for l, r in segs:
            g.add_node(("chr8", l, "-"))
            g.add_node(("chr8", r, "+"))
            g.add_sequence_edge("chr8", l, r)
        for i in range(len(segs) - 1):
            g.add_concordant_edge("chr8", segs[i][1], "+", "chr8", segs[i + 1][0], "-")
        for nd in v["discordant_nodes"]:
            g.add_discordant_edge(nd[0], nd[1], nd[2], nd[0], nd[1], nd[2])
        for q in v["queries"]:
            l, r = q["pos"]
            c = q["cutoff"]
            assert (g.nextminus("chr8", l, c), g.lastminus("chr8", l, c), g.nextplus("chr8", r, c), g.lastplus("chr8", r, c)) == \
                (q["nextminus"], q["lastminus"], q["nextplus"], q["lastplus"]), (v["segments"], q)


def test_container_maintenance():
    """del_endnode / del_discordant_endnodes / del_discordant_edges / del_source_edges (bg:142-164, :210-253)."""
    g = P.BreakpointGraph()
    for nd in (("chr8", 100, "-"), ("chr8", 199, "+"), ("chr8", 200, "-"), ("chr8", 300, "+")):
        g.add_node(nd)
    g.add_sequence_edge("chr8", 100, 199)
    g.add_sequence_edge("chr8", 200, 300)
    g.add_endnode(("chr8", 100, "-"))
    g.add_endnode(("chr8", 300, "+"))
    with pytest.warns(UserWarning):
        g.add_endnode(("chr8", 100, "-"))
    g.add_discordant_edge("chr8", 199, "+", "chr8", 200, "-", lr_count=4)
    g.add_discordant_edge("chr8", 300, "+", "chr8", 100, "-", lr_count=9)
    g.add_source_edge("chr8", 200, "-")
    g.add_source_edge("chr8", 199, "+")
    assert g.endnodes == {("chr8", 100, "-"): [1], ("chr8", 300, "+"): [1]}
    g.del_source_edges([0], {1: 0})
    assert [e[3:6] for e in g.source_edges] == [["chr8", 199, "+"]]
    assert g.nodes[("chr8", 199, "+")][3] == [0] and g.nodes[("chr8", 200, "-")][3] == []
    g.del_discordant_edges([0], {1: 0})
    assert len(g.discordant_edges) == 1 and g.discordant_edges[0][9] == 9
    assert g.nodes[("chr8", 300, "+")][2] == [0] and g.nodes[("chr8", 199, "+")][2] == []
    assert g.endnodes == {("chr8", 100, "-"): [0], ("chr8", 300, "+"): [0]}
    g.del_discordant_endnodes()
    assert g.endnodes == {}
    with pytest.warns(UserWarning):
        g.del_endnode(("chr8", 100, "-"))


# ---- the attribute surface the downstream steps rely on -----------------------------------------------------------
GRAPH_SURFACE = {            # name -> type; cd:46-67, 92-107, 126-146, 1017-1038, 1500-1527
    "sequence_edges": list, "concordant_edges": list, "discordant_edges": list, "source_edges": list,
    "nodes": dict, "endnodes": dict, "max_cn": float, "amplicon_intervals": list,
}
GRAPH_METHODS = ["infer_discordant_edge_multiplicities", "infer_max_seq_multiplicity", "compute_cn_lr", "sort_edges",
                 "add_node", "add_endnode", "del_endnode", "del_discordant_endnodes", "add_sequence_edge",
                 "add_concordant_edge", "add_discordant_edge", "del_discordant_edges", "add_source_edge",
                 "del_source_edges", "nextminus", "lastminus", "nextplus", "lastplus"]
BUILD_SURFACE = {            # cd:1500-1555, :1945, :2067; ibg:1059-1323; CoRAL.py:29
    "lr_graph": list, "amplicon_intervals": list, "ccid2id": dict, "path_constraints": dict,
    "longest_path_constraints": dict, "cycles": dict, "cycle_weights": dict, "path_constraints_satisfied": dict,
    "chimeric_alignments": dict, "large_indel_alignments": dict, "read_length": dict, "new_bp_list": list,
    "new_bp_stats": list, "new_bp_ccids": list, "amplicon_interval_connections": dict, "normal_cov": float,
    "min_bp_match_cutoff_": int, "min_cluster_cutoff": (int, float), "nm_stats": list, "cns_intervals_by_chr": dict,
}
BUILD_METHODS = ["compute_path_constraints", "closebam", "read_cns", "fetch", "hash_alignment_to_seg",
                 "find_amplicon_intervals", "find_smalldel_breakpoints", "find_breakpoints", "build_graph", "assign_cov",
                 "pos2cni", "addbp"]


def check_surface(b):
    """Shared with the -m gpu twin (tests/test_gpu_e2e.py): the object a real build returns has everything the cycle step
    touches, with the reference's container types and field layouts."""
    import collections.abc
    for name, typ in BUILD_SURFACE.items():
        assert isinstance(getattr(b, name), typ), name
    for name in BUILD_METHODS:
        assert callable(getattr(b, name)), name
    assert callable(b.lr_bamfh.fetch) and callable(b.lr_bamfh.close)
    assert len(b.lr_graph) >= 1
    for g in b.lr_graph:
        for name, typ in GRAPH_SURFACE.items():
            assert isinstance(getattr(g, name), typ), name
        for name in GRAPH_METHODS:
            assert callable(getattr(g, name)), name
        mult = g.infer_discordant_edge_multiplicities()              # cd:146
        assert isinstance(mult, list) and len(mult) == len(g.discordant_edges) and all(isinstance(m, int) for m in mult)
        assert isinstance(g.infer_max_seq_multiplicity(), int)
        for e in g.sequence_edges:                                   # bg:176
            assert len(e) == 9 and isinstance(e[0], str) and isinstance(e[5], int) and isinstance(e[6], int) and \
                isinstance(e[7], int) and isinstance(e[8], float)
        for e in g.concordant_edges:                                 # bg:190; e[9] iterated at ibg:1298
            assert len(e) == 11 and isinstance(e[8], int) and isinstance(e[9], collections.abc.Set) and isinstance(e[10], float)
            assert all(isinstance(rn, str) for rn in e[9])
        for e in g.discordant_edges:                                 # bg:207; e[10] iterated at ibg:1071
            assert len(e) == 12 and isinstance(e[9], int) and isinstance(e[10], collections.abc.Set) and isinstance(e[11], float)
            assert e[9] == len(e[10])
            for r_ in e[10]:
                assert isinstance(r_, tuple) and len(r_) == 3 and isinstance(r_[0], str) and \
                    isinstance(r_[1], int) and isinstance(r_[2], int)
        for nd, adj in g.nodes.items():                              # bg:124; order used at cd:1524-1526
            assert isinstance(nd, tuple) and len(nd) == 3 and len(adj) == 4 and all(isinstance(a, list) for a in adj)
        for nd in g.endnodes:
            assert nd in g.nodes
    for aint in b.amplicon_intervals:                                # cd:1945
        assert len(aint) == 4 and aint[3] in b.ccid2id
    b.compute_path_constraints()                                     # cd:2067
    for ai in range(len(b.lr_graph)):
        pc = b.path_constraints[ai]
        assert len(pc) == 3 and len(pc[0]) == len(pc[1]) == len(pc[2])
    rec = next(iter(b.lr_bamfh.fetch(b.amplicon_intervals[0][0], b.amplicon_intervals[0][1], b.amplicon_intervals[0][2] + 1)))
    for attr in ("query_name", "mapq", "reference_name", "reference_start", "reference_end"):       # ibg:1306-1310
        assert hasattr(rec, attr)
    b.closebam()                                                     # CoRAL.py:29


def test_cycle_step_surface_on_a_real_build(tmp_path, monkeypatch):
    from coral_amd import infer_breakpoint_graph as ibg, synth
    from coral_amd.records import DeviceRecords
    from tests.product_check import install_cpu_kernel_fakes
    install_cpu_kernel_fakes(monkeypatch)
    cfg, rec = synth.dataset("small", "cpu")
    cn, seeds = str(tmp_path / "cn.bed"), str(tmp_path / "seeds.bed")
    synth.write_cn_bed(cfg, cn)
    synth.write_seed_bed(cfg, seeds)
    b = ibg.build_graph_from_records(DeviceRecords(rec, "cpu"), seeds, cn, str(tmp_path / "out"))
    check_surface(b)
