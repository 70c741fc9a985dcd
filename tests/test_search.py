"""The native side of the interval search (csrc/coral_search.cpp) on the CPU: the pair FILTER against the reference's own
alignment2bp / alignment2bp_l vectors (the pair table coming from the oracle-made stand-in of the GPU kernel), and one whole
search step against its own single-query form and against real Python sets."""
import json
import os

import numpy as np
import pytest

from coral_amd import synth
from coral_amd.chimeric import ChimericTable, PairSearch
from tests.canon import uncanon_unit
from tests.product_check import install_cpu_kernel_fakes, pair_table_cpu


def table_from_cas(cas, chroms):
    """ChimericTable holding the given (qint, rint, qual) tuples, one read each (rint '-' rows keep ra > rb)."""
    T = ChimericTable()
    tid_of = {c: k for k, c in enumerate(chroms)}
    off, cols = [0], {k: [] for k in ("qs", "qe", "tid", "ra", "rb", "strand", "mapq", "read")}
    for r, (qint, rint, qual) in enumerate(cas):
        for q, ri, mq in zip(qint, rint, qual):
            cols["qs"].append(q[0]); cols["qe"].append(q[1]); cols["tid"].append(tid_of[ri[0]])
            cols["ra"].append(ri[1]); cols["rb"].append(ri[2]); cols["strand"].append(0 if ri[3] == "+" else 1)
            cols["mapq"].append(mq); cols["read"].append(r)
        off.append(len(cols["qs"]))
    T.off = np.array(off, dtype=np.int64)
    for k, v in cols.items():
        setattr(T, k, np.array(v, dtype=np.int64))
    T.name_id = np.arange(len(cas), dtype=np.int64)
    T.failed = np.zeros(len(cas), dtype=bool)
    return T


def search_over(T, pairs, n_tid):
    T.cni0 = np.full(T.n_rows, -1, dtype=np.int64)
    T.cni1 = np.full(T.n_rows, -1, dtype=np.int64)
    T.pairs = pairs
    z = np.zeros(0, dtype=np.int64)
    return PairSearch(T, np.zeros(T.n_reads, dtype=np.int64), z, z, np.zeros(n_tid + 1, dtype=np.int64), z, z)


def as_lists(cands, chroms, name_of):
    return [[chroms[cands.c1[k]], int(cands.p1[k]), "+-"[cands.o1[k]], chroms[cands.c2[k]], int(cands.p2[k]), "+-"[cands.o2[k]],
             (name_of(int(cands.read[k])), int(cands.i[k]), int(cands.j[k])), int(cands.gap[k]), int(cands.swapped[k]),
             int(cands.mqa[k]), int(cands.mqb[k])] for k in range(len(cands))]


def cpu_pairs(T, chroms):
    from coral_amd.global_names import chr_idx
    cols = np.stack([T.qs, T.qe, T.tid, T.ra, T.rb, T.strand, T.mapq])
    return pair_table_cpu(cols, T.off, chroms, np.array([chr_idx.get(c, -1) for c in chroms], dtype=np.int32))


def test_pair_filter_against_reference_vectors(golden_dir):
    with open(os.path.join(golden_dir, "unit_vectors.json")) as fp:
        vec = json.load(fp)
    chroms = synth.CHROMS
    tid_of = {c: k for k, c in enumerate(chroms)}
    cas = [uncanon_unit(v["ca"]) for v in vec["alignment2bp"]]
    T = table_from_cas(cas, chroms)
    S = search_over(T, cpu_pairs(T, chroms), len(chroms))
    n_pos = 0
    for k, v in enumerate(vec["alignment2bp"]):
        i1, i2 = [(tid_of[i[0]], i[1], i[2]) for i in (v["i1"], v["i2"])]
        got = as_lists(S.between([k], i1, i2), chroms, lambda r: "rd%d" % r)
        assert got == uncanon_unit(v["out"]), k
        n_pos += len(got)
    assert n_pos > 30
    cas = [uncanon_unit(v["ca"]) for v in vec["alignment2bp_l"]]
    ivs = [(tid_of[i[0]], i[1], i[2]) for i in vec["alignment2bp_l"][0]["intervals"]]
    T = table_from_cas(cas, chroms)
    S = search_over(T, cpu_pairs(T, chroms), len(chroms))
    got = as_lists(S.within(ivs), chroms, lambda r: "rd%d" % r)
    exp = [c for v in vec["alignment2bp_l"] for c in uncanon_unit(v["out"])]
    assert got == exp and len(exp) > 30
    # a permuted selection keeps the given order; an empty one gives nothing
    sel = [7, 3, 200, 11, 0]
    one_by_one = [c for r in sel for c in as_lists(S.between([r], ivs[0], ivs[1]), chroms, str)]
    assert as_lists(S.between(sel, ivs[0], ivs[1]), chroms, str) == one_by_one
    assert len(S.between([], ivs[0], ivs[1])) == 0


def test_contig_outside_the_reference_list_raises_keyerror():
    """A candidate between contigs that global_names.chr_idx does not list is a KeyError in the reference (bu:293)."""
    chroms = ["chr8", "chrUn_decoy"]
    ca = ([[0, 999], [1000, 1999]], [["chr8", 5000, 5999, "+"], ["chrUn_decoy", 100, 1099, "+"]], [60, 60])
    T = table_from_cas([ca], chroms)
    S = search_over(T, cpu_pairs(T, chroms), len(chroms))
    with pytest.raises(KeyError):
        S.between([0], (0, 0, 10**6), (1, 0, 10**6))
    assert len(S.between([0], (0, 0, 10**6), (0, 0, 10**6))) == 0           # not selected -> never looked at


@pytest.mark.parametrize("name", ["small", "ultra"])
def test_search_step_equals_single_queries(name, tmp_path, monkeypatch):
    """coral_search_step (reach sets, runs, set-iteration order, candidates of every run) equals alignment2bp of its own read
    order run by run, and its runs / orders equal what REAL sets of str give (the product's verification hook)."""
    from coral_amd import infer_breakpoint_graph as ibg
    from coral_amd.records import DeviceRecords
    install_cpu_kernel_fakes(monkeypatch)
    monkeypatch.setattr(ibg, "_VERIFY_SET_ORDER", True)
    cfg, rec = synth.dataset(name, "cpu")
    cn, seeds = str(tmp_path / "cn.bed"), str(tmp_path / "seeds.bed")
    synth.write_cn_bed(cfg, cn)
    synth.write_seed_bed(cfg, seeds)
    b = ibg.bam_to_breakpoint_nanopore(None, seeds, records=DeviceRecords(rec, "cpu"))
    b.read_cns(cn)
    b.fetch()
    b.hash_alignment_to_seg()
    b.find_amplicon_intervals()               # (with the verification hook on) -> the final, merged amplicon intervals
    S = b._search()
    n_cand = 0
    for chrom, s, e, _ in b.amplicon_intervals:
        si, ei = b.pos2cni(chrom, s)[0], b.pos2cni(chrom, e)[0]
        tid = b._tid_of[chrom]
        groups, cands, orders, called = S.step(tid, s, e, si, ei, want_orders=True)
        chroms = b.rec.header_chroms
        b._verify_step(tid, si, ei, [(chroms[int(g[0])], int(g[1]), int(g[2])) for g in groups], orders)
        by = b.cns_intervals_by_chr
        for g, c, order in zip(groups, cands, orders):
            cname = chroms[int(g[0])]
            target = (int(g[0]), by[cname][int(g[1])][1], by[cname][int(g[2])][2])
            want = S.between(order, target, (tid, s, e))
            for f in c.FIELDS:
                assert np.array_equal(getattr(c, f), getattr(want, f)), f
            n_cand += len(c)
        # the calls made inside the step equal coral_call_breakpoints run on the run's candidates afterwards
        for c, got in zip(cands, called):
            want_sizes, want_calls = b._cluster_and_call(c, False)
            assert list(got[0]) == list(want_sizes) and len(got[1]) == len(want_calls)
            for x, y in zip(got[1], want_calls):
                assert x[:3] == y[:3] and x[3].tolist() == y[3].tolist() and list(x[4]) == list(y[4])
                assert [type(v) for v in x[4]] == [type(v) for v in y[4]]
    assert n_cand > 20


def test_rows_by_interval_equals_the_per_interval_loop():
    """The vectorised record-by-interval selection of find_smalldel_breakpoints (ibg:750-766) against the plain loop: same rows,
    same order, records in two intervals twice; overlapping interval lists take the loop itself."""
    import numpy as np
    from coral_amd.infer_breakpoint_graph import rows_by_interval
    rng = np.random.default_rng(4)
    for trial in range(200):
        n = int(rng.integers(0, 60))
        tid = np.sort(rng.integers(0, 3, n))
        pos = rng.integers(0, 5000, n)
        end = pos + rng.integers(1, 1500, n)
        k = int(rng.integers(0, 8))
        ivs = []
        if trial % 3:                      # disjoint per contig, in random list order
            for t in range(3):
                cuts = np.sort(rng.choice(7000, size=2 * int(rng.integers(0, 4)), replace=False))
                ivs += [(t, int(cuts[2 * j]), int(cuts[2 * j + 1])) for j in range(len(cuts) // 2)]
            ivs = [ivs[i] for i in rng.permutation(len(ivs))]
        else:                              # arbitrary, overlapping
            for _ in range(k):
                s = int(rng.integers(0, 6000))
                ivs.append((int(rng.integers(0, 3)), s, s + int(rng.integers(0, 2500))))
        want = [i for (t, s, e) in ivs for i in range(n) if tid[i] == t and pos[i] < e + 1 and end[i] > s]
        assert rows_by_interval(tid, pos, end, ivs).tolist() == want, (trial, ivs)
