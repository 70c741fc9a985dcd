"""GPU parity: every kernel behind the C ABI vs the CPU oracle's per-record primitives (bit-exact integers)."""
import numpy as np
import pytest
import torch

from coral_amd import synth

pytestmark = pytest.mark.gpu


def _odd_records():
    M, I, D, N, S, H, P, EQ, X = range(9)
    alns = [
        dict(tid=0, pos=100, cigar=[(S, 5), (M, 50), (D, 700), (M, 20), (I, 3), (M, 10)], name="a"),
        dict(tid=0, pos=120, cigar=[(D, 4), (M, 30), (N, 900), (EQ, 10), (X, 2), (EQ, 5), (H, 7)], name="b"),   # leading D, N gap, =/X
        dict(tid=0, pos=130, cigar=[], flag=4, name="c", has_seq=1, qlen=40),                                   # placed unmapped, no CIGAR
        dict(tid=0, pos=140, cigar=[(S, 20), (I, 5)], name="d"),                                              # no aligned block at all
        dict(tid=0, pos=150, cigar=[(M, 10), (D, 300), (D, 301), (M, 10)], name="e"),                         # adjacent D ops sum to 601
        dict(tid=0, pos=160, cigar=[(M, 10), (D, 300), (I, 2), (D, 300), (M, 10)], name="f"),                 # 600 exactly: not a gap
        dict(tid=0, pos=170, cigar=[(M, 10), (D, 800), (M, 10)], mapq=19, name="g"),                          # low MAPQ: no gap rows
        dict(tid=0, pos=180, cigar=[(M, 200)], has_seq=0, flag=256, name="a"),                                # SEQ-less secondary
        dict(tid=0, pos=190, cigar=[(M, 5), (P, 2), (M, 5)] + [(I, 1), (M, 3)] * 200, name="h", nonacgt=[191, 193]),  # > 256 ops
        dict(tid=2, pos=5, cigar=[(M, 1000), (D, 601), (M, 1000), (D, 5000), (M, 10)], name="i"),
        dict(tid=2, pos=900, cigar=[(M, 4)], name="j"),
    ]
    return synth.records_from_alignments(alns)


def _adversarial_records(seed=5, n=160):
    """Random CIGARs built to sit on the gap filter's edges: D/N runs whose sums hover around min_gap / 2 and min_gap,
    runs of non-aligned ops longer than one and two lanes (4 and 8 ops), placed across quad, chunk (256 ops) and batch
    (2048 ops) boundaries, records that end in D/N/S, and records without any aligned op."""
    M, I, D, N, S, H, P, EQ, X = range(9)
    rng = np.random.default_rng(seed)
    alns, pos = [], 1000
    for r in range(n):
        ops = []
        target = int(rng.choice([3, 9, 250, 262, 600, 2040, 2050, 2300, 4100]))
        if rng.random() < 0.3:
            ops.append((S, int(rng.integers(1, 50))))
        while len(ops) < target:
            kind = rng.random()
            if kind < 0.55:
                ops.append((int(rng.choice([M, EQ, X])), int(rng.integers(1, 40))))
            elif kind < 0.75:
                ops.append((int(rng.choice([D, N])), int(rng.choice([1, 2, 149, 150, 151, 299, 300, 301, 302, 599, 600, 601, 1200]))))
            elif kind < 0.9:
                ops.append((int(rng.choice([I, P])), int(rng.integers(1, 6))))
            else:                                   # a run of non-aligned ops, up to three lanes long
                for _ in range(int(rng.integers(2, 13))):
                    ops.append((int(rng.choice([D, N, I, P])), int(rng.choice([1, 100, 150, 200, 299, 300, 301, 602]))))
        tail = rng.random()
        if tail < 0.3:
            ops.append((int(rng.choice([S, H])), int(rng.integers(1, 30))))
        elif tail < 0.45:
            ops.append((D, int(rng.choice([5, 700]))))
        if r % 23 == 7:
            ops = [(S, 10), (I, 4), (D, 700), (I, 2)]      # no aligned block at all
        # merge nothing: adjacent equal ops are legal in BAM and must not be merged by us
        alns.append(dict(tid=0, pos=pos, cigar=ops, name="adv%d" % r, mapq=int(rng.choice([60, 60, 60, 19]))))
        pos += int(rng.integers(1, 50))
    return synth.records_from_alignments(alns)


def _cases():
    yield "odd", _odd_records()
    yield "adversarial", _adversarial_records()
    for name in ("tiny", "small", "ultra"):
        yield name, synth.generate(synth.named_config(name), "cpu")
    yield "cfg1_8k", synth.generate(synth.scaled_config("cfg1", 8000), "cpu")


@pytest.fixture(scope="module", params=["odd", "adversarial", "tiny", "small", "ultra", "cfg1_8k"])
def case(request):
    from coral_amd.records import DeviceRecords
    from oracle.hostrecords import HostRecords
    rec = dict(_cases())[request.param]
    return request.param, rec, HostRecords(rec), DeviceRecords(rec, "cuda:0")


def test_cigar_scan(case):
    """coral_cigar_scan (record groups from a work cursor, register ring, record boundaries as lane masks) against the oracle's
    per-record blocks: sums, first / last block, every large gap in the reference's order."""
    from coral_amd import kernels
    name, rec, host, dr = case
    res = kernels.cigar_scan(dr, 600, 20, gap_cap=64)       # small cap: exercises the overflow/retry path
    mb, qi = res.mbases.cpu().numpy(), res.qinfer.cpu().numpy()
    b0, b1 = res.blk_first.cpu().numpy(), res.blk_last.cpu().numpy()
    gaps = []
    for i in range(host.n):
        bl = host.blocks(i)
        assert mb[i] == sum(e - s for s, e in bl), (name, i)
        assert qi[i] == (host.infer_read_length(i) or 0), (name, i)
        assert (b0[i], b1[i]) == ((bl[0][0], bl[-1][1]) if bl else (-1, -1)), (name, i)
        if host.mapq[i] >= 20:
            for k in range(len(bl) - 1):
                if abs(bl[k + 1][0] - bl[k][1]) > 600:
                    gaps.append((i, bl[k][1], bl[k + 1][0]))
    got = [(int(g[0]), int(g[2]), int(g[3])) for g in res.gaps]
    for g in res.gaps:
        bl = host.blocks(int(g[0]))
        assert (int(g[4]), int(g[5])) == (bl[0][0], bl[-1][1])
    assert got == gaps, name
    if name == "odd":
        assert len(gaps) == 5     # a, b(N), e, i x2


@pytest.mark.parametrize("min_gap", [0, 1, 3, 299, 300, 301, 601, 1199, 5000])
def test_cigar_scan_gap_filter_thresholds(min_gap):
    """The conservative gap filter against the oracle's blocks for thresholds around the D/N lengths of the adversarial set
    (odd and even: the filter flags a lane at G > min_gap // 2)."""
    from coral_amd import kernels
    from coral_amd.records import DeviceRecords
    from oracle.hostrecords import HostRecords
    rec = _adversarial_records(seed=17 + min_gap, n=120)
    host, dr = HostRecords(rec), DeviceRecords(rec, "cuda:0")
    res = kernels.cigar_scan(dr, min_gap, 20)
    want = []
    for i in range(host.n):
        bl = host.blocks(i)
        if host.mapq[i] >= 20:
            want += [(i, bl[k][1], bl[k + 1][0]) for k in range(len(bl) - 1) if bl[k + 1][0] - bl[k][1] > min_gap]
        assert int(res.blk_first[i]) == (bl[0][0] if bl else -1) and int(res.blk_last[i]) == (bl[-1][1] if bl else -1), i
        assert int(res.mbases[i]) == sum(e - s for s, e in bl)
    assert [(int(g[0]), int(g[2]), int(g[3])) for g in res.gaps] == want
    assert len(want) > 50 or min_gap >= 1199


def test_cigar_scan_record_shapes():
    """Records the flat stream has to cut correctly: many records inside one 1 KiB chunk, records without any op between
    records with ops, single-quad records, a record of exactly 64 / 128 quads (chunk-aligned ends), one far longer than the
    ring (> 8 KiB of ops), all next to each other and at both ends of the file."""
    from coral_amd import kernels
    from coral_amd.records import DeviceRecords
    from oracle.hostrecords import HostRecords
    M, I, D, N, S, H, P, EQ, X = range(9)
    rng = np.random.default_rng(3)
    alns, pos = [], 50

    def add(ops, **kw):
        nonlocal pos
        alns.append(dict(tid=0, pos=pos, cigar=ops, name="s%d" % len(alns), **kw))
        pos += 7
    add([], flag=4, has_seq=1, qlen=30)                                      # file starts with an op-less record
    for n_ops in (1, 2, 3, 4, 5, 8, 1, 1, 1, 255, 256, 257, 511, 512, 513, 3, 4):
        add([((M, D)[k % 2], 1 + int(rng.integers(0, 700 if k % 2 else 9))) for k in range(n_ops)])
    add([], flag=4)
    add([], flag=4)
    add([(M, 3), (D, 900), (M, 3)] * 1200)                                   # 3600 ops = 14 KiB: longer than the ring, many gaps
    for _ in range(40):                                                      # a burst of tiny records: dozens per chunk
        add([(M, int(rng.integers(1, 20)))] + ([(D, 650), (M, 2)] if rng.random() < 0.3 else []))
    add([], flag=4)                                                          # and ends with one
    rec = synth.records_from_alignments(alns)
    host, dr = HostRecords(rec), DeviceRecords(rec, "cuda:0")
    res = kernels.cigar_scan(dr, 600, 20)
    want = []
    for i in range(host.n):
        bl = host.blocks(i)
        assert int(res.mbases[i]) == sum(e - s for s, e in bl), i
        assert int(res.qinfer[i]) == (host.infer_read_length(i) or 0), i
        assert (int(res.blk_first[i]), int(res.blk_last[i])) == ((bl[0][0], bl[-1][1]) if bl else (-1, -1)), i
        want += [(i, bl[k][1], bl[k + 1][0]) for k in range(len(bl) - 1) if bl[k + 1][0] - bl[k][1] > 600]
    assert [(int(g[0]), int(g[2]), int(g[3])) for g in res.gaps] == want and len(want) > 1200


def _random_segments(host, rng, n):
    segs = []
    tids = np.unique(host.tid[host.tid >= 0])
    for _ in range(n):
        t = int(rng.choice(tids))
        sel = host.tid == t
        lo, hi = int(host.pos[sel].min()), int(host.end[sel].max())
        s = int(rng.integers(lo - 50, hi))
        e = s + int(rng.choice([1, 2, 17, 300, 5000, 200000]))
        segs.append((t, s, e))
    return segs


def test_segment_coverage(case):
    from coral_amd import kernels
    name, rec, host, dr = case
    rng = np.random.default_rng(5)
    res = kernels.cigar_scan(dr)
    segs = _random_segments(host, rng, 60)
    # plus a tiling (shared boundaries, as sequence edges have) and an all-covering segment
    t0 = int(host.tid[0])
    lo, hi = int(host.pos[host.tid == t0].min()), int(host.end[host.tid == t0].max())
    cuts = np.unique(np.concatenate([[lo - 10, hi + 10], rng.integers(lo, hi, 12)]))
    segs += [(t0, int(a), int(b)) for a, b in zip(cuts[:-1], cuts[1:])]
    segs.append((t0, lo - 10, hi + 10))
    n_reads, n_bases = kernels.segment_coverage(dr, res, segs)
    for j, (t, s, e) in enumerate(segs):
        c = host.chroms[t]
        exp_reads = sum(1 for i in host.region(c, s, e) if host.infer_read_length(i))
        assert n_reads[j] == exp_reads, (name, j, segs[j])
        assert n_bases[j] == host.count_coverage_sum(c, s, e), (name, j, segs[j])


def test_point_cover(case):
    from coral_amd import kernels
    name, rec, host, dr = case
    rng = np.random.default_rng(6)
    pts = []
    for (t, s, e) in _random_segments(host, rng, 40):
        pts += [(t, s), (t, s + 1), (t, s - 101), (t, s + 101)]
    pts.append(pts[0])                       # duplicate query point
    got = kernels.point_cover(dr, pts, pair_cap=128)
    for (t, p), g in zip(pts, got):
        exp = host.region(host.chroms[t], p, p + 1)
        assert np.array_equal(g, exp), (name, t, p)


def test_empty_inputs():
    from coral_amd import kernels
    from coral_amd.records import DeviceRecords
    rec = synth.records_from_alignments([])
    dr = DeviceRecords(rec, "cuda:0")
    res = kernels.cigar_scan(dr)
    assert res.gaps.shape[0] == 0 and res.mbases.numel() == 0
    nr, nb = kernels.segment_coverage(dr, res, [(0, 0, 100)])
    assert nr.tolist() == [0] and nb.tolist() == [0]
    assert [g.tolist() for g in kernels.point_cover(dr, [(0, 5)])] == [[]]


def _table_from_cas(cas, chroms):
    """ChimericTable holding the given (qint, rint, qual) tuples, one read each (rint '-' rows keep ra > rb)."""
    from coral_amd.chimeric import ChimericTable
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


def _search_over(T, pairs, n_tid):
    """PairSearch over a hand-made table: only the pair filter is exercised (no CN segments, empty inverted index)."""
    from coral_amd.chimeric import PairSearch
    T.cni0 = np.full(T.n_rows, -1, dtype=np.int64)
    T.cni1 = np.full(T.n_rows, -1, dtype=np.int64)
    T.pairs = pairs
    z = np.zeros(0, dtype=np.int64)
    return PairSearch(T, np.zeros(T.n_reads, dtype=np.int64), z, z, np.zeros(n_tid + 1, dtype=np.int64), z, z)


def _gpu_pairs(dr, T):
    import torch
    from coral_amd import kernels
    rows = np.stack([T.qs, T.qe, T.tid, T.ra, T.rb, T.strand, T.mapq, np.zeros_like(T.qs)], axis=1).astype(np.int32)
    out = kernels.pair_table(dr, torch.from_numpy(T.off.astype(np.int32)).to(dr.device),
                             torch.from_numpy(np.ascontiguousarray(rows)).to(dr.device), T.n_reads, T.n_rows)
    return out[:2 * T.n_rows].cpu().numpy()


def candidates_as_lists(cands, chroms, name_of):
    return [[chroms[cands.c1[k]], int(cands.p1[k]), "+-"[cands.o1[k]], chroms[cands.c2[k]], int(cands.p2[k]), "+-"[cands.o2[k]],
             (name_of(int(cands.read[k])), int(cands.i[k]), int(cands.j[k])), int(cands.gap[k]), int(cands.swapped[k]),
             int(cands.mqa[k]), int(cands.mqb[k])] for k in range(len(cands))]


def test_bp_pair_table_against_reference_vectors(golden_dir):
    """coral_bp_pair_table (K4, GPU) + the native pair filter vs the known-answer vectors of the REFERENCE's alignment2bp /
    alignment2bp_l (tests/golden/unit_vectors.json): 240 single-read queries and one all-reads query."""
    import json, os
    from coral_amd.records import DeviceRecords
    from tests.canon import uncanon_unit
    with open(os.path.join(golden_dir, "unit_vectors.json")) as fp:
        vec = json.load(fp)
    chroms = synth.CHROMS
    dr = DeviceRecords(synth.records_from_alignments([]), "cuda:0")
    tid_of = {c: k for k, c in enumerate(chroms)}
    cas = [uncanon_unit(v["ca"]) for v in vec["alignment2bp"]]
    T = _table_from_cas(cas, chroms)
    S = _search_over(T, _gpu_pairs(dr, T), len(chroms))
    n_pos = 0
    for k, v in enumerate(vec["alignment2bp"]):
        i1, i2 = [(tid_of[i[0]], i[1], i[2]) for i in (v["i1"], v["i2"])]
        got = candidates_as_lists(S.between([k], i1, i2), chroms, lambda r: "rd%d" % r)
        assert got == uncanon_unit(v["out"]), k
        n_pos += len(got)
    assert n_pos > 30
    cas = [uncanon_unit(v["ca"]) for v in vec["alignment2bp_l"]]
    ivs = [(tid_of[i[0]], i[1], i[2]) for i in vec["alignment2bp_l"][0]["intervals"]]
    T = _table_from_cas(cas, chroms)
    S = _search_over(T, _gpu_pairs(dr, T), len(chroms))
    got = candidates_as_lists(S.within(ivs), chroms, lambda r: "rd%d" % r)
    exp = [c for v in vec["alignment2bp_l"] for c in uncanon_unit(v["out"])]
    assert got == exp and len(exp) > 30


def test_bp_pair_table_equals_cpu_stand_in():
    """The GPU pair table equals, bit for bit, the oracle-made table the CPU host-logic tests run on (tests/product_check.py),
    on real chimeric tables incl. reads with many alignments and low-MAPQ middles."""
    from coral_amd.chimeric import build_chimeric_table
    from coral_amd.records import DeviceRecords
    from tests.product_check import pair_table_cpu
    for name in ("tiny_edge", "ultra", "small"):
        cfg, rec = synth.dataset(name, "cpu")
        dr = DeviceRecords(rec, "cuda:0")
        T = build_chimeric_table(dr)
        cols = np.stack([T.qs, T.qe, T.tid, T.ra, T.rb, T.strand, T.mapq])
        want = pair_table_cpu(cols, T.off, dr.header_chroms, dr.chr_rank)
        assert T.pairs.shape == want.shape and np.array_equal(T.pairs, want), name
        assert (T.pairs[:, 5] & 2).sum() > 50


def test_reads_with_more_than_64_alignments():
    """No per-read limit: a read with 90 distinct SA entries goes through coral_sa_table and the pair table (the reference has
    no limit either)."""
    from coral_amd.chimeric import build_chimeric_table
    from coral_amd.records import DeviceRecords
    M, S = 0, 4
    n = 90
    sa = [(0, 1000 + 700 * k, k % 2, 100 * k if k else 0, 100, 0, 100 * (n - k), 60, 1) for k in range(1, n + 1)]
    recs = [dict(tid=0, pos=10, cigar=[(M, 100), (S, 100 * n)], name="long", sa=sa), dict(tid=0, pos=20, cigar=[(M, 50)], name="y")]
    T = build_chimeric_table(DeviceRecords(synth.records_from_alignments(recs), "cuda:0"))
    assert T.n_reads == 1 and T.n_rows == n
    assert (np.diff(T.qs) >= 0).all()
    assert T.pairs.shape == (2 * n, 8) and (T.pairs[:, 5] & 1).sum() == (n - 1) + (n - 2)


@pytest.mark.parametrize("name", ["tiny", "tiny_edge", "small", "ultra"])
def test_sa_table_matches_oracle_fetch(name):
    """coral_sa_table (K3) vs the oracle's fetch(): same reads in the same (dict) order, same failed reads, same rows
    (qs, qe, rint, mapq, NM rate) in the same (qs, qe)-sorted order, same read lengths."""
    from coral_amd import kernels
    from coral_amd.chimeric import build_chimeric_table
    from coral_amd.records import DeviceRecords
    from oracle import coral_oracle as O
    from oracle.hostrecords import HostRecords
    cfg, rec = synth.dataset(name, "cpu")
    h = HostRecords(rec)
    ob = O.OracleGraphBuild.__new__(O.OracleGraphBuild)
    ob.rec, ob.read_length, ob.chimeric_alignments, ob.nm_stats = h, {}, {}, [0.0, 0.0, 0]
    ob.fetch()
    dr = DeviceRecords(rec, "cuda:0")
    T = build_chimeric_table(dr)
    names = dr.names
    assert [names[i] for i in T.name_id] == list(ob.chimeric_alignments)
    assert {names[i]: int(v) for i, v in enumerate(T.read_length) if v >= 0} == ob.read_length
    for r, (rn, ca) in enumerate(ob.chimeric_alignments.items()):
        a, b = int(T.off[r]), int(T.off[r + 1])
        if len(ca) == 3:
            assert T.failed[r] and a == b
            continue
        assert not T.failed[r]
        assert [[int(T.qs[k]), int(T.qe[k])] for k in range(a, b)] == ca[0]
        assert [[dr.header_chroms[T.tid[k]], int(T.ra[k]), int(T.rb[k]), "+-"[T.strand[k]]] for k in range(a, b)] == ca[1]
        assert [int(T.mapq[k]) for k in range(a, b)] == ca[2]
        assert [float(T.nm[k]) for k in range(a, b)] == ca[3]
    if name == "tiny_edge":
        assert "edge_noprimary" not in ob.chimeric_alignments and any(T.failed)


def test_sa_table_error_semantics():
    """Unknown SA CIGAR shape -> KeyError (cp:255); zero-length query interval -> ZeroDivisionError (cp:268)."""
    from coral_amd.chimeric import build_chimeric_table
    from coral_amd.records import DeviceRecords
    M, S = 0, 4
    bad_shape = synth.records_from_alignments([
        dict(tid=0, pos=10, cigar=[(S, 5), (M, 50)], name="x", sa=[(0, 500, 0, -2, 40, 0, 0, 60, 1)]),
        dict(tid=0, pos=20, cigar=[(M, 50)], name="y")])
    with pytest.raises(KeyError):
        build_chimeric_table(DeviceRecords(bad_shape, "cuda:0"))
    zero = synth.records_from_alignments([
        dict(tid=0, pos=10, cigar=[(S, 5), (M, 50)], name="x", sa=[(0, 500, 0, 54, 1, 0, 0, 60, 1)]),     # SM '+': qs=54, qe=rl-1=54
        dict(tid=0, pos=20, cigar=[(M, 50)], name="y")])
    with pytest.raises(ZeroDivisionError):
        build_chimeric_table(DeviceRecords(zero, "cuda:0"))
