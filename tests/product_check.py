"""Shared checker: run the PRODUCT graph builder phase by phase and compare with the reference's golden snapshots."""
import json
import os

import numpy as np
import torch

from coral_amd import synth
from tests.canon import canon, graph_snapshot, records_digest, strip_cn

HASHSEED0 = os.environ.get("PYTHONHASHSEED") == "0"
_cache = {}


def load_case(golden_dir, case):
    with open(os.path.join(golden_dir, "e2e_%s.json" % case)) as fp:
        gold = json.load(fp)
    name = gold["config"]
    if name not in _cache:
        _cache[name] = synth.dataset(name, "cpu")
    cfg, rec = _cache[name]
    assert records_digest(rec) == gold["records_sha256"], "synthetic inputs changed: regenerate the goldens"
    return gold, cfg, rec


def cn_close(a, b, tol=1e-6):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert abs(x - y) <= tol * max(1.0, abs(y)), (x, y)


def compare_graph_text(a, b):
    """Every column byte-identical except the %f CN column: 1e-6 relative (north_star tolerance for CN floats)."""
    la, lb = a.splitlines(), b.splitlines()
    assert len(la) == len(lb)
    for x, y in zip(la, lb):
        fx, fy = x.split("\t"), y.split("\t")
        assert len(fx) == len(fy)
        if fx[0] == "sequence":
            assert fx[:3] == fy[:3] and fx[4:] == fy[4:], (x, y)
            assert abs(float(fx[3]) - float(fy[3])) <= 1e-6 * max(1.0, abs(float(fy[3]))) + 1e-6
        elif fx[0] in ("concordant", "discordant", "source"):
            assert fx[:2] == fy[:2] and fx[3:] == fy[3:], (x, y)
            assert abs(float(fx[2]) - float(fy[2])) <= 1e-6 * max(1.0, abs(float(fy[2]))) + 1e-6
        else:
            assert x == y


def check_product_against_golden(case, golden_dir, tmp_path, device):
    """Phase-by-phase equality with the reference (strict order checks need PYTHONHASHSEED=0)."""
    from coral_amd import infer_breakpoint_graph as ibg
    from coral_amd.breakpoint_graph import breakpoint_info_text, compute_cn_lr, graph_text
    from coral_amd.records import DeviceRecords
    gold, cfg, rec = load_case(golden_dir, case)
    fmt = gold.get("cn_format", "bed")
    cn = str(tmp_path / ("cn." + fmt)); seeds = str(tmp_path / "seeds.bed")
    (synth.write_cn_bed if fmt == "bed" else synth.write_cn_cns)(cfg, cn); synth.write_seed_bed(cfg, seeds)
    dr = DeviceRecords(rec, device)
    b = ibg.bam_to_breakpoint_nanopore(None, seeds, records=dr)
    b.min_bp_cov_factor = gold["min_bp_support"]
    b.read_cns(cn)
    assert b.normal_cov == gold["A2"]["normal_cov"]
    assert b.min_cluster_cutoff == gold["A2"]["min_cluster_cutoff"]
    b.fetch()
    assert len(b.read_length) == gold["A3"]["n_read_length"]
    assert canon(dict(b.chimeric_alignments.items())) == gold["A3"]["chimeric_alignments"]
    assert canon(b.nm_stats) == gold["A3"]["nm_stats"]
    b.hash_alignment_to_seg()
    assert canon(dict(b.chimeric_alignments.items())) == gold["A4"]["chimeric_alignments"]
    # inverted index: chromosome key order and every per-segment read list must match; the order in which the
    # CN-segment keys were first inserted is not observable (the reference only ever looks keys up)
    def seg_norm(c):
        return [[k, sorted(v["__dict__"], key=lambda kv: kv[0])] for k, v in c["__dict__"]]
    assert seg_norm(canon(b.chimeric_alignments_seg.as_dict())) == seg_norm(gold["A4"]["chimeric_alignments_seg"])
    b.find_amplicon_intervals()
    if HASHSEED0:
        assert canon(b.amplicon_intervals) == gold["A5"]["amplicon_intervals"]
        assert canon(b.new_bp_list) == gold["A5"]["new_bp_list"]
        assert canon(b.amplicon_interval_connections) == gold["A5"]["amplicon_interval_connections"]
        assert canon(b.new_bp_stats) == gold["A5"]["new_bp_stats"]
    else:
        assert sorted(map(json.dumps, canon(b.amplicon_intervals))) == sorted(map(json.dumps, gold["A5"]["amplicon_intervals"]))
    b.find_smalldel_breakpoints()
    assert canon(b.large_indel_alignments) == gold["A6"]["large_indel_alignments"]
    if HASHSEED0:
        assert canon(b.new_bp_list) == gold["A6"]["new_bp_list"]
    b.find_breakpoints()
    if HASHSEED0:
        assert canon(b.new_bp_list) == gold["A7"]["new_bp_list"]
        assert canon(b.new_bp_stats) == gold["A7"]["new_bp_stats"]
        assert canon(b.new_bp_ccids) == gold["A7"]["new_bp_ccids"]
        assert canon(b.amplicon_interval_connections) == gold["A7"]["amplicon_interval_connections"]
    b.build_graph()
    if HASHSEED0:
        assert canon(b.ccid2id) == gold["A9"]["ccid2id"]
        assert [graph_snapshot(g) for g in b.lr_graph] == gold["A9"]["graphs"]
    if "A10" not in gold:      # --output_bp variant
        files = {}
        for gi, g in enumerate(b.lr_graph):
            stats = []
            for e in g.discordant_edges:
                for k, bp in enumerate(b.new_bp_list):
                    if e[:6] == bp[:6]:
                        stats.append(b.new_bp_stats[k]); break
            files["out_amplicon%d_breakpoints.txt" % (gi + 1)] = breakpoint_info_text(g, stats)
        if HASHSEED0:
            assert files == gold["files"]
        return b
    b.assign_cov()
    if HASHSEED0:
        assert [graph_snapshot(g) for g in b.lr_graph] == gold["A10"]["graphs"]
    for g in b.lr_graph:
        compute_cn_lr(g, b.normal_cov)
    if HASHSEED0:
        for g, gg in zip(b.lr_graph, gold["A11"]["graphs"]):
            s, cns = strip_cn(graph_snapshot(g))
            sg, cng = strip_cn(gg)
            assert s == sg
            cn_close(cns, cng)
        # what the cycle step asks of each graph it receives (cd:146, :623, :1029; bg:609-693)
        assert [g.infer_discordant_edge_multiplicities() for g in b.lr_graph] == gold["A11x"]["discordant_edge_multiplicities"]
        assert [g.infer_max_seq_multiplicity() for g in b.lr_graph] == gold["A11x"]["max_seq_multiplicity"]
        files = {"out_amplicon%d_graph.txt" % (gi + 1): graph_text(g) for gi, g in enumerate(b.lr_graph)}
        assert sorted(files) == sorted(gold["files"])
        for k in files:
            compare_graph_text(files[k], gold["files"][k])
        if "F2" in gold:                  # path constraints of the cycle step
            b.compute_path_constraints()
            assert canon(b.path_constraints) == gold["F2"]["path_constraints"]
    else:
        got = sorted(l.split("\t")[0:3:2] for g in b.lr_graph for l in graph_text(g).splitlines() if l.startswith("disc"))
        exp = sorted(l.split("\t")[0:3:2] for t in gold["files"].values() for l in t.splitlines() if l.startswith("disc"))
        assert len(got) == len(exp)
    return b


def pair_table_cpu(cols, off, chroms, chr_rank, cutoff=100, min_mapq=20, gap_=100, gap_mapq=10):
    """coral_bp_pair_table stand-in (layout of csrc/coral_kernels.hip K4: slot 2 * g + kind, 8 ints per slot), every candidate
    made by the ORACLE's interval2bp; the interval-independent tests restated from bu:70-96 / :129-186."""
    from oracle import coral_oracle as O
    n_rows = cols.shape[1]
    pairs = np.zeros((2 * n_rows, 8), dtype=np.int32)
    tid_of = {c: k for k, c in enumerate(chroms)}
    qs, qe, tid, ra, rb, strand, mapq = (cols[k].tolist() for k in range(7))

    def fill(slot, a, b, mid, skip):
        gap = qs[b] - qe[a]
        ok = mapq[a] >= min_mapq and mapq[b] >= min_mapq and ((mapq[mid] < gap_mapq) if skip else (gap + cutoff >= 0))
        bad = chr_rank[tid[a]] < 0 or chr_rank[tid[b]] < 0
        if bad:            # the reference raises KeyError here; the candidate content is never looked at
            c = [chroms[tid[a]], rb[a], "+-"[strand[a]], chroms[tid[b]], ra[b], "-+"[strand[b]], None, gap, 0]
        else:
            c = O.interval2bp([chroms[tid[a]], ra[a], rb[a], "+-"[strand[a]]], [chroms[tid[b]], ra[b], rb[b], "+-"[strand[b]]],
                              ("r", 0, 1), gap)
        grr = (ra[b] - rb[a]) if strand[b] == 0 else (rb[a] - ra[b])
        far = abs(gap - grr) > max(gap_, abs(gap * 0.2))
        o1, o2 = "+-".index(c[2]), "+-".index(c[5])
        bits = 1 | (2 if ok else 0) | (o1 << 2) | (o2 << 3) | (16 if c[8] else 0) | (32 if strand[a] != strand[b] else 0) | \
            (64 if far else 0) | (128 if bad else 0) | ((mapq[a] & 255) << 8) | ((mapq[b] & 255) << 16)
        pairs[slot] = [tid_of[c[0]], c[1], tid_of[c[3]], c[4], gap, bits, a, b]

    for r in range(len(off) - 1):
        base, n = int(off[r]), int(off[r + 1] - off[r])
        for k in range(n):
            g = base + k
            if k + 1 < n:
                fill(2 * g, g, g + 1, g, False)
            if 1 <= k and k + 1 < n:
                fill(2 * g + 1, g - 1, g + 1, g, True)
    return pairs


# ---- oracle-backed stand-ins for the device kernels (CPU tests of the host logic only) ----------------
def install_cpu_kernel_fakes(monkeypatch):
    from coral_amd import kernels
    from oracle.hostrecords import HostRecords
    hosts = {}

    class _LocalShard:
        """What the record kernels of this process read: the product's own arrays of the local shard (coral_amd.records), as the
        records object the oracle's HostRecords wraps.  Fields the record kernels never see (qlen, NM, names, SA) are empty."""

        def __init__(self, dr):
            g = lambda t: t.cpu().numpy()
            self.n, self.header_chroms = dr.n, dr.header_chroms
            self.tid, self.pos, self.end, self.n_cigar = g(dr.tid), g(dr.pos), g(dr.end), g(dr.n_cigar)
            fm = g(dr.flagmq).astype(np.int64)
            self.flag, self.mapq, self.has_seq = fm & 0xFFFF, (fm >> 16) & 0xFF, (fm >> 24) & 1
            z = np.zeros(dr.n, dtype=np.int64)
            self.qlen, self.nm, self.name_id = z, z, z
            self.cigar_off, self.cigar = g(dr.cigar_off), g(dr.cigar)
            self.sa_off, self.sa, self.sa_nm = np.zeros(dr.n + 1, dtype=np.int64), np.zeros((0, 8), dtype=np.int64), z[:0]
            self.nonacgt_rec, self.nonacgt_pos = z[:0], z[:0]

        def materialise_names(self):
            return []

    def host_of(dr):
        if id(dr) not in hosts:
            hosts[id(dr)] = (HostRecords(_LocalShard(dr)), dr)          # (dr kept alive: ids are not re-used meanwhile)
        return hosts[id(dr)][0]

    # The stand-ins replace only the LOCAL launches (records [dr.lo, dr.hi) of this process, ordinals local to the shard); the
    # exchange and ordering code of coral_amd.kernels / coral_amd.sharding stays the product's own.
    def scan_local(dr, min_gap, min_mapq, gap_cap):
        h = host_of(dr)
        mb, qi, b0, b1, rows = [], [], [], [], []
        for i in range(dr.n):
            bl = h.blocks(i)
            mb.append(sum(e - s for s, e in bl)); qi.append(h.infer_read_length(i) or 0)
            b0.append(bl[0][0] if bl else -1); b1.append(bl[-1][1] if bl else -1)
            if h.mapq[i] >= min_mapq:
                for k in range(len(bl) - 1):
                    if abs(bl[k + 1][0] - bl[k][1]) > min_gap:
                        rows.append((i, k + 1, bl[k][1], bl[k + 1][0], b0[-1], b1[-1]))
        summary = torch.tensor([mb, qi, b0, b1], dtype=torch.int32).t().contiguous().reshape(-1, 4)
        return summary, lambda: torch.tensor(rows, dtype=torch.int64).reshape(-1, 6)

    def coverage_local(dr, scan, sg):
        h = host_of(dr)
        out = torch.zeros((2, len(sg)), dtype=torch.int64)
        for j, (t, s, e) in enumerate(sg):
            idx = h.region(h.chroms[t], s, e)
            out[0, j] = sum(1 for i in idx if h.infer_read_length(i))
            tot = 0
            for i in idx:
                if h.has_seq[i] and h.n_cigar[i]:
                    tot += sum(max(0, min(b, e) - max(a, s)) for a, b in h.blocks(i))
            out[1, j] = tot
        return out

    def points_local(dr, uniq, pair_cap):
        h = host_of(dr)
        keys = []
        for j, (t, p) in enumerate(uniq):
            keys += [(j << 32) | int(i) for i in h.region(h.chroms[t], p, p + 1)]
        return torch.tensor(keys, dtype=torch.int64)

    def sa_table_local(dr):
        """coral_sa_table stand-in: the oracle's fetch() (string SA entries, per-read Python lists) turned into arrays."""
        from oracle import coral_oracle as O

        class _WholeFile:          # the host mirrors of the whole file (the product builds the SA table from them too)
            n, header_chroms = dr.n_total, dr.header_chroms
            tid, pos, end, flag, mapq, qlen, has_seq, nm = dr.h_tid, dr.h_pos, dr.h_end, dr.h_flag, dr.h_mapq, dr.h_qlen, dr.h_has_seq, dr.h_nm
            name_id, n_cigar, sa_off, sa, sa_nm = dr.h_name_id, dr.h_n_cigar, dr.h_sa_off, dr.h_sa, dr.h_sa_nm
            cigar_off, cigar = np.zeros(dr.n_total + 1, dtype=np.int64), np.zeros(0, dtype=np.int32)
            nonacgt_rec, nonacgt_pos = dr.h_nonacgt_rec, dr.h_nonacgt_pos
            materialise_names = staticmethod(lambda: dr.names)
        h = HostRecords(_WholeFile)
        ob = O.OracleGraphBuild.__new__(O.OracleGraphBuild)
        ob.rec, ob.read_length, ob.chimeric_alignments, ob.nm_stats = h, {}, {}, [0.0, 0.0, 0]
        ob.fetch()
        name_id_of = {nm: k for k, nm in enumerate(h.names)}
        tid_of = {c: k for k, c in enumerate(h.chroms)}
        rows, off, names, failed = [], [0], [], []
        for rn, ca in ob.chimeric_alignments.items():
            names.append(name_id_of[rn])
            failed.append(len(ca) == 3)
            if len(ca) == 4:
                for q, ri, mq, nmr in zip(*ca):
                    rows.append([q[0], q[1], tid_of[ri[0]], ri[1], ri[2], 0 if ri[3] == "+" else 1, mq, int(round(nmr * (q[1] - q[0])))])
            off.append(len(rows))
        rl = np.full(len(h.names), -1, dtype=np.int64)
        for rn, v in ob.read_length.items():
            rl[name_id_of[rn]] = v
        cols = np.ascontiguousarray(np.array(rows, dtype=np.int64).reshape(-1, 8).T)          # [8, n_rows], as the product's wrapper
        return (cols, np.array(off, dtype=np.int64), np.array(names, dtype=np.int64),
                np.array(failed, dtype=bool), rl, pair_table_cpu(cols, off, h.chroms, dr.chr_rank), None, None)

    def hash_rows_local(dr, T, seg, tid_has_segs):
        """coral_hash_rows stand-in: point queries by linear search over the segment table (the reference's IntervalTree
        queries, ibg:190-191), entries appended alignment by alignment and stably sorted by (contig, segment)."""
        segs_of = {}
        for t, st, en, ix in seg.T.tolist():
            segs_of.setdefault(t, []).append((st, en, ix))
        c0 = np.full(T.n_rows, -3, dtype=np.int64)
        c1 = np.full(T.n_rows, -3, dtype=np.int64)
        entries = []
        for r in range(T.n_rows):
            t = int(T.tid[r])
            if not tid_has_segs[t]:
                continue
            lo, hi = min(int(T.ra[r]), int(T.rb[r])), max(int(T.ra[r]), int(T.rb[r]))
            hits = [[ix for st, en, ix in segs_of.get(t, ()) if st <= p < en] for p in (lo, hi)]
            assert all(len(h) <= 1 for h in hits)
            c0[r], c1[r] = (hits[0] or [-1])[0], (hits[1] or [-1])[0]
            if c0[r] >= 0:
                entries.append(((t << 32) | int(c0[r]), r))
            if c1[r] >= 0 and c1[r] != c0[r]:
                entries.append(((t << 32) | int(c1[r]), r))
        entries.sort(key=lambda e: e[0])                      # stable
        e = np.array(entries, dtype=np.int64).reshape(-1, 2)
        return c0, c1, e[:, 0].copy(), e[:, 1].copy()

    monkeypatch.setattr(kernels, "_hash_rows_local", hash_rows_local)
    monkeypatch.setattr(kernels, "_sa_table_local", sa_table_local)
    monkeypatch.setattr(kernels, "_scan_local", scan_local)
    monkeypatch.setattr(kernels, "_coverage_local", coverage_local)
    monkeypatch.setattr(kernels, "_points_local", points_local)
