"""Rank 0's merge of the per-rank decodes, measured on the CPU (VERDICT r02 next #1b: <= 0.15 s for 8 pieces of the 2 M-read file).

    python tools/bench_merge.py [n_reads [world [threads]]]

Records of config 3's read / record / name structure (2 M reads, 2.16 M records, 164 k reads with records in several places) with
THIN CIGARs (one indel event per 4 kb instead of per 20 bp: the merge never sees CIGARs, and the full-size 16 GB CIGAR array is
beyond this container's CPU generator), cut into `world` consecutive record ranges as `load_bam_sharded` sees them: range-local name ids in
first-seen order + the range's own name table.  Timed: HostMirrors.pack per piece (what a rank sends), unpack + from_pieces
(what rank 0 does: native name join, id remap, column concatenation).  Checked against a single-process numbering.
"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
from coral_amd import synth
from coral_amd.names import NameTable
from coral_amd.records import HostMirrors

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = synth.scaled_config("cfg3", n_reads)
synth.EV_SPACING = 4000          # (this tool only) one indel event per 4 kb instead of per 20 bp: same records and names, thin CIGARs
t = time.perf_counter()
rec = synth.generate(cfg, "cpu", chunk_pieces=400000)
print("generated %d records / %d names in %.1f s" % (rec.n, rec.n_names, time.perf_counter() - t), flush=True)
whole = HostMirrors.piece_of(rec)
names = rec.name_table()


class _Piece:
    pass


def cut(a, b):
    """Records [a, b) as a rank's own decode would deliver them: local name ids in first-seen order, local name table."""
    p = _Piece()
    for k in ("tid", "pos", "end", "flag", "mapq", "qlen", "has_seq", "nm", "n_cigar"):
        setattr(p, k, getattr(rec, k)[a:b])
    gid = rec.name_id[a:b].numpy()
    u, first, inv = np.unique(gid, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank_of = np.empty(len(u), dtype=np.int64)
    rank_of[order] = np.arange(len(u))
    p.name_id = torch.from_numpy(rank_of[inv].astype(np.int32))
    local_global = u[order]
    lens = np.diff(names.off)[local_global]
    off = np.zeros(len(u) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    idx = np.repeat(names.off[local_global] - off[:-1], lens) + np.arange(off[-1])
    p.names = NameTable(names.blob[idx], off)
    p.name_table = lambda: p.names
    p.n_names = len(u)
    s0, s1 = int(rec.sa_off[a]), int(rec.sa_off[b])
    p.sa_off = rec.sa_off[a:b + 1] - s0
    p.sa, p.sa_nm = rec.sa[s0:s1], rec.sa_nm[s0:s1]
    m = (rec.nonacgt_rec >= a) & (rec.nonacgt_rec < b)
    p.nonacgt_rec, p.nonacgt_pos = rec.nonacgt_rec[m] - a, rec.nonacgt_pos[m]
    return p


cuts = [rec.n * r // world for r in range(world + 1)]
pieces = [cut(cuts[r], cuts[r + 1]) for r in range(world)]
t = time.perf_counter()
packed = [HostMirrors.pack(HostMirrors.piece_of(p)) for p in pieces]
t_pack = time.perf_counter() - t
print("pack (all %d ranks, one after the other): %.3f s, %.1f MB in total" % (world, t_pack, sum(len(b) for b in packed) / 1e6), flush=True)
best = None
for trial in range(5):
    t = time.perf_counter()
    host = HostMirrors.from_pieces([HostMirrors.unpack(b) for b in packed], n_threads=threads)
    dt = time.perf_counter() - t
    best = dt if best is None else min(best, dt)
    print("rank-0 merge of %d pieces (%d threads): %.3f s" % (world, threads, dt), flush=True)
assert host.n_total == rec.n and host.n_names == rec.n_names
assert np.array_equal(host.h_name_id, whole["name_id"]), "unified name ids differ from the single-process numbering"
assert host._names == names
for k in ("tid", "pos", "end", "flag", "mapq", "nm", "n_cigar"):
    assert np.array_equal(getattr(host, "h_" + k), whole[k]), k
assert np.array_equal(host.h_sa, whole["sa"]) and np.array_equal(host.h_nonacgt_rec, whole["nonacgt_rec"])
print("OK: merged mirrors == single-process mirrors (%d records, %d names); best merge %.3f s" % (rec.n, rec.n_names, best))
