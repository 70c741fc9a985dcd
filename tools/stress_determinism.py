"""Repeat the full-size build many times and check that every run gives the same result (look-ahead threads, atomics in the
kernels, set replay): hash of the graph text + breakpoint list + read-support sets."""
import hashlib, os, sys, tempfile, time
sys.path.insert(0, ".")
import torch
from coral_amd import synth, sharding
from coral_amd.breakpoint_graph import graph_text
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = synth.named_config("cfg3")
work = tempfile.mkdtemp()
cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
synth.write_cn_bed(cfg, cn); synth.write_seed_bed(cfg, seeds)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000); torch.cuda.synchronize()
dr = sharding.shard_records(rec, 0, 1, "cuda:0"); del rec
seen = {}
t0 = time.time()
for k in range(n_runs):
    os.environ["CORAL_AHEAD_THREADS"] = str(1 + k % 4)          # 1..4 worker threads
    b = sharding.build_graph_sharded(dr, seeds, cn, None)
    h = hashlib.sha256()
    for g in b.lr_graph:
        h.update(graph_text(g).encode())
        for e in g.discordant_edges:
            h.update(repr(sorted(e[10])).encode())
    h.update(repr([bp[:9] for bp in b.new_bp_list]).encode())
    h.update(repr(b.amplicon_intervals).encode())
    seen.setdefault(h.hexdigest(), []).append(k)
print("%d runs in %.1fs, distinct results: %d %s" % (n_runs, time.time() - t0, len(seen), {k[:12]: len(v) for k, v in seen.items()}))
assert len(seen) == 1, "non-deterministic result"
