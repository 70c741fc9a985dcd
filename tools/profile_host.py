"""cProfile of one full graph-build step on GPU-resident synthetic records (host-logic hot spots)."""
import cProfile, pstats, sys, os, time, tempfile, io
sys.path.insert(0, ".")
import torch
from coral_amd import synth, sharding
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
cfg = synth.scaled_config(name, n)
work = tempfile.mkdtemp()
cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
synth.write_cn_bed(cfg, cn); synth.write_seed_bed(cfg, seeds)
t = time.time(); rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000); torch.cuda.synchronize(); print("gen %.1fs" % (time.time() - t), flush=True)
t = time.time(); dr = sharding.shard_records(rec, 0, 1, "cuda:0"); print("DeviceRecords %.2fs" % (time.time() - t), flush=True)
sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "w"))
pr = cProfile.Profile(); t = time.time(); pr.enable()
b = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "p"))
pr.disable(); print("step %.2fs  graphs %d  bps %d  chimeric %d" % (time.time() - t, len(b.lr_graph), len(b.new_bp_list), len(b.chimeric_alignments)), flush=True)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue())
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25); print(s.getvalue())
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_callees("find_smalldel_breakpoints"); print(s.getvalue()[:6000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_callees("_add_clustered|_call_breakpoints"); print(s.getvalue()[:6000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_callees("assign_cov"); print(s.getvalue()[:5000])
