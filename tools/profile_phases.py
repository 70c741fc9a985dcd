"""cProfile of one step, callee breakdown of the phases outside the interval search."""
import cProfile, pstats, sys, os, tempfile, io
sys.path.insert(0, ".")
import torch
from coral_amd import synth, sharding
cfg = synth.named_config("cfg3")
work = tempfile.mkdtemp()
cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
synth.write_cn_bed(cfg, cn); synth.write_seed_bed(cfg, seeds)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000); torch.cuda.synchronize()
dr = sharding.shard_records(rec, 0, 1, "cuda:0"); del rec
sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "w"))
pr = cProfile.Profile(); pr.enable()
b = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "p"))
pr.disable()
for pat in ("launch_record_kernels", "build_chimeric_table", "[(]fetch[)]", "hash_alignment_to_seg", "assign_cov", "_sa_table_local", "segment_coverage", "point_cover"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_callees(pat); print(s.getvalue()[:3500])
