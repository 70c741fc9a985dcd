"""Where does a step's wall time go outside the phases? (alloc/free of the previous result, gc, ...)"""
import sys, os, time, tempfile, gc
sys.path.insert(0, ".")
import torch
from coral_amd import synth, sharding
from coral_amd import infer_breakpoint_graph as ibg
cfg = synth.named_config("cfg3")
work = tempfile.mkdtemp()
cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
synth.write_cn_bed(cfg, cn); synth.write_seed_bed(cfg, seeds)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000); torch.cuda.synchronize()
dr = sharding.shard_records(rec, 0, 1, "cuda:0"); del rec
b = None
for i in range(5):
    t0 = time.perf_counter()
    nb = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "p%d" % i))
    t1 = time.perf_counter()
    b = nb            # frees the previous step's result
    nb = None
    t2 = time.perf_counter()
    t3 = time.perf_counter(); x = [[] for _ in range(2000)]; t4 = time.perf_counter()      # allocations that trip the collector
    print("   first allocations after the build: %.1f ms" % ((t4 - t3) * 1e3))
    print("step %d: build %.1f ms (phases sum %.1f ms) free-previous %.1f ms gc counts %s" % (i, (t1 - t0) * 1e3, sum(ibg.PHASE_SECONDS.values()) * 1e3, (t2 - t1) * 1e3, gc.get_count()), flush=True)
