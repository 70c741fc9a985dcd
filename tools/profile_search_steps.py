"""Per-step timeline of the native interval search at config 3 (CORAL_SEARCH_PROFILE=2) + where a whole step of the bench goes
outside the phases (free of the previous result, garbage collection).     python tools/profile_search_steps.py"""
import gc
import os
import sys
import tempfile
import time

sys.path.insert(0, ".")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import torch
from coral_amd import synth, sharding
from coral_amd import infer_breakpoint_graph as ibg

cfg = synth.named_config("cfg3")
work = tempfile.mkdtemp()
cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
synth.write_cn_bed(cfg, cn)
synth.write_seed_bed(cfg, seeds)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000)
rec.names = rec.name_table()
torch.cuda.synchronize()
dr = sharding.shard_records(rec, 0, 1, "cuda:0")
del rec
acc = {}


def timed(obj, name, label=None):
    orig = getattr(obj, name)
    label = label or name

    def wrap(*a, **kw):
        t = time.perf_counter()
        try:
            return orig(*a, **kw)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
    setattr(obj, name, wrap)


from coral_amd import kernels, chimeric
B = ibg.bam_to_breakpoint_nanopore
for nm in ("_find_intervals_native", "_merge_intervals", "_search", "_coverage", "_add_clustered"):
    timed(B, nm)
timed(chimeric.PairSearch, "bfs", "PairSearch.bfs (native call + copies)")
timed(chimeric.PairSearch, "within", "PairSearch.within")
for nm in ("segment_coverage", "point_cover", "_coverage_local", "_points_local", "hash_rows", "sa_table"):
    timed(kernels, nm, "kernels." + nm)
timed(ibg, "call_breakpoints", "call_breakpoints")
timed(ibg, "compute_cn_lr", "compute_cn_lr")
b = None
for i in range(8):
    acc.clear()
    if i == 7:
        os.environ["CORAL_SEARCH_PROFILE"] = "2"
    t0 = time.perf_counter()
    nb = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "p%d" % i))
    t1 = time.perf_counter()
    ph = dict(ibg.PHASE_SECONDS)
    g0 = gc.get_count()
    b = nb
    nb = None
    t2 = time.perf_counter()
    os.environ.pop("CORAL_SEARCH_PROFILE", None)
    if i >= 4:
        print("   inside: " + "  ".join("%s %.1f" % (k, v * 1e3) for k, v in sorted(acc.items())) + "   | phases: " +
              " ".join("%s %.1f" % (k[:10], v * 1e3) for k, v in ph.items()), flush=True)
    print("step %d: build %.1f ms = phases %.1f + %.1f outside them; freeing the previous result %.1f ms; gc counts after the build %s"
          % (i, (t1 - t0) * 1e3, sum(ph.values()) * 1e3, (t1 - t0 - sum(ph.values())) * 1e3, (t2 - t1) * 1e3, g0), flush=True)
