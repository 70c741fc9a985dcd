"""Where do the ~30 ms of find_amplicon_intervals go at config 3?  Wall-clock timers around the pieces of the interval search
(no cProfile: its per-call overhead distorts Python-heavy code) + the native side's own counters (CORAL_SEARCH_PROFILE=1).

    CORAL_SEARCH_PROFILE=1 python tools/profile_search.py [threads ...]
"""
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
B = ibg.bam_to_breakpoint_nanopore


def timed(name):
    orig = getattr(B, name)

    def wrap(self, *a, **kw):
        t = time.perf_counter()
        try:
            return orig(self, *a, **kw)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
            acc[name + "#"] = acc.get(name + "#", 0) + 1
    setattr(B, name, wrap)


for nm in ("find_interval_i", "_search_step", "_prefetch_step", "addbp", "_merge_intervals", "_support", "_search", "_read_hashes"):
    timed(nm)
orig_call = B._call_breakpoints


def call_wrap(self, c, advance_subcluster, called=None):           # time spent INSIDE the generator (not in its consumer)
    g = orig_call(self, c, advance_subcluster, called)
    while True:
        t = time.perf_counter()
        try:
            v = next(g)
        except StopIteration:
            acc["_call_breakpoints"] = acc.get("_call_breakpoints", 0.0) + time.perf_counter() - t
            return
        acc["_call_breakpoints"] = acc.get("_call_breakpoints", 0.0) + time.perf_counter() - t
        yield v


B._call_breakpoints = call_wrap
orig_step = ibg.PairSearch.step
orig_result = ibg.PairSearch._result


def step_wrap(self, *a, **kw):
    t = time.perf_counter()
    try:
        return orig_step(self, *a, **kw)
    finally:
        acc["PairSearch.step"] = acc.get("PairSearch.step", 0.0) + time.perf_counter() - t


def result_wrap(self, *a, **kw):
    t = time.perf_counter()
    try:
        return orig_result(self, *a, **kw)
    finally:
        acc["PairSearch._result"] = acc.get("PairSearch._result", 0.0) + time.perf_counter() - t


ibg.PairSearch.step = step_wrap
ibg.PairSearch._result = result_wrap

for threads in (sys.argv[1:] or ["6"]):
    os.environ["CORAL_SEARCH_THREADS"] = threads
    for i in range(7):
        acc.clear()
        b = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "p"))
        ph = ibg.PHASE_SECONDS
        if i >= 4:
            print("threads %s build %d: find_amplicon_intervals %.1f ms | " % (threads, i, ph["find_amplicon_intervals"] * 1e3) +
                  "  ".join("%s %.1f%s" % (k, v * 1e3, (" (x%d)" % acc[k + "#"]) if k + "#" in acc else "") for k, v in sorted(acc.items())
                            if not k.endswith("#")), flush=True)
        b = None
print("intervals searched: see x-counts; _search_step includes PairSearch.step (= native wait + _result); find_interval_i includes everything but "
      "_prefetch_step of the seeds and _merge_intervals")
