"""Per-step wall time against the per-phase times of build_graph_from_records (where do slow steps lose their time?)."""
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
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    t0 = time.perf_counter()
    nb = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "p%d" % i))
    t1 = time.perf_counter()
    ph = dict(ibg.PHASE_SECONDS)
    b = nb            # frees the previous step's result
    nb = None
    t2 = time.perf_counter()
    print("step %2d: build %6.1f ms  phases sum %6.1f  free-previous %5.1f  | %s" % (
        i, (t1 - t0) * 1e3, sum(ph.values()) * 1e3, (t2 - t1) * 1e3, " ".join("%s %.1f" % (k[:9], v * 1e3) for k, v in ph.items())), flush=True)
if os.environ.get("CORAL_THREAD_CPU") == "1":          # which threads burned CPU time?  (utime + stime per task, in ticks)
    rows = []
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            rows.append((int(rest[11]) + int(rest[12]), comm, tid))
        except Exception:
            pass
    rows.sort(reverse=True)
    print("threads: %d; cpu ticks (100/s) of the busiest: %s" % (len(rows), ", ".join("%s:%d" % (c, t) for t, c, _ in rows[:14])))
