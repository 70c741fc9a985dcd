"""Which host thread pool burns CPU during builds?  Per-task CPU ticks before / after 6 builds under different pool settings."""
import sys, os, time, tempfile
sys.path.insert(0, ".")
mode = sys.argv[1]
import torch
if mode == "torch4":
    torch.set_num_threads(4)
from coral_amd import synth, sharding
from threadpoolctl import threadpool_info, threadpool_limits
cfg = synth.named_config("cfg3")
work = tempfile.mkdtemp()
cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
synth.write_cn_bed(cfg, cn); synth.write_seed_bed(cfg, seeds)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000); torch.cuda.synchronize()
dr = sharding.shard_records(rec, 0, 1, "cuda:0"); del rec

def ticks():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            rest = f[f.rindex(")") + 2:].split()
            out[tid] = int(rest[11]) + int(rest[12])
        except Exception:
            pass
    return out
sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "w"))
print(mode, "pools:", [(d["user_api"], d["internal_api"], d["num_threads"]) for d in threadpool_info()], flush=True)
a = ticks(); t0 = time.perf_counter()
for i in range(6):
    b = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "p%d" % i))
dt = time.perf_counter() - t0
z = ticks()
d = sorted(((z[k] - a.get(k, 0)) for k in z), reverse=True)
print(mode, "6 builds %.0f ms; tasks %d; cpu ticks during builds: total %d, top %s" % (dt * 1e3, len(z), sum(d), d[:8]), flush=True)
