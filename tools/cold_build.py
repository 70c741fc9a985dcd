"""Where does the FIRST build on fresh records go?  (0.38 s cold against 0.062 s steady state at 2 M reads.)

    python tools/cold_build.py [config [n_reads [profile]]] > gpurun_out/cold_build.txt

Builds three times on a fresh DeviceRecords each (so every per-records one-off is paid every time; only the per-process one-offs —
library load, hipcub workspaces, BLAS pool, pinned pools, set-replay self-check — are paid by build 0 alone), then three times on the
same DeviceRecords (steady state).  With `profileK`, fresh-records build K (0 = cold process) runs under cProfile.
"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, ".")
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1" if _v.startswith("OPENBLAS") else "4")
import torch
from coral_amd import synth, sharding
from coral_amd import infer_breakpoint_graph as ibg

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = synth.named_config(name)
if len(sys.argv) > 2 and int(sys.argv[2]):
    cfg = synth.scaled_config(name, int(sys.argv[2]))
prof = sys.argv[3] if len(sys.argv) > 3 else ""
work = tempfile.mkdtemp()
cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
synth.write_cn_bed(cfg, cn)
synth.write_seed_bed(cfg, seeds)
t = time.perf_counter()
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000)
torch.cuda.synchronize()
print("generated %d records of %s in %.1f s" % (rec.n, cfg.name, time.perf_counter() - t), flush=True)
t = time.perf_counter()
rec.names = rec.name_table()
print("name table (blob + offsets) of the synthetic reads: %.3f s (a decoded BAM brings it)" % (time.perf_counter() - t), flush=True)


if os.environ.get("PRE_POOLS"):
    t = time.perf_counter()
    from coral_amd import hostpools
    hostpools.apply_once()
    print("hostpools.apply_once: %.3f s (once per process; inside the first build unless done before)" % (time.perf_counter() - t), flush=True)


def one(dr, tag, profile=False):
    torch.cuda.synchronize()
    os.environ["CORAL_TRACE_OPEN"] = "1"
    pr = cProfile.Profile() if profile else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    b = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "out"))
    if pr:
        pr.disable()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ph = {k: round(v * 1e3, 1) for k, v in ibg.PHASE_SECONDS.items()}
    print("%-28s %7.1f ms   %s" % (tag, dt * 1e3, ph), flush=True)
    if pr:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
        print(s.getvalue())
    return b


for k in range(3):
    t = time.perf_counter()
    dr = sharding.shard_records(rec, 0, 1, "cuda:0")
    torch.cuda.synchronize()
    print("DeviceRecords(fresh) %d: %.3f s" % (k, time.perf_counter() - t), flush=True)
    one(dr, "build %d on fresh records" % k, profile=(prof == "profile%d" % k))
for k in range(3):
    one(dr, "build %d on the same records" % k)
