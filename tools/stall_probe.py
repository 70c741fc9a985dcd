"""Where does the main thread stall?  A 1 ms interval timer samples the innermost coral_amd frame; signals are delivered between
bytecodes, so a long gap between two samples = one long C-level call, and the sample after the gap names the line that made it."""
import sys, os, time, tempfile, signal, collections
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
samples = []

def on_tick(signum, frame):
    f, where = frame, None
    while f is not None:
        fn = f.f_code.co_filename
        if "coral_amd" in fn or "tools/" in fn:
            where = "%s:%d %s" % (os.path.basename(fn), f.f_lineno, f.f_code.co_name)
            break
        f = f.f_back
    samples.append((time.perf_counter(), where, "%s:%d" % (os.path.basename(frame.f_code.co_filename), frame.f_lineno)))

signal.signal(signal.SIGALRM, on_tick)
b = None
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    del samples[:]
    signal.setitimer(signal.ITIMER_REAL, 0.001, 0.001)
    t0 = time.perf_counter()
    nb = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "p%d" % i))
    t1 = time.perf_counter()
    signal.setitimer(signal.ITIMER_REAL, 0, 0)
    b = nb
    gaps = sorted(((samples[k + 1][0] - samples[k][0], samples[k + 1][1], samples[k + 1][2], samples[k][1]) for k in range(len(samples) - 1)),
                  reverse=True)[:3]
    print("step %2d: %.1f ms; longest gaps: %s" % (i, (t1 - t0) * 1e3, "; ".join("%.1f ms before [%s | %s] after [%s]" % (g[0] * 1e3, g[1], g[2], g[3]) for g in gaps)), flush=True)
