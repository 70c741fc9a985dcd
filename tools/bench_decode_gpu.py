"""Decode throughput of the two BAM pipelines on one file of config-3 reads: coral_bam_decode_* (host threads) against
coral_bamgpu_* (inflate + parse on the GPU).      python tools/bench_decode_gpu.py [reads] [level] [repeats] [batch_bytes]"""
import json, os, sys, tempfile, time
sys.path.insert(0, ".")
import torch
from coral_amd import bam, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cfg = synth.scaled_config("cfg3", n)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000).to("cpu")
d = tempfile.mkdtemp(prefix="coral_dec_")
p = os.path.join(d, "x.bam")
t0 = time.perf_counter()
bam.write_bam_native(rec, p, seed=1, level=level)
tw = time.perf_counter() - t0
size = os.path.getsize(p)
out = {"reads": n, "records": rec.n, "level": level, "bgzf_GB": size / 1e9, "write_s": round(tw, 1)}
del rec
import subprocess
print("BAM written: %.2f GB in %.1f s" % (size / 1e9, tw), flush=True)
t0 = time.perf_counter()
c = bam.decode_bam(p)
tc = time.perf_counter() - t0
print("host decode: %.2f s" % tc, flush=True)
out["cpu"] = {"seconds": round(tc, 3), "reads_per_s": n / tc, "threads": bam.LAST_DECODE["threads"], "GB_per_s_inflated": bam.LAST_DECODE["uncompressed_bytes"] / tc / 1e9}
ref_sum = int(c.cigar.to(torch.int64).sum())
del c
best = None
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = bam.decode_bam_gpu(p, "cuda:0", batch_bytes=batch)
    torch.cuda.synchronize()
    tg = time.perf_counter() - t0
    st = dict(bam.LAST_DECODE)
    assert int(g.cigar.to(torch.int64).sum()) == ref_sum
    del g
    row = {"seconds": round(tg, 3), "reads_per_s": n / tg, "GB_per_s_inflated": st["uncompressed_bytes"] / tg / 1e9,
           "GB_per_s_compressed": st["compressed_bytes"] / tg / 1e9, "batches": st["batches"], "rewalked_segments": st["rewalked_segments"],
           "host_seconds": round(st["host_seconds"], 3), "nonacgt_records_fetched": st["nonacgt_records_fetched"], "native_seconds": round(st["seconds"], 3),
           "read_seconds": round(st["read_seconds"], 3), "setup_seconds": round(st["setup_seconds"], 3),
           "waited_for_file_seconds": round(st["waited_for_file_seconds"], 3), "waited_for_gpu_seconds": round(st["waited_for_gpu_seconds"], 3)}
    out.setdefault("gpu_runs", []).append(row)
    if best is None or tg < best["seconds"]:
        best = row
out["gpu"] = best
print(json.dumps(out))
import shutil; shutil.rmtree(d, ignore_errors=True)
