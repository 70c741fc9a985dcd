"""Launch time of coral_cigar_scan on GPU-generated synthetic records, back to back and after idle, next to a plain read of the
same bytes by a library reduction (not part of the test suite)."""
import sys, time
import torch
sys.path.insert(0, ".")
from coral_amd import synth, kernels, _lib
from coral_amd.records import DeviceRecords

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
cfg = synth.scaled_config(name, n)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000)
torch.cuda.synchronize()
dr = DeviceRecords(rec, "cuda:0")
del rec
B = dr.algorithmic_bytes()
print("kernel %s  records %d  ops %d  algorithmic bytes %.3f GB" % (_lib.lib().coral_scan_kernel_name().decode(), dr.n, dr.total_ops, B / 1e9), flush=True)
ref = kernels.cigar_scan(dr)
print("gap rows", ref.gaps.shape[0])

def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps

for rep in range(3):
    ms = timed(lambda: kernels._scan_local(dr, 600, 20, 1 << 16), 10)
    print("back to back: %.3f ms/launch -> %.1f GB/s (%.1f%% of 8 TB/s)" % (ms, B / ms / 1e6, B / ms / 1e6 / 80), flush=True)
for rep in range(2):
    ms = timed(lambda: dr.cigar.view(torch.int64).sum(), 5)
    print("torch sum of the CIGAR bytes: %.3f ms -> %.1f GB/s" % (ms, dr.cigar.numel() * 4 / ms / 1e6), flush=True)
for idle in (0.2, 1.0):
    time.sleep(idle)
    ms = timed(lambda: kernels._scan_local(dr, 600, 20, 1 << 16), 1)
    print("single launch after %.1f s idle: %.3f ms -> %.1f GB/s" % (idle, ms, B / ms / 1e6), flush=True)
