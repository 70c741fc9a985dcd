"""Target for the rocprofv3 --pmc passes: config-3 records on the GPU, then ONLY libcoral_hip kernels a few times."""
import sys
sys.path.insert(0, ".")
import torch
from coral_amd import synth, kernels
from coral_amd.records import DeviceRecords
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
name = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
cfg = synth.scaled_config(name, n)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000)
dr = DeviceRecords(rec, "cuda:0")
from coral_amd import _lib
print("records", dr.n, "alg bytes", dr.algorithmic_bytes(), flush=True)
print("library", _lib.lib().coral_version().decode(), flush=True)
for _ in range(3):
    sc = kernels.cigar_scan(dr)
t, ws, we = cfg.windows[1]
segs = [(t, ws + k * 250000, ws + (k + 1) * 250000) for k in range((we - ws) // 250000)]
for _ in range(2):
    kernels.segment_coverage(dr, sc, segs)
torch.cuda.synchronize()
print("done", flush=True)
