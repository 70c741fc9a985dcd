"""Target for rocprofv3 --pmc passes over the scan-kernel variants and the plain-read probes (one launch each, twice)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from coral_amd import synth, kernels, _lib
from coral_amd.records import DeviceRecords
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
cfg = synth.scaled_config("cfg3", n)
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000)
dr = DeviceRecords(rec, "cuda:0")
L = _lib.lib()
print("records", dr.n, "alg bytes", dr.algorithmic_bytes(), flush=True)
ms = C.c_float(0)
scr = torch.zeros(4, dtype=torch.int32, device="cuda")
for rep in range(2):
    for v in (3, 5, 7, 13, 15, 19):
        assert L.coral_set_scan_variant(v) == 0
        kernels.cigar_scan(dr)
    for m in (1, 2):
        L.coral_set_probe_mode(m)
        _lib.check(L.coral_time_stream_read(dr.cigar.data_ptr(), dr.cigar.numel(), scr.data_ptr(), 1, C.byref(ms), dr.stream()), "probe")
L.coral_set_scan_variant(15)
torch.cuda.synchronize()
print("done", flush=True)
