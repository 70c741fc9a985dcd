"""Micro-benchmark of coral_cigar_scan on GPU-generated synthetic records (not part of the test suite)."""
import sys, time
import ctypes as C
import torch
sys.path.insert(0, ".")
from coral_amd import synth, kernels, _lib
from coral_amd.records import DeviceRecords

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
cfg = synth.scaled_config(name, n)
t = time.time()
rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000)
torch.cuda.synchronize()
print("generate %.1fs  records %d  ops %d  cigar bytes %.2f GB" % (time.time() - t, rec.n, int(rec.n_cigar.sum()), rec.cigar.numel() * 4 / 1e9), flush=True)
dr = DeviceRecords(rec, "cuda:0")
L0 = _lib.lib()
L0.coral_set_scan_variant(1); r1 = kernels.cigar_scan(dr)
L0.coral_set_scan_variant(2); res = kernels.cigar_scan(dr)
import numpy as np
assert torch.equal(r1.mbases, res.mbases) and torch.equal(r1.qinfer, res.qinfer) and torch.equal(r1.blk_first, res.blk_first) and torch.equal(r1.blk_last, res.blk_last) and np.array_equal(r1.gaps, res.gaps), "variants disagree"
assert L0.coral_set_scan_variant(7) == 0; r7 = kernels.cigar_scan(dr)
assert torch.equal(r7.mbases, res.mbases) and torch.equal(r7.qinfer, res.qinfer) and torch.equal(r7.blk_first, res.blk_first) and torch.equal(r7.blk_last, res.blk_last) and np.array_equal(r7.gaps, res.gaps), "filtered variant disagrees"
for v in (15, 25, 26, 27):
    assert L0.coral_set_scan_variant(v) == 0; r8 = kernels.cigar_scan(dr)
    assert torch.equal(r8.mbases, res.mbases) and torch.equal(r8.qinfer, res.qinfer) and torch.equal(r8.blk_first, res.blk_first) and torch.equal(r8.blk_last, res.blk_last) and np.array_equal(r8.gaps, res.gaps), "packed variant disagrees"
L0.coral_set_scan_variant(6); r6 = kernels.cigar_scan(dr)
assert torch.equal(r6.mbases, res.mbases) and torch.equal(r6.qinfer, res.qinfer) and torch.equal(r6.blk_first, res.blk_first) and torch.equal(r6.blk_last, res.blk_last) and np.array_equal(r6.gaps, res.gaps), "flat variant disagrees"
print("gaps", res.gaps.shape, "variants agree (1, 2, 6)")
L = _lib.lib()
rs = dr.c_struct()
mb = torch.empty(dr.n, dtype=torch.int32, device="cuda"); qi = torch.empty_like(mb); b0 = torch.empty_like(mb); b1 = torch.empty_like(mb)
gaps = torch.empty((1 << 20, 4), dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
ms = C.c_float(0)
VARIANTS = [15, 25, 26, 27]
for it in range(3 * len(VARIANTS)):
    L.coral_set_scan_variant(VARIANTS[it % len(VARIANTS)])
    _lib.check(L.coral_time_cigar_scan(C.byref(rs), 600, 20, mb.data_ptr(), qi.data_ptr(), b0.data_ptr(), b1.data_ptr(), gaps.data_ptr(), cnt.data_ptr(), 1 << 20, 10, C.byref(ms), dr.stream()), "time")
    B = dr.algorithmic_bytes()
    print("variant %d" % VARIANTS[it % len(VARIANTS)], "scan %.3f ms/launch  alg bytes %.3f GB  -> %.1f GB/s (%.1f%% of 8 TB/s)" % (ms.value, B / 1e9, B / ms.value / 1e6, B / ms.value / 1e6 / 80), flush=True)

scr = torch.zeros(4, dtype=torch.int32, device="cuda")
for it in range(4):
    L.coral_set_probe_mode(1 + it % 2)
    print("probe mode %d:" % (1 + it % 2), end=" ")
    _lib.check(L.coral_time_stream_read(dr.cigar.data_ptr(), dr.cigar.numel(), scr.data_ptr(), 10, C.byref(ms), dr.stream()), "stream")
    L.coral_set_scan_variant(2)
    print("plain streaming read %.3f ms  -> %.1f GB/s" % (ms.value, dr.cigar.numel() * 4 / ms.value / 1e6), flush=True)

import time as _t
L.coral_set_scan_variant(int(sys.argv[3]) if len(sys.argv) > 3 else 3)
for idle in (0.0, 0.2, 1.0, 1.0):
    _t.sleep(idle)
    _lib.check(L.coral_time_cigar_scan(C.byref(rs), 600, 20, mb.data_ptr(), qi.data_ptr(), b0.data_ptr(), b1.data_ptr(), gaps.data_ptr(), cnt.data_ptr(), 1 << 20, 1, C.byref(ms), dr.stream()), "time")
    print("single launch after %.1fs idle: %.3f ms -> %.1f GB/s" % (idle, ms.value, B / ms.value / 1e6), flush=True)
for idle in (1.0, 1.0):
    _t.sleep(idle)
    scr2 = torch.zeros(1 << 24, device="cuda"); scr2.add_(1.0); scr2.add_(1.0)      # ~0.1 ms of unrelated GPU work right before
    _lib.check(L.coral_time_cigar_scan(C.byref(rs), 600, 20, mb.data_ptr(), qi.data_ptr(), b0.data_ptr(), b1.data_ptr(), gaps.data_ptr(), cnt.data_ptr(), 1 << 20, 1, C.byref(ms), dr.stream()), "time")
    print("single launch after %.1fs idle + small torch op: %.3f ms -> %.1f GB/s" % (idle, ms.value, B / ms.value / 1e6), flush=True)
