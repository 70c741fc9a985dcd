"""Repeated GPU decodes of one BAM with varying batch sizes, every result compared with the host decoder's (races between the
feeder thread, the host-side worker and the caller would show as differing fields).   python tools/stress_decode_gpu.py [reads] [rounds]"""
import os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np, torch
from coral_amd import bam, synth
from tests.test_bam_io import FIELDS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rec = synth.generate(synth.scaled_config("cfg3", n), "cuda:0", chunk_pieces=200000).to("cpu")
p = os.path.join(tempfile.mkdtemp(prefix="coral_stress_"), "x.bam")
bam.write_bam_native(rec, p, seed=1)
ref = bam.decode_bam(p)
refs = {k: getattr(ref, k).cpu().numpy() for k in FIELDS}
for r in range(rounds):
    batch = [0, 64 << 20, 16 << 20, 200 << 20, 4 << 20][r % 5]
    t0 = time.perf_counter()
    g = bam.decode_bam_gpu(p, "cuda:0", batch_bytes=batch, n_threads=[16, 3, 1, 8][r % 4])
    dt = time.perf_counter() - t0
    assert g.n == ref.n and g.names == ref.names
    for k in FIELDS:
        assert np.array_equal(getattr(g, k).cpu().numpy(), refs[k]), (r, k)
    print("round %2d: batch %9d  %.3f s  batches %d  ok" % (r, batch, dt, bam.LAST_DECODE["batches"]), flush=True)
# other shapes of data: ultra-long reads (records of several BGZF blocks), short ones (thousands per block), other zlib levels
for name, reads, level in (("cfg5", 4000, 6), ("ultra", 600, 9), ("tiny", 20000, 6), ("cfg1", 8000, 0), ("cfg2", 8000, 4)):
    rec = synth.generate(synth.scaled_config(name, reads), "cuda:0", chunk_pieces=200000).to("cpu")
    bam.write_bam_native(rec, p, seed=2, level=level)
    ref = bam.decode_bam(p)
    for batch in (0, 8 << 20):
        g = bam.decode_bam_gpu(p, "cuda:0", batch_bytes=batch)
        assert g.n == ref.n and g.names == ref.names
        for k in FIELDS:
            assert np.array_equal(getattr(g, k).cpu().numpy(), getattr(ref, k).cpu().numpy()), (name, level, batch, k)
    print("%s, %d reads, zlib level %d: %d records, %.2f GB inflated, equal (whole file and 8 MiB batches)" % (
        name, reads, level, ref.n, bam.LAST_DECODE["uncompressed_bytes"] / 1e9), flush=True)
print("all equal")
