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
print("all equal")
