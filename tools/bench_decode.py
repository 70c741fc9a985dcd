"""BAM decode throughput of coral_bam_decode_* on the host cores (outside the graded kernels)."""
import os, sys, time, tempfile
sys.path.insert(0, ".")
from coral_amd import synth, bam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
cfg = synth.scaled_config("cfg3", n)
rec = synth.generate(cfg, "cpu")
d = tempfile.mkdtemp()
p = os.path.join(d, "x.bam")
t = time.time(); bam.write_bam(rec, p, fast_seq=True); tw = time.time() - t
sz = os.path.getsize(p)
bases = int(rec.qlen[rec.has_seq.bool()].sum())
print("wrote %d records, %.1f MB BAM, %.2f Gbases in %.1fs" % (rec.n, sz / 1e6, bases / 1e9, tw), flush=True)
for nt in (1, 4, 8, 16):
    if nt > (os.cpu_count() or 1):
        break
    t = time.time(); back = bam.decode_bam(p, n_threads=nt); dt = time.time() - t
    print("decode threads=%2d  %.2fs  %.0f MB/s compressed  %.0f reads/s  %.2f Gbases/s" % (nt, dt, sz / dt / 1e6, n / dt, bases / dt / 1e9), flush=True)
assert back.n == rec.n and bool((back.cigar == rec.cigar).all())
