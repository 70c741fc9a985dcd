"""Decode throughput of coral_bam_decode_* on a synthetic BAM written with the native writer (not part of the test suite)."""
import os, sys, time, tempfile
sys.path.insert(0, ".")
from coral_amd import synth, bam
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
threads = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [bam.default_threads()]
cfg = synth.scaled_config(name, n)
rec = synth.generate(cfg, "cpu")
path = os.path.join(tempfile.mkdtemp(), "x.bam")
t = time.time(); bam.write_bam_native(rec, path, seed=1); tw = time.time() - t
size = os.path.getsize(path)
print("wrote %d records (%d reads), %.2f GB in %.1f s" % (rec.n, n, size / 1e9, tw), flush=True)
for nt in threads:
    for rep in range(2):
        t = time.time(); back = bam.decode_bam(path, n_threads=nt); dt = time.time() - t
        st = bam.LAST_DECODE
        print("threads %2d: %.2f s  %.0f reads/s  %.2f GB/s compressed  %.2f GB/s inflated (native part %.2f s)" % (
            nt, dt, n / dt, size / dt / 1e9, st["uncompressed_bytes"] / dt / 1e9, st["seconds"]), flush=True)
assert back.n == rec.n
