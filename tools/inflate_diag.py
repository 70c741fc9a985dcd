"""Diagnostics for k_bgzf_inflate against zlib: which of the test streams fail, with what status, where the output first differs."""
import sys
sys.path.insert(0, ".")
import zlib
from tests.test_bam_gpu import _streams, _inflate_on_gpu
pairs = _streams()
out, status, desc = _inflate_on_gpu(pairs)
nbad = 0
for k, ((c, d), (_, _, o, n), st) in enumerate(zip(pairs, desc, status)):
    got = out[o:o + n]
    if st != 0 or got != d:
        nbad += 1
        first = next((i for i in range(n) if got[i] != d[i]), -1)
        print("stream %d: in %d B -> out %d B, status %d, first diff at %d (dst offset %d, dst %% 256 = %d); data head %r" % (
            k, len(c), n, st, first, o, o % 256, d[:16]))
        if first >= 0:
            print("   expected", d[max(0, first - 8):first + 24].hex(), "\n   got     ", got[max(0, first - 8):first + 24].hex())
print("streams", len(pairs), "bad", nbad)
