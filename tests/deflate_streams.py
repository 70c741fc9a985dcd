"""Raw DEFLATE test streams (with their plain text) shared by the host test of the decoder core (tests/test_inflate_core.py) and
the GPU test of the inflate kernel (tests/test_bam_gpu.py): every block type, levels and strategies, sizes around the 64-lane
boundaries, two-letter texts (very long matches, code sets with 11- and 12-bit codes), several blocks per stream."""
import os
import random
import struct
import zlib


def streams():
    rnd = random.Random(7)
    cases = [b"", b"a", b"hello hello hello hello", bytes(65280), b"\xff" * 65280, os.urandom(65280),
             bytes(rnd.choice(b"ACGT") for _ in range(65280)), bytes(rnd.getrandbits(8) & 0x33 for _ in range(30000)),
             b"".join(b"%d,%d;" % (rnd.randrange(1000), rnd.randrange(10 ** 6)) for _ in range(5000))[:65280],
             b"".join(struct.pack("<I", (rnd.randrange(1, 40) << 4) | rnd.choice([0, 0, 0, 1, 2])) for _ in range(16000))]
    # Fibonacci-like symbol frequencies: Huffman code lengths up to the 15-bit limit (the canonical loop behind the 10-bit table)
    fib, a, b = [], 1, 1
    for sym in range(24):
        fib.append(bytes([65 + sym]) * a)
        a, b = b, a + b
    skew = bytearray(b"".join(fib))
    rnd.shuffle(skew)
    cases.append(bytes(skew[:65000]))
    for n in (1, 2, 3, 5, 63, 64, 65, 100, 1000, 40000):
        cases.append(os.urandom(n))
        cases.append(bytes(rnd.choice(b"ab") for _ in range(n)))
    out = []
    for data in cases:
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
                out.append((co.compress(data) + co.flush(), data))
    for _ in range(20):                                 # several DEFLATE blocks per stream, empty stored blocks in between
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        data, comp = b"", b""
        for _k in range(rnd.randrange(1, 6)):
            piece = os.urandom(rnd.randrange(0, 3000)) if rnd.random() < 0.5 else bytes(rnd.choice(b"ACGTN") for _ in range(rnd.randrange(0, 9000)))
            data += piece
            comp += co.compress(piece) + co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH]))
        out.append((comp + co.flush(), data))
    return out


