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
    out += crafted_streams()
    for _ in range(20):                                 # several DEFLATE blocks per stream, empty stored blocks in between
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        data, comp = b"", b""
        for _k in range(rnd.randrange(1, 6)):
            piece = os.urandom(rnd.randrange(0, 3000)) if rnd.random() < 0.5 else bytes(rnd.choice(b"ACGTN") for _ in range(rnd.randrange(0, 9000)))
            data += piece
            comp += co.compress(piece) + co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH]))
        out.append((comp + co.flush(), data))
    return out


# ---- streams written token by token (fixed Huffman codes, RFC 1951 §3.2.6): matches of exactly chosen length and distance,
# around every boundary of the GPU kernel's hand-written loop (coral_bamgpu.hip, DevWaveT::fast): the 2 KiB ring and its
# "source has left the ring" limit (distance 1984 / 1985), 64-byte copy chunks, matches that overlap their source, the longest
# match, the 256-byte output lines, the last 260 bytes of a block.
_LBASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
_LEXT = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
_DBASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577]
_DEXT = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):                       # LSB first (extra bits, header fields)
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 0xff)
            self.acc >>= 8
            self.n -= 8

    def code(self, value, nbits):                      # a Huffman code: most significant bit first
        self.put(int(format(value, "0%db" % nbits)[::-1], 2), nbits)

    def done(self):
        if self.n:
            self.put(0, 8 - self.n)
        return bytes(self.out)


def fixed_block(tokens):
    """tokens: ('L', byte) | ('M', length, distance)  ->  (raw DEFLATE stream of one final fixed-Huffman block, its plain text)."""
    b, text = _Bits(), bytearray()
    b.put(1, 1)
    b.put(1, 2)

    def sym(x):
        if x < 144: b.code(0x30 + x, 8)
        elif x < 256: b.code(0x190 + x - 144, 9)
        elif x < 280: b.code(x - 256, 7)
        else: b.code(0xc0 + x - 280, 8)
    for t in tokens:
        if t[0] == "L":
            sym(t[1])
            text.append(t[1])
        else:
            _, n, d = t
            assert 3 <= n <= 258 and 1 <= d <= len(text) and d <= 32768
            i = max(k for k in range(29) if _LBASE[k] <= n)
            if n == 258: i = 28
            sym(257 + i)
            b.put(n - _LBASE[i], _LEXT[i])
            j = max(k for k in range(30) if _DBASE[k] <= d)
            b.code(j, 5)
            b.put(d - _DBASE[j], _DEXT[j])
            for _k in range(n):
                text.append(text[-d])
    sym(256)
    return b.done(), bytes(text)


def crafted_streams():
    rnd = random.Random(17)
    out = []
    lens = (3, 4, 5, 63, 64, 65, 66, 127, 128, 129, 257, 258)
    dists = (1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 1983, 1984, 1985, 2047, 2048, 2049, 4031, 4032, 4033, 4999)
    for order in range(3):
        toks = [("L", rnd.randrange(256)) for _ in range(5000)]            # history longer than any ring
        combos = [(n, d) for n in lens for d in dists]
        rnd.shuffle(combos)
        for n, d in combos:
            toks.append(("M", n, d))
            for _ in range(rnd.randrange(0, 3) if order else 1):           # literal runs of 0..2 between the matches
                toks.append(("L", rnd.randrange(256)))
        out.append(fixed_block(toks))
    # matches only (no literal in between), every output-line phase; the block ends with a match
    toks = [("L", 65 + k % 7) for k in range(300)]
    for k in range(400):
        toks.append(("M", 3 + (k * 7) % 256, 1 + (k * 13) % 299))
    out.append(fixed_block(toks))
    # very short blocks and blocks shorter than the 260 bytes the hand-written loop leaves to the general one
    for n in (1, 2, 3, 100, 259, 260, 261, 300, 600):
        toks = [("L", rnd.randrange(256)) for _ in range(min(n, 40))]
        while sum(1 if t[0] == "L" else t[1] for t in toks) + 3 <= n:
            left = n - sum(1 if t[0] == "L" else t[1] for t in toks)
            have = sum(1 if t[0] == "L" else t[1] for t in toks)
            toks.append(("M", min(left, rnd.choice((3, 17, 70, 258))), rnd.randrange(1, have + 1)))
        out.append(fixed_block(toks))
    for comp, text in out:
        assert zlib.decompress(comp, -15) == text
    return out
