"""The DEFLATE core of the GPU BGZF path (coral_amd/csrc/coral_inflate_core.h) compiled for the HOST (tests/native/
inflate_host.cpp: same decode logic, lanes as loops) against zlib: every block type, levels and strategies, several blocks per
stream, corrupt streams (an error or a wrong size, never a crash or a hang).  The device build of the same code is checked
against zlib in tests/test_bam_gpu.py."""
import ctypes as C
import os
import random
import struct
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def inflate(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("inflate") / "libinflate_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "native", "inflate_host.cpp")], check=True)
    L = C.CDLL(so)

    def run(comp, n):
        """All symbol loops (1: what the device's general loop runs; 2: the same behind a fast path that hands over in every
        state, as the device's hand-written loop does; 0: the plain loop); they must agree on data and on success."""
        res = []
        for paired in (1, 0, 2):
            out, prod = C.create_string_buffer(max(n, 1) + 64), C.c_int(0)
            rc = L.coral_test_inflate(comp, len(comp), out, n, C.byref(prod), paired)
            assert out.raw[n:] == bytes(len(out.raw) - n), "wrote beyond the capacity"
            res.append((rc, out.raw[:prod.value]))
        for other in res[1:]:
            assert (res[0][0] == 0) == (other[0] == 0) and (res[0][0] != 0 or res[0] == other)
        return res[0]
    return run


def test_core_equals_zlib(inflate):
    rnd = random.Random(1)
    cases = [b"", b"a", b"hello hello hello hello", bytes(65280), b"\xff" * 65280, os.urandom(65280),
             bytes(rnd.choice(b"ACGT") for _ in range(65280)), bytes(rnd.getrandbits(8) & 0x33 for _ in range(30000)),
             b"".join(b"%d,%d;" % (rnd.randrange(1000), rnd.randrange(10 ** 6)) for _ in range(5000))[:65280],
             b"".join(struct.pack("<I", (rnd.randrange(1, 40) << 4) | rnd.choice([0, 0, 0, 1, 2])) for _ in range(16000))]
    for n in (1, 2, 3, 5, 100, 1000, 40000):
        cases += [os.urandom(n), bytes(rnd.choice(b"ab") for _ in range(n))]
    for data in cases:
        for level in (0, 1, 3, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strat)
                comp = co.compress(data) + co.flush()
                assert inflate(comp, len(data)) == (0, data), (len(data), level, strat)


def test_core_on_the_gpu_test_streams(inflate):
    """The very streams the inflate kernel is checked on (tests/test_bam_gpu.py)."""
    from tests.deflate_streams import streams
    for comp, data in streams():
        assert inflate(comp, len(data)) == (0, data)


def test_core_multi_block_streams(inflate):
    rnd = random.Random(2)
    for _ in range(40):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        data, comp = b"", b""
        for _k in range(rnd.randrange(1, 6)):
            piece = os.urandom(rnd.randrange(0, 3000)) if rnd.random() < 0.5 else bytes(rnd.choice(b"ACGTN") for _ in range(rnd.randrange(0, 9000)))
            data += piece
            comp += co.compress(piece) + co.flush(rnd.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH]))
        comp += co.flush()
        assert inflate(comp, len(data)) == (0, data)


def test_core_survives_corrupt_streams(inflate):
    rnd = random.Random(3)
    for _ in range(400):
        data = bytes(rnd.choice(b"ACGT") for _ in range(5000))
        comp = bytearray(zlib.compress(data, 6)[2:-4])
        for _k in range(rnd.randrange(1, 4)):
            comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        rc, got = inflate(bytes(comp), len(data))
        try:
            ok = zlib.decompress(bytes(comp), -15) == data
        except zlib.error:
            ok = False
        if ok:
            assert (rc, got) == (0, data)
        assert rc != 0 or len(got) == len(data)
    for cut in (0, 1, 5, 50):                      # truncated input
        comp = zlib.compress(os.urandom(3000), 6)[2:-4]
        rc, got = inflate(comp[:cut], 3000)
        assert rc != 0
