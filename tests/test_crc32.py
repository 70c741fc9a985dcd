"""coral_amd/csrc/coral_crc32.h (the checksum of the GPU decoder's k_bgzf_crc: per-lane chunk remainders combined by
multiplication with x^(8 n) mod P) against zlib.crc32, on the host."""
import ctypes as C
import os
import random
import subprocess
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chunked_crc32_equals_zlib(tmp_path):
    so = str(tmp_path / "libcrc_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "native", "crc_host.cpp")], check=True)
    L = C.CDLL(so)
    L.coral_test_crc32.restype = C.c_uint32
    rnd = random.Random(4)
    for n in list(range(0, 70)) + [255, 256, 257, 1000, 4095, 65279, 65280, 65536] + [rnd.randrange(70000) for _ in range(40)]:
        data = os.urandom(n)
        for chunks in (64, 1, 7):
            assert L.coral_test_crc32(data, n, chunks) == (zlib.crc32(data) & 0xffffffff), (n, chunks)
