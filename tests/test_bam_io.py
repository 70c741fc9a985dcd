"""Native BAM decoder (host entry points of the C ABI) round trip against the pure-Python writer, plus an independent
pure-Python read of the same file through the gzip module (BGZF is multi-member gzip)."""
import gzip
import struct

import numpy as np
import pytest
import torch

from coral_amd import bam, synth

FIELDS = ("tid", "pos", "end", "flag", "mapq", "qlen", "has_seq", "nm", "name_id", "n_cigar", "cigar_off", "cigar",
          "sa_off", "sa", "sa_nm", "nonacgt_rec", "nonacgt_pos")


def assert_same(a, b):
    assert a.n == b.n
    for k in FIELDS:
        x, y = getattr(a, k).cpu().numpy(), getattr(b, k).cpu().numpy()
        assert x.shape == y.shape and np.array_equal(x, y), k
    assert a.materialise_names() == b.materialise_names()
    assert a.header_chroms == b.header_chroms and a.header_lens == b.header_lens


def test_roundtrip_synthetic_tiny(tmp_path):
    rec = synth.generate(synth.scaled_config("tiny", 600), "cpu")
    p = str(tmp_path / "t.bam")
    bam.write_bam(rec, p, seed=11)
    back = bam.decode_bam(p, n_threads=3)
    assert_same(rec, back)
    assert int((rec.sa_off[1:] - rec.sa_off[:-1] > 0).sum()) > 0 and rec.nonacgt_pos.numel() > 0


def _odd():
    M, I, D, N, S, H, P, EQ, X = range(9)
    big = [(M, 3), (I, 1)] * 33000 + [(M, 5)]            # 66001 ops -> CG tag path
    return synth.records_from_alignments([
        dict(tid=0, pos=100, cigar=[(S, 5), (M, 50), (D, 700), (M, 20), (I, 3), (M, 10)], name="a", nm=7,
             sa=[(7, 1000, 1, 10, 2000, -30, 55, 60, 12), (11, 5, 0, 0, 300, 4, 9000, 3, 0)], nonacgt=[101, 860]),
        dict(tid=0, pos=120, cigar=[(H, 9), (EQ, 10), (X, 2), (N, 900), (M, 30), (H, 7)], name="b", flag=2064, mapq=0),
        dict(tid=0, pos=130, cigar=[], flag=4, name="c", qlen=40),
        dict(tid=0, pos=180, cigar=[(M, 200)], has_seq=0, flag=256, name="a"),
        dict(tid=3, pos=7, cigar=big, name="long"),
        dict(tid=24, pos=16000, cigar=[(M, 500)], name="mito"),
    ])


def test_roundtrip_odd_records_and_long_cigar(tmp_path):
    rec = _odd()
    p = str(tmp_path / "odd.bam")
    bam.write_bam(rec, p)
    back = bam.decode_bam(p, n_threads=2)
    assert_same(rec, back)
    assert int(back.n_cigar.max()) == 66001


def test_file_is_standard_bam(tmp_path):
    """Independent check of the writer: gzip module + struct parsing of header and first record."""
    rec = _odd()
    p = str(tmp_path / "odd.bam")
    bam.write_bam(rec, p)
    raw = gzip.open(p, "rb").read()
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    assert n_ref == 25
    o += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, o)[0]
        o += 8 + ln
    bs, tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiiBBHHHi", raw, o)
    assert (tid, pos, mapq, n_cig, flag, l_seq) == (0, 100, 60, 6, 0, 88)
    name = raw[o + 36: o + 36 + l_name]
    assert name == b"a\0"
    cig = np.frombuffer(raw, dtype="<u4", count=n_cig, offset=o + 36 + l_name)
    assert [(int(c & 15), int(c >> 4)) for c in cig] == [(4, 5), (0, 50), (2, 700), (0, 20), (1, 3), (0, 10)]
    seq = raw[o + 36 + l_name + 4 * n_cig: o + 36 + l_name + 4 * n_cig + 44]
    codes = [(b >> 4, b & 15) for b in seq]
    flat = [c for pair in codes for c in pair][:88]
    assert flat[5 + 1] == 15 and all(c in (1, 2, 4, 8, 15) for c in flat)      # the N planted at ref 101 = query offset 6
    assert b"SAZchr8,1000,-,10S2000M30D55S,60,12;chr12,5,+,300M4I9000S,3,0;\0" in raw[o: o + 4 + bs]


def test_decoder_rejects_garbage(tmp_path):
    from coral_amd import _lib
    p = tmp_path / "bad.bam"
    p.write_bytes(b"this is not a bam file at all, not even gzip")
    with pytest.raises(_lib.CoralHipError):
        bam.decode_bam(str(p))
    with pytest.raises(_lib.CoralHipError):
        bam.decode_bam(str(tmp_path / "missing.bam"))


def test_sa_shapes_tokenised(tmp_path):
    """SA CIGARs outside [S]M[I|D][S]: no S / no M -> zero clips or zero M (the read fails as a whole later,
    cigar_parsing.py:248-253); containing S and M but another shape -> leading clip = -2 (KeyError in the reference,
    cigar_parsing.py:255)."""
    M, S = 0, 4
    rec = synth.records_from_alignments([dict(tid=0, pos=10, cigar=[(S, 5), (M, 50)], name="x")])
    rec.sa_text = {0: "chr1,500,+,7S40M,60,1;chr2,9,-,47M,3,2;chr3,8,+,7H40M,1,0;chr4,7,-,1S2M3S4M,0,5;chrZ,6,+,3S4M2I1S,9,9;"}
    p = str(tmp_path / "s.bam")
    bam.write_bam(rec, p)
    back = bam.decode_bam(p)
    assert back.sa.tolist() == [[0, 500, 0, 7, 40, 0, 0, 60], [1, 9, 1, 0, 0, 0, 0, 3], [2, 8, 0, 0, 0, 0, 0, 1],
                                [3, 7, 1, -2, 2, 0, 3, 0], [-1, 6, 0, 3, 4, 2, 1, 9]]
    assert back.sa_nm.tolist() == [1, 2, 0, 5, 9]
