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


def test_native_writer_roundtrip_and_is_standard_bam(tmp_path):
    """coral_bam_write -> coral_bam_decode gives the records back; the file is multi-member gzip that the gzip module reads,
    with the BAM magic, and the odd records (CG-tag CIGAR, unmapped, no SEQ, non-ACGT bases) survive."""
    for name, rec in (("synthetic", synth.generate(synth.scaled_config("tiny", 1500), "cpu")), ("odd", _odd())):
        p = str(tmp_path / (name + ".bam"))
        bam.write_bam_native(rec, p, seed=5, n_threads=3)
        assert_same(rec, bam.decode_bam(p, n_threads=3))
        raw = gzip.open(p, "rb").read()
        assert raw[:4] == b"BAM\x01"
    assert raw.count(b"SAZ") >= 1


def _concat(parts):
    """Records of consecutive byte ranges put together again (read-name ids are local to a range: compare by name)."""
    out = {}
    for k in FIELDS:
        if k in ("cigar_off", "sa_off", "nonacgt_rec", "name_id"):
            continue
        out[k] = np.concatenate([getattr(p, k).cpu().numpy() for p in parts])
    out["n_cigar_padded"] = np.concatenate([np.diff(p.cigar_off.cpu().numpy()) for p in parts])
    out["sa_count"] = np.concatenate([np.diff(p.sa_off.cpu().numpy()) for p in parts])
    base, na = 0, []
    for p in parts:
        na.append(p.nonacgt_rec.cpu().numpy() + base)
        base += p.n
    out["nonacgt_rec"] = np.concatenate(na)
    out["names"] = [p.names[i] for p in parts for i in p.name_id.tolist()]
    return out


@pytest.mark.parametrize("config,n_reads", [("tiny", 3000), ("ultra", 400)])
def test_byte_ranges_partition_the_file(config, n_reads, tmp_path):
    """Decoding the file as `world` byte ranges (one per GPU process) gives exactly the records of the whole-file decode, in
    order, none dropped, none repeated — for short reads (many records per BGZF block) and for ultra-long ones (records of
    several blocks straddling range boundaries), for any number of ranges and threads."""
    cfg = synth.scaled_config(config, n_reads)
    rec = synth.generate(cfg, "cpu")
    p = str(tmp_path / "x.bam")
    bam.write_bam_native(rec, p, seed=3, n_threads=4)
    whole = _concat([bam.decode_bam(p, n_threads=2)])
    assert len(whole["tid"]) == rec.n
    for world in (2, 3, 5, 8, 13):
        parts = [bam.decode_bam(p, n_threads=1 + (r % 3), rank=r, world=world) for r in range(world)]
        got = _concat(parts)
        assert sum(q.n for q in parts) == rec.n, world
        for k, v in whole.items():
            assert (list(v) == list(got[k])) if k == "names" else np.array_equal(v, got[k]), (world, k)
        assert all(q.header_chroms == rec.header_chroms for q in parts)


def test_decoder_rejects_garbage(tmp_path):
    from coral_amd._lib import CoralHipError
    p = tmp_path / "junk.bam"
    p.write_bytes(b"this is not a BGZF file" * 10)
    with pytest.raises(CoralHipError):
        bam.decode_bam(str(p))
    rec = synth.generate(synth.scaled_config("tiny", 300), "cpu")
    good = tmp_path / "good.bam"
    bam.write_bam_native(rec, str(good))
    data = bytearray(good.read_bytes())
    data[len(data) // 2: len(data) // 2 + 64] = b"\x00" * 64            # corrupt a block in the middle
    bad = tmp_path / "bad.bam"
    bad.write_bytes(bytes(data))
    with pytest.raises(CoralHipError):
        bam.decode_bam(str(bad))
    trunc = tmp_path / "trunc.bam"
    trunc.write_bytes(bytes(good.read_bytes()[:-5000]))
    with pytest.raises(CoralHipError):
        bam.decode_bam(str(trunc))
    # a byte flipped inside a STORED block: the block still inflates, only the trailer's CRC-32 tells
    stored = tmp_path / "stored.bam"
    bam.write_bam_native(rec, str(stored), level=0)
    data = bytearray(stored.read_bytes())
    data[len(data) // 2] ^= 0x40
    flipped = tmp_path / "flipped.bam"
    flipped.write_bytes(bytes(data))
    with pytest.raises(CoralHipError):
        bam.decode_bam(str(flipped))
    assert bam.decode_bam(str(stored)).n == rec.n
