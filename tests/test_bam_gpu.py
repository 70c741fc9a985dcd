"""BAM decode on the GPU (csrc/coral_bamgpu.hip): the inflate kernel against zlib, and the whole pipeline against the CPU
pipeline (coral_bam_decode_*, itself pinned by tests/test_bam_io.py) field by field — whole files, byte ranges, batches small
enough that records straddle them, odd records, corrupt input."""
import ctypes as C
import os
import random
import struct
import zlib

import numpy as np
import pytest
import torch

from coral_amd import bam, synth, _lib
from tests.test_bam_io import FIELDS, _concat, _odd, assert_same

pytestmark = pytest.mark.gpu


from tests.deflate_streams import streams as _streams


def _inflate_on_gpu(pairs, pad=0):
    """One coral_bgzf_inflate launch over all streams (stream k starts `pad + k % 4` bytes after the previous one: every
    alignment of input and output occurs)."""
    L = _lib.lib()
    comp, desc, o_in, o_out = bytearray(), [], 0, 0
    for k, (c, d) in enumerate(pairs):
        gap = pad + (k % 4)
        comp += b"\xaa" * gap
        o_in += gap
        o_out += k % 3
        desc.append((o_in, len(c), o_out, len(d)))
        comp += c
        o_in += len(c)
        o_out += len(d)
    comp += bytes(4096)
    dev = "cuda:0"
    t_comp = torch.frombuffer(bytearray(comp), dtype=torch.uint8).to(dev)
    t_desc = torch.tensor(desc, dtype=torch.int64).to(torch.int32).contiguous().to(dev)      # (values < 2^31)
    t_out = torch.full((o_out + 64,), 0x55, dtype=torch.uint8, device=dev)
    t_status = torch.full((len(pairs),), -1, dtype=torch.int32, device=dev)
    rc = L.coral_bgzf_inflate(t_comp.data_ptr(), t_desc.data_ptr(), len(pairs), t_out.data_ptr(), t_status.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.coral_bam_last_error()
    torch.cuda.synchronize()
    return t_out.cpu().numpy().tobytes(), t_status.cpu().tolist(), desc


def test_inflate_kernel_equals_zlib():
    pairs = _streams()
    out, status, desc = _inflate_on_gpu(pairs)
    assert status == [0] * len(pairs)
    prev_end = 0
    for (c, d), (_, _, o, n) in zip(pairs, desc):
        assert out[o:o + n] == d
        assert all(b == 0x55 for b in out[prev_end:o])                # nothing written between the outputs
        prev_end = o + n


def test_inflate_kernel_reports_corrupt_streams():
    rnd = random.Random(3)
    pairs = []
    for _ in range(200):
        data = bytes(rnd.choice(b"ACGT") for _ in range(5000))
        comp = bytearray(zlib.compress(data, 6)[2:-4])
        for _k in range(rnd.randrange(1, 4)):
            comp[rnd.randrange(len(comp))] ^= 1 << rnd.randrange(8)
        pairs.append((bytes(comp), data))
    out, status, desc = _inflate_on_gpu(pairs)
    for (c, d), (_, _, o, n), st in zip(pairs, desc, status):
        try:
            ok = zlib.decompress(c, -15) == d
        except zlib.error:
            ok = False
        if st == 0:
            assert out[o:o + n] == d or not ok      # status 0 means the right number of bytes came out of a well-formed stream
        if ok:
            assert st == 0 and out[o:o + n] == d    # (a flip inside unused padding bits leaves the stream valid)


def _same_records(a, b):
    """GPU result `a` against CPU result `b` (the CIGAR words of `a` are a device tensor)."""
    assert a.cigar.is_cuda
    assert_same(a, b)


@pytest.mark.parametrize("config,n_reads,batch", [("tiny", 3000, 0), ("tiny", 3000, 1 << 20), ("ultra", 300, 0), ("ultra", 300, 1 << 21),
                                                  ("cfg3", 1500, 0), ("cfg3", 1500, 16 << 20)])
def test_gpu_decode_equals_cpu_decode(config, n_reads, batch, tmp_path):
    """Whole file; `batch` small enough that there are many batches and records straddle them (ultra: records of several
    hundred KiB against 2 MiB batches)."""
    rec = synth.generate(synth.scaled_config(config, n_reads), "cpu")
    p = str(tmp_path / "x.bam")
    bam.write_bam_native(rec, p, seed=3, n_threads=4)
    cpu = bam.decode_bam(p, n_threads=4)
    gpu = bam.decode_bam_gpu(p, "cuda:0", batch_bytes=batch)
    _same_records(gpu, cpu)
    st = dict(bam.LAST_DECODE)
    assert st["where"] == "gpu" and st["uncompressed_bytes"] > 0
    if batch:
        assert st["batches"] > 1


def test_gpu_decode_odd_records(tmp_path):
    """CG-tag CIGAR (66001 ops), unmapped, no SEQ, hard clips, = / X / N ops, non-ACGT bases, SA shapes — python-written file."""
    rec = _odd()
    p = str(tmp_path / "odd.bam")
    bam.write_bam(rec, p)
    _same_records(bam.decode_bam_gpu(p), bam.decode_bam(p))
    M, S = 0, 4
    rec = synth.records_from_alignments([dict(tid=0, pos=10, cigar=[(S, 5), (M, 50)], name="x")])
    rec.sa_text = {0: "chr1,500,+,7S40M,60,1;chr2,9,-,47M,3,2;chr3,8,+,7H40M,1,0;chr4,7,-,1S2M3S4M,0,5;chrZ,6,+,3S4M2I1S,9,9;"}
    p = str(tmp_path / "s.bam")
    bam.write_bam(rec, p)
    g = bam.decode_bam_gpu(p)
    _same_records(g, bam.decode_bam(p))
    assert g.sa.tolist()[3] == [3, 7, 1, -2, 2, 0, 3, 0]


@pytest.mark.parametrize("config,n_reads", [("tiny", 3000), ("ultra", 400)])
def test_gpu_byte_ranges_partition_the_file(config, n_reads, tmp_path):
    """Every rank's GPU decode of its byte range equals the CPU decode of the same range, and the ranges put together are the
    whole file."""
    rec = synth.generate(synth.scaled_config(config, n_reads), "cpu")
    p = str(tmp_path / "x.bam")
    bam.write_bam_native(rec, p, seed=3, n_threads=4)
    whole = _concat([bam.decode_bam(p, n_threads=2)])
    for world in (2, 3, 5):
        parts = [bam.decode_bam_gpu(p, rank=r, world=world, batch_bytes=(1 << 21) if r % 2 else 0) for r in range(world)]
        for r, q in enumerate(parts):
            _same_records(q, bam.decode_bam(p, n_threads=2, rank=r, world=world))
        got = _concat(parts)
        for k, v in whole.items():
            assert (list(v) == list(got[k])) if k == "names" else np.array_equal(v, got[k]), (world, k)
    # batches so small that a range's first batch may not hold three records in a row: the search for the range's first record
    # then goes on over several batches (as the host pipeline's does over its chunks)
    for r in range(3):
        _same_records(bam.decode_bam_gpu(p, rank=r, world=3, batch_bytes=1 << 20), bam.decode_bam(p, n_threads=2, rank=r, world=3))


def test_gpu_decode_small_files(tmp_path):
    """One record; two records of one read; a file whose records are all tiny (thousands per BGZF block, many per segment of the
    record-start search) decoded whole and as seven byte ranges of a few blocks each."""
    M, S = 0, 4
    one = synth.records_from_alignments([dict(tid=2, pos=77, cigar=[(S, 3), (M, 40)], name="only")])
    p = str(tmp_path / "one.bam")
    bam.write_bam(one, p)
    _same_records(bam.decode_bam_gpu(p), bam.decode_bam(p))
    parts = [bam.decode_bam_gpu(p, rank=r, world=5) for r in range(5)]          # byte ranges without any block of their own
    assert [q.n for q in parts] == [bam.decode_bam(p, rank=r, world=5).n for r in range(5)] and sum(q.n for q in parts) == 1
    two = synth.records_from_alignments([dict(tid=0, pos=5, cigar=[(M, 30), (S, 10)], name="r", sa=[(1, 900, 1, 30, 10, 0, 0, 60, 1)]),
                                         dict(tid=1, pos=900, cigar=[(S, 30), (M, 10)], name="r", flag=2048, sa=[(0, 6, 0, 0, 30, 0, 10, 60, 0)])])
    p = str(tmp_path / "two.bam")
    bam.write_bam(two, p)
    g = bam.decode_bam_gpu(p)
    _same_records(g, bam.decode_bam(p))
    assert g.name_id.tolist() == [0, 0] and g.n_names == 1
    tiny = synth.records_from_alignments([dict(tid=k % 3, pos=100 + k, cigar=[(M, 20 + k % 7)], name="q%d" % (k // 2), has_seq=k % 5 != 0)
                                          for k in range(6000)])
    p = str(tmp_path / "tiny.bam")
    bam.write_bam(tiny, p)
    whole = bam.decode_bam(p)
    _same_records(bam.decode_bam_gpu(p), whole)
    parts = [bam.decode_bam_gpu(p, rank=r, world=7) for r in range(7)]
    assert sum(q.n for q in parts) == whole.n
    for r, q in enumerate(parts):
        _same_records(q, bam.decode_bam(p, rank=r, world=7))


def test_gpu_decode_rejects_corrupt_input(tmp_path):
    from coral_amd._lib import CoralHipError
    junk = tmp_path / "junk.bam"
    junk.write_bytes(b"this is not a BGZF file" * 10)
    with pytest.raises(CoralHipError):
        bam.decode_bam_gpu(str(junk))
    rec = synth.generate(synth.scaled_config("tiny", 300), "cpu")
    good = tmp_path / "good.bam"
    bam.write_bam_native(rec, str(good))
    data = bytearray(good.read_bytes())
    data[len(data) // 2: len(data) // 2 + 64] = b"\x00" * 64
    bad = tmp_path / "bad.bam"
    bad.write_bytes(bytes(data))
    with pytest.raises(CoralHipError):
        bam.decode_bam_gpu(str(bad))
    trunc = tmp_path / "trunc.bam"
    trunc.write_bytes(bytes(good.read_bytes()[:-5000]))
    with pytest.raises(CoralHipError):
        bam.decode_bam_gpu(str(trunc))
    # a byte flipped inside a STORED block: the block still inflates, only its CRC-32 tells (both pipelines check it, as htslib does)
    stored = tmp_path / "stored.bam"
    bam.write_bam_native(rec, str(stored), level=0)
    data = bytearray(stored.read_bytes())
    data[len(data) // 2] ^= 0x40
    flipped = tmp_path / "flipped.bam"
    flipped.write_bytes(bytes(data))
    with pytest.raises(CoralHipError, match="CRC32"):
        bam.decode_bam_gpu(str(flipped))
    with pytest.raises(CoralHipError):
        bam.decode_bam(str(flipped))
    _same_records(bam.decode_bam_gpu(str(stored)), bam.decode_bam(str(stored)))
    _same_records(bam.decode_bam_gpu(str(good)), bam.decode_bam(str(good)))      # and the decoder is fine afterwards


def test_graph_build_from_gpu_decoded_records(tmp_path):
    """The records of the GPU decode feed the graph build directly (CIGARs never visit host memory)."""
    from coral_amd import infer_breakpoint_graph as ibg
    from coral_amd.breakpoint_graph import graph_text
    from coral_amd.records import DeviceRecords
    cfg = synth.named_config("tiny")
    rec = synth.generate(cfg, "cpu")
    p = str(tmp_path / "t.bam")
    bam.write_bam_native(rec, p, seed=1)
    cn, seeds = str(tmp_path / "cn.bed"), str(tmp_path / "seeds.bed")
    synth.write_cn_bed(cfg, cn)
    synth.write_seed_bed(cfg, seeds)
    a = ibg.build_graph_from_records(DeviceRecords(bam.decode_bam_gpu(p), "cuda:0"), seeds, cn, str(tmp_path / "g"))
    b = ibg.build_graph_from_records(DeviceRecords(bam.decode_bam(p), "cuda:0"), seeds, cn, str(tmp_path / "c"))
    assert [graph_text(g) for g in a.lr_graph] == [graph_text(g) for g in b.lr_graph] and len(a.lr_graph) >= 1


def test_bam_file_to_graph_matches_oracle(tmp_path):
    """The whole chain on a BAM FILE of 30 000 config-3 reads: GPU decode -> graph build, against the CPU oracle working on the host
    pipeline's decode of the same file (order-normalised graph text; the strict comparison at 2 M reads is tools/validate_full_size.py
    ... bam, profiles/r02_full_size_parity_from_bam.md)."""
    from coral_amd import infer_breakpoint_graph as ibg
    from coral_amd.breakpoint_graph import graph_text
    from coral_amd.records import DeviceRecords
    from oracle import coral_oracle as O
    from oracle.hostrecords import HostRecords
    from tests.product_check import compare_graph_text
    cfg = synth.scaled_config("cfg3", 30000)
    rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000).to("cpu")
    p = str(tmp_path / "x.bam")
    bam.write_bam_native(rec, p, seed=1)
    cn, seeds = str(tmp_path / "cn.bed"), str(tmp_path / "seeds.bed")
    synth.write_cn_bed(cfg, cn)
    synth.write_seed_bed(cfg, seeds)
    b = ibg.build_graph_from_records(DeviceRecords(bam.load_bam(p, "cuda:0"), "cuda:0"), seeds, cn, str(tmp_path / "g"))
    assert bam.LAST_DECODE["where"] == "gpu"
    ob, ofiles = O.reconstruct_graph(HostRecords(bam.decode_bam(p)), seeds, cn)
    assert len(b.lr_graph) == len(ob.lr_graph) >= 1 and b.normal_cov == ob.normal_cov
    for g, og in zip(b.lr_graph, ob.lr_graph):
        got = sorted(graph_text(g).splitlines())
        exp = sorted(O.graph_text(og).splitlines())
        compare_graph_text("\n".join(got), "\n".join(exp))
