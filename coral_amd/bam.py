"""BAM input/output for the graph-build path.

``decode_bam`` is the product path: the native multi-threaded BGZF/BAM decoder of libcoral_hip.so
(csrc/coral_bam.cpp) turns the file into the SoA ``Records`` the kernels consume — the BAM is read once.
``write_bam`` is a small pure-Python BAM writer used by tests and tools to materialise synthetic records as a
real coordinate-sorted BAM (htslib / pysam are not available offline); it is not on the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import zlib
from typing import Optional

import numpy as np
import torch

from . import _lib
from .names import NameTable
from .synth import Records, hash_u32, S_SEQ, sa_entry_string


def default_threads() -> int:
    """Decoder threads: the CPUs this process may actually use (affinity mask and, in a container, the cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(64, n))


LAST_DECODE = {}      # statistics of the last decode_bam call (bench.py): seconds, compressed / uncompressed bytes, threads


def decode_bam(path: str, n_threads: Optional[int] = None, rank: int = 0, world: int = 1) -> Records:
    """Decode the BAM file (or, with ``world`` > 1, the ``rank``-th of ``world`` byte ranges of it: one process per GPU, every
    rank inflates and parses only its share; consecutive ranges neither drop nor repeat a record).  Read-name ids are local
    to the returned records."""
    L = _lib.lib()
    if n_threads is None:
        n_threads = default_threads()
    h = C.c_void_p()
    rc = L.coral_bam_decode_range(path.encode(), n_threads, rank, world, C.byref(h))
    if rc != 0:
        raise _lib.CoralHipError("coral_bam_decode(%s) failed (%d): %s" % (path, rc, L.coral_bam_last_error().decode()))
    st, secs = (C.c_int64 * 3)(), C.c_double(0.0)
    L.coral_bam_decode_stats(h, st, C.byref(secs))
    LAST_DECODE.clear()
    LAST_DECODE.update(seconds=float(secs.value), compressed_bytes=int(st[0]), uncompressed_bytes=int(st[1]), blocks=int(st[2]),
                       threads=int(n_threads))
    try:
        return _records_from_handle(L, h, None, 0)
    finally:
        L.coral_bam_decode_close(h)


def load_bam(path: str, device="cuda:0", n_threads: Optional[int] = None, rank: int = 0, world: int = 1) -> Records:
    """The product's way from a BAM file to records: inflate and parse on the GPU (``decode_bam_gpu``) when ``device`` is one;
    ``CORAL_BAM_DECODE=cpu`` selects the host pipeline (``decode_bam``: same result, CIGAR words in host memory)."""
    if torch.device(device).type == "cuda" and os.environ.get("CORAL_BAM_DECODE", "gpu") != "cpu":
        return decode_bam_gpu(path, device, n_threads=n_threads, rank=rank, world=world)
    return decode_bam(path, n_threads=n_threads, rank=rank, world=world)


def _records_from_handle(L, h, cigar, cigar_words: int) -> Records:
    """Fill a Records object from a decode handle (coral_bam_decode_sizes / _fill).  ``cigar`` None: the handle holds the op
    words (CPU pipeline); a tensor: the ops are already on the device (GPU pipeline) and the handle holds everything else."""
    sz = (C.c_int64 * 8)()
    _lib.check(L.coral_bam_decode_sizes(h, sz), "coral_bam_decode_sizes")
    n, ncig, nsa, nna, nnames, nbytes, nref, rbytes = (int(v) for v in sz)
    i32 = lambda k: np.empty(max(k, 0), dtype=np.int32)
    tid, pos, end, flag, mapq, qlen, has_seq, nm, name_id, n_cigar = (i32(n) for _ in range(10))
    cigar_off = np.empty(n + 1, dtype=np.int64)
    host_cigar = np.empty(ncig, dtype=np.uint32) if cigar is None else None
    sa_off = np.empty(n + 1, dtype=np.int64)
    sa = np.empty((nsa, 8), dtype=np.int32)
    sa_nm = i32(nsa)
    na_rec = np.empty(nna, dtype=np.int64)
    na_pos = i32(nna)
    names_blob = np.empty(max(nbytes, 1), dtype=np.uint8)
    name_off = np.zeros(nnames + 1, dtype=np.int64)
    ref_buf = C.create_string_buffer(max(rbytes, 1))
    ref_lens = i32(nref)
    ptr = lambda a: a.ctypes.data
    _lib.check(L.coral_bam_decode_fill(h, ptr(tid), ptr(pos), ptr(end), ptr(flag), ptr(mapq), ptr(qlen), ptr(has_seq),
                                       ptr(nm), ptr(name_id), ptr(n_cigar), ptr(cigar_off), ptr(host_cigar) if host_cigar is not None else None,
                                       ptr(sa_off), ptr(sa), ptr(sa_nm), ptr(na_rec), ptr(na_pos), ptr(names_blob),
                                       ptr(name_off), C.addressof(ref_buf), ptr(ref_lens)), "coral_bam_decode_fill")
    names = NameTable(names_blob[:nbytes], name_off)                  # the names stay bytes: a str is made when one is asked for
    refs = ref_buf.raw[:rbytes].split(b"\0")[:nref]
    t = torch.from_numpy
    if cigar is None:
        cigar = t(host_cigar.view(np.int32))
    else:
        assert int(cigar_off[-1]) == cigar_words, "device CIGAR words and host offsets disagree"
    return Records(n=n, tid=t(tid), pos=t(pos), end=t(end), flag=t(flag), mapq=t(mapq), qlen=t(qlen), has_seq=t(has_seq),
                   nm=t(nm), name_id=t(name_id), n_cigar=t(n_cigar), cigar_off=t(cigar_off),
                   cigar=cigar, sa_off=t(sa_off), sa=t(sa), sa_nm=t(sa_nm), nonacgt_rec=t(na_rec),
                   nonacgt_pos=t(na_pos), n_names=nnames, name_gid=None, names=names,
                   header_chroms=[x.decode() for x in refs], header_lens=[int(x) for x in ref_lens])


def decode_bam_gpu(path: str, device="cuda:0", n_threads: Optional[int] = None, rank: int = 0, world: int = 1,
                   batch_bytes: int = 0) -> Records:
    """The same result as ``decode_bam`` with the inflate and the record parsing on the GPU (csrc/coral_bamgpu.hip): the host
    only reads the file and uploads COMPRESSED bytes; the CIGAR words of the returned Records are a device tensor (they never
    exist in host memory), everything else is host-side as before.  ``batch_bytes``: inflated bytes per batch (0 = the default, 2.52 GiB)."""
    L = _lib.lib()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.CoralHipError("decode_bam_gpu needs a GPU device (the CPU pipeline is decode_bam)")
    torch.cuda.set_device(dev)
    if n_threads is None:
        n_threads = default_threads()
    h, ws_bytes = C.c_void_p(), C.c_int64(0)
    rc = L.coral_bamgpu_open(path.encode(), n_threads, rank, world, batch_bytes, C.byref(h), C.byref(ws_bytes))
    if rc != 0:
        raise _lib.CoralHipError("coral_bamgpu_open(%s) failed (%d): %s" % (path, rc, L.coral_bam_last_error().decode()))
    try:
        ws = torch.empty(int(ws_bytes.value) + 256, dtype=torch.uint8, device=dev)
        base = (ws.data_ptr() + 255) & ~255
        stream = torch.cuda.current_stream(dev).cuda_stream
        # `ws` may be a recycled block of torch's caching allocator: that is ordered only against the allocating stream, while the
        # decoder's feeder thread and its own streams start writing into the workspace at once.  Let whatever the current stream
        # still has queued (possibly on the block's previous owner) finish first — once per decode.
        torch.cuda.current_stream(dev).synchronize()
        fail = lambda what, rc: _lib.CoralHipError("%s(%s) failed (%d): %s" % (what, path, rc, L.coral_bam_last_error().decode()))
        rc = L.coral_bamgpu_start(h, base, int(ws_bytes.value))
        if rc != 0:
            raise fail("coral_bamgpu_start", rc)
        pieces, total = [], 0
        out = (C.c_int64 * 4)()
        while True:
            rc = L.coral_bamgpu_next(h, out, stream)
            if rc != 0:
                raise fail("coral_bamgpu_next", rc)
            if not out[2]:
                break
            words = int(out[1])
            piece = torch.empty(max(words, 1), dtype=torch.int32, device=dev)
            rc = L.coral_bamgpu_emit(h, piece.data_ptr(), None, stream)
            if rc != 0:
                raise fail("coral_bamgpu_emit", rc)
            if words:
                pieces.append(piece[:words])
                total += words
        cigar = torch.cat(pieces) if len(pieces) > 1 else (pieces[0] if pieces else torch.zeros(0, dtype=torch.int32, device=dev))
        del pieces
        dh = C.c_void_p()
        rc = L.coral_bamgpu_host(h, C.byref(dh))
        if rc != 0:
            raise fail("coral_bamgpu_host", rc)
        st, secs = (C.c_int64 * 3)(), C.c_double(0.0)
        L.coral_bam_decode_stats(dh, st, C.byref(secs))
        gst, gsecs = (C.c_int64 * 4)(), (C.c_double * 6)()
        L.coral_bamgpu_stats(h, gst, gsecs)
        LAST_DECODE.clear()
        LAST_DECODE.update(seconds=float(gsecs[0]), compressed_bytes=int(st[0]), uncompressed_bytes=int(st[1]), blocks=int(st[2]),
                           threads=int(n_threads), where="gpu", batches=int(gst[0]), rewalked_segments=int(gst[1]),
                           nonacgt_records_fetched=int(gst[2]), batch_bytes=int(gst[3]), host_seconds=float(gsecs[1]), read_seconds=float(gsecs[2]),
                           setup_seconds=float(gsecs[3]), waited_for_file_seconds=float(gsecs[4]), waited_for_gpu_seconds=float(gsecs[5]),
                           workspace_bytes=int(ws_bytes.value))
        return _records_from_handle(L, dh, cigar, total)
    finally:
        # (close drains the decoder's streams — a byte range may have batches of its overhang still being inflated — and only
        # then the workspace, which those kernels write, is released: `ws` lives until this function returns)
        L.coral_bamgpu_close(h)


def write_bam_native(rec: Records, path: str, seed: int = 0, level: int = 1, n_threads: Optional[int] = None) -> None:
    """Serialise ``rec`` as a coordinate-sorted BAM with the native multi-threaded writer (coral_bam_write): what benchmarks
    and the larger tests use — same content rules as ``write_bam`` (deterministic ACGT with N at the listed positions, QUAL
    absent, NM / SA / CG tags), different (hash-made) bases."""
    L = _lib.lib()
    g = lambda x, dt: np.ascontiguousarray(x.cpu().numpy(), dtype=dt)
    i32 = lambda k: g(getattr(rec, k), np.int32)
    cols = [i32(k) for k in ("tid", "pos", "flag", "mapq", "qlen", "has_seq", "nm", "name_id", "n_cigar")]
    cigar_off, cigar = g(rec.cigar_off, np.int64), g(rec.cigar, np.int32).view(np.uint32)
    sa_off, sa, sa_nm = g(rec.sa_off, np.int64), g(rec.sa, np.int32), g(rec.sa_nm, np.int32)
    na_rec, na_pos = g(rec.nonacgt_rec, np.int64), g(rec.nonacgt_pos, np.int32)
    names = [s.encode() for s in rec.materialise_names()]
    name_arr = (C.c_char_p * max(len(names), 1))(*names)
    refs = [c.encode() for c in rec.header_chroms]
    ref_arr = (C.c_char_p * max(len(refs), 1))(*refs)
    ref_lens = np.ascontiguousarray(rec.header_lens, dtype=np.int32)
    ptr = lambda a: a.ctypes.data
    rc = L.coral_bam_write(path.encode(), rec.n, *[ptr(c) for c in cols], ptr(cigar_off), ptr(cigar), ptr(sa_off), ptr(sa), ptr(sa_nm),
                           len(na_rec), ptr(na_rec), ptr(na_pos), name_arr, len(refs), ref_arr, ptr(ref_lens), seed, level,
                           n_threads or default_threads())
    if rc != 0:
        raise _lib.CoralHipError("coral_bam_write(%s) failed (%d): %s" % (path, rc, L.coral_bam_last_error().decode()))


# ----------------------------------------------------------------------------------------------
# pure-Python writer (tests / tools)
# ----------------------------------------------------------------------------------------------
_SEQ_CODE = {65: 1, 67: 2, 71: 4, 84: 8, 78: 15}     # A C G T N


def _reg2bin(beg: int, end: int) -> int:
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


_BGZF_EMPTY = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")      # 28 bytes; at the end: the EOF marker


def _bgzf_blocks(data: bytes, level: int = 1, block_size: int = 0xff00, empty_block_every: int = 0):
    """BGZF blocks of ``data`` (``block_size`` payload bytes each, at most 0xff00) + the EOF marker.  ``empty_block_every`` = k:
    an empty block (the 28 bytes of the EOF marker, legal anywhere in a BGZF stream) after every k-th data block."""
    assert 0 < block_size <= 0xff00
    for n, i in enumerate(range(0, len(data), block_size)):
        chunk = data[i:i + block_size]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = len(comp) + 25
        yield (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
               struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
        if empty_block_every and (n + 1) % empty_block_every == 0:
            yield _BGZF_EMPTY
    yield _BGZF_EMPTY      # BGZF EOF marker


_NM_PACK = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}


def write_bam(rec: Records, path: str, seed: int = 0, long_cigar_as_cg: bool = True, fast_seq: bool = False, *, aux=None,
              nm_type="i", with_qual: bool = False, block_size: int = 0xff00, empty_block_every: int = 0,
              header_comment: str = "") -> None:
    """Serialise ``rec`` as a coordinate-sorted BAM (SEQ = deterministic ACGT with N at the listed non-ACGT
    positions, QUAL absent, tags NM:i and SA:Z; CIGARs with more than 65535 ops go to the CG:B,I tag).

    Options for files shaped like what aligners and htslib really write (tests of the decoders):
      ``aux(i)``        -> (raw tag bytes in FRONT of NM, raw tag bytes BEHIND the last tag) of record i: any SAM aux tags;
      ``nm_type``       one of c C s S i I (htslib stores integers in the smallest type that fits), None (no NM tag), or a
                        callable i -> one of these;
      ``with_qual``     real QUAL bytes (a hash of the position, 0..60) instead of 0xff;
      ``block_size``    payload bytes per BGZF block (small: header, records and tags straddle blocks);
      ``empty_block_every``  an empty BGZF block after every k-th block;
      ``header_comment``     extra @CO text (a long header spans several BGZF blocks)."""
    g = lambda x: x.cpu().numpy()
    tid, pos, flag, mapq, qlen, has_seq, nm, name_id, n_cigar = (g(getattr(rec, k)) for k in
                                                                ("tid", "pos", "flag", "mapq", "qlen", "has_seq", "nm", "name_id", "n_cigar"))
    cigar_off, cigar = g(rec.cigar_off), g(rec.cigar).view(np.uint32)
    sa_off, sa, sa_nm = g(rec.sa_off), g(rec.sa), g(rec.sa_nm)
    names = rec.materialise_names()
    na = {}
    for r, p in zip(g(rec.nonacgt_rec), g(rec.nonacgt_pos)):
        na.setdefault(int(r), []).append(int(p))
    out = bytearray()
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % (c, l) for c, l in zip(rec.header_chroms, rec.header_lens))
    if header_comment:
        text += "".join("@CO\t%s\n" % line for line in header_comment.split("\n"))
    out += b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(rec.header_chroms))
    for c, l in zip(rec.header_chroms, rec.header_lens):
        out += struct.pack("<i", len(c) + 1) + c.encode() + b"\0" + struct.pack("<i", l)
    rng = np.random.default_rng(seed)
    lut = np.zeros(256, dtype=np.uint8)
    for k, v in _SEQ_CODE.items():
        lut[k] = v
    ref_adv = np.array([1, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0])
    qry_adv = np.array([1, 1, 0, 0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0])
    for i in range(rec.n):
        ops = cigar[cigar_off[i]: cigar_off[i] + n_cigar[i]]
        l_seq = int(qlen[i]) if has_seq[i] else 0
        seq_bytes = b""
        if l_seq:
            if fast_seq:            # decode benchmarks: any ACGT content will do
                s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, l_seq, dtype=np.uint8)]
            else:
                k = torch.arange(l_seq, dtype=torch.int64) + i * (1 << 22)
                s = np.frombuffer(b"ACGT", dtype=np.uint8)[(hash_u32(seed, S_SEQ, k) & 3).numpy()].copy()
            if i in na:     # reference position -> query offset through the CIGAR
                op, ln = ops & 15, (ops >> 4).astype(np.int64)
                r0 = int(pos[i]) + np.cumsum(ref_adv[op] * ln) - ref_adv[op] * ln
                q0 = np.cumsum(qry_adv[op] * ln) - qry_adv[op] * ln
                for p in na[i]:
                    j = np.nonzero((ref_adv[op] * qry_adv[op] == 1) & (r0 <= p) & (p < r0 + ln))[0][0]
                    s[q0[j] + (p - r0[j])] = 78
            code = lut[s]
            if l_seq & 1:
                code = np.append(code, 0)
            seq_bytes = ((code[0::2] << 4) | code[1::2]).astype(np.uint8).tobytes()
        front, behind = aux(i) if aux is not None else (b"", b"")
        ty = nm_type(i) if callable(nm_type) else nm_type
        tags = front + (b"NM" + ty.encode() + struct.pack(_NM_PACK[ty], int(nm[i])) if ty is not None else b"")
        if i in getattr(rec, "sa_text", {}):            # tests: verbatim SA text (odd CIGAR shapes)
            tags += b"SAZ" + rec.sa_text[i].encode() + b"\0"
        elif sa_off[i + 1] > sa_off[i]:
            ents = [sa_entry_string(sa[j], int(sa_nm[j]), rec.header_chroms) for j in range(sa_off[i], sa_off[i + 1])]
            tags += b"SAZ" + (";".join(ents) + ";").encode() + b"\0"
        rlen = int((ref_adv[ops & 15] * (ops >> 4)).sum()) if not (flag[i] & 4) else 0
        cig_field = ops
        if long_cigar_as_cg and len(ops) > 65535:
            tags += b"CGBI" + struct.pack("<I", len(ops)) + ops.astype("<u4").tobytes()
            cig_field = np.array([(l_seq << 4) | 4, (rlen << 4) | 3], dtype=np.uint32)
        name = names[name_id[i]].encode() + b"\0"
        body = struct.pack("<iiBBHHHiiii", int(tid[i]), int(pos[i]), len(name), int(mapq[i]),
                           _reg2bin(int(pos[i]), int(pos[i]) + max(1, rlen)), len(cig_field), int(flag[i]), l_seq, -1, -1, 0)
        if with_qual and l_seq:
            qual_bytes = ((hash_u32(seed, S_SEQ, torch.arange(l_seq, dtype=torch.int64) + (i + 7) * (1 << 22)) % 61).numpy()
                          .astype(np.uint8).tobytes())
        else:
            qual_bytes = b"\xff" * l_seq
        body += name + cig_field.astype("<u4").tobytes() + seq_bytes + qual_bytes + tags + behind
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as fp:
        for blk in _bgzf_blocks(bytes(out), block_size=block_size, empty_block_every=empty_block_every):
            fp.write(blk)
