"""Alignment records resident in HBM (structure of arrays) + the small host mirrors the host logic needs.

PyTorch is used for device memory and streams only (plumbing); the kernels are in csrc/ behind the C ABI.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .names import NameTable


class HostMirrors:
    """Per-record fields of the WHOLE file the host logic works on (32 bytes per record, no CIGARs) + tokenised SA rows,
    non-ACGT positions and read names (``coral_amd.names.NameTable``: one byte blob + offsets, never 2 M ``str``).  Built from
    one Records object (single process) or from the gathered per-rank pieces (``from_pieces``: every rank decoded its own
    byte range of the BAM; read-name ids are unified here, natively)."""

    # what a rank contributes to the gather, in transport order: fixed-width columns, then ONE name blob + its offsets
    PIECE_SPEC = (("tid", np.int32), ("pos", np.int32), ("end", np.int32), ("flag", np.int32), ("mapq", np.int32), ("qlen", np.int32),
                  ("has_seq", np.uint8), ("nm", np.int32), ("name_id", np.int32), ("n_cigar", np.int32), ("sa_count", np.int64),
                  ("sa", np.int32), ("sa_nm", np.int32), ("nonacgt_rec", np.int64), ("nonacgt_pos", np.int32),
                  ("name_blob", np.uint8), ("name_off", np.int64))

    def __init__(self, rec=None):
        if rec is not None:
            d = HostMirrors.piece_of(rec, with_names=False)
            self._from_arrays(d, names=rec.names if isinstance(rec.names, NameTable) else None, lazy_names=rec)

    @staticmethod
    def piece_of(rec, with_names=True) -> dict:
        """The host-side fields of one Records object as numpy arrays (what a rank contributes to the gather)."""
        h = lambda t: t.detach().cpu().numpy()
        sa_off = h(rec.sa_off).astype(np.int64)
        d = dict(tid=h(rec.tid).astype(np.int32), pos=h(rec.pos).astype(np.int32), end=h(rec.end).astype(np.int32),
                 flag=h(rec.flag).astype(np.int32), mapq=h(rec.mapq).astype(np.int32), qlen=h(rec.qlen).astype(np.int32),
                 has_seq=h(rec.has_seq).astype(np.uint8), nm=h(rec.nm).astype(np.int32), name_id=h(rec.name_id).astype(np.int32),
                 n_cigar=h(rec.n_cigar).astype(np.int32), sa_count=np.diff(sa_off).astype(np.int64),
                 sa=h(rec.sa).astype(np.int32).reshape(-1, 8), sa_nm=h(rec.sa_nm).astype(np.int32),
                 nonacgt_rec=h(rec.nonacgt_rec).astype(np.int64), nonacgt_pos=h(rec.nonacgt_pos).astype(np.int32),
                 n_names=int(rec.n_names))
        if with_names:
            t = rec.name_table()
            d["name_blob"], d["name_off"] = t.blob[int(t.off[0]):int(t.off[-1])], t.off - t.off[0]
        return d

    @classmethod
    def pack(cls, piece: dict) -> np.ndarray:
        """One contiguous uint8 buffer: int64 element counts (one per PIECE_SPEC entry), then the raw columns, each starting at
        a multiple of 8 bytes.  No pickling, no text: the receiver maps the columns in place (``unpack``)."""
        cols = [np.ascontiguousarray(piece[k], dtype=dt).reshape(-1) for k, dt in cls.PIECE_SPEC]
        head = np.array([len(c) for c in cols], dtype=np.int64)
        sizes = [(c.nbytes + 7) & ~7 for c in cols]
        buf = np.zeros(head.nbytes + sum(sizes), dtype=np.uint8)
        buf[:head.nbytes] = head.view(np.uint8)
        at = head.nbytes
        for c, sz in zip(cols, sizes):
            buf[at:at + c.nbytes] = c.view(np.uint8)
            at += sz
        return buf

    @classmethod
    def unpack(cls, buf: np.ndarray) -> dict:
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        nh = 8 * len(cls.PIECE_SPEC)
        head = buf[:nh].view(np.int64)
        d, at = {}, nh
        for (k, dt), n in zip(cls.PIECE_SPEC, head.tolist()):
            nb = n * np.dtype(dt).itemsize
            d[k] = buf[at:at + nb].view(dt)
            at += (nb + 7) & ~7
        d["sa"] = d["sa"].reshape(-1, 8)
        d["n_names"] = len(d["name_off"]) - 1
        return d

    @classmethod
    def from_pieces(cls, pieces: List[dict], n_threads: int = 8) -> "HostMirrors":
        """Pieces of consecutive record ranges (rank order = file order), each with range-local name ids and its own name
        table (``name_blob`` / ``name_off``): one mirror of the whole file.  Global name ids are given in order of first
        appearance over the file, exactly what a single-process decode gives (coral_names_unify: a native hash join over
        the name bytes — no ``str``, no dict)."""
        import ctypes as C
        L = _lib.lib()
        P = len(pieces)
        n_names = np.array([len(p["name_off"]) - 1 for p in pieces], dtype=np.int64)
        blobs = [np.ascontiguousarray(p["name_blob"], dtype=np.uint8) for p in pieces]
        offs = [np.ascontiguousarray(p["name_off"], dtype=np.int64) for p in pieces]
        luts = [np.zeros(max(int(n), 1), dtype=np.int32) for n in n_names]
        out_blob = np.empty(max(sum(len(b) for b in blobs), 1), dtype=np.uint8)
        out_off = np.zeros(int(n_names.sum()) + 1, dtype=np.int64)
        arr = lambda xs: (C.c_void_p * max(P, 1))(*[x.ctypes.data for x in xs])
        n_global = C.c_int64(0)
        _lib.check(L.coral_names_unify(P, n_names.ctypes.data, arr(blobs), arr(offs), arr(luts), out_blob.ctypes.data,
                                       out_off.ctypes.data, C.byref(n_global), n_threads), "coral_names_unify")
        ng = int(n_global.value)
        names = NameTable(out_blob[:int(out_off[ng])], out_off[:ng + 1])
        cat = lambda k: np.concatenate([p[k] for p in pieces])
        base = np.cumsum([0] + [len(p["tid"]) for p in pieces])
        d = {k: cat(k) for k, _ in cls.PIECE_SPEC if k not in ("name_id", "nonacgt_rec", "name_blob", "name_off")}
        d["name_id"] = np.concatenate([lut[p["name_id"]] if len(p["name_id"]) else np.zeros(0, dtype=np.int32)
                                       for p, lut in zip(pieces, luts)]).astype(np.int32, copy=False)
        d["nonacgt_rec"] = np.concatenate([p["nonacgt_rec"] + base[i] for i, p in enumerate(pieces)])
        d["n_names"] = ng
        self = cls()
        self._from_arrays(d, names=names, lazy_names=None)
        return self

    def _from_arrays(self, d: dict, names, lazy_names):
        self.h_tid, self.h_pos, self.h_end = d["tid"], d["pos"], d["end"]
        self.h_flag, self.h_mapq = d["flag"], d["mapq"]
        self.h_has_seq = d["has_seq"].astype(bool)
        self.h_qlen = np.where(self.h_has_seq, d["qlen"], 0).astype(np.int32)      # pysam query_length
        self.h_nm, self.h_name_id, self.h_n_cigar = d["nm"], d["name_id"], d["n_cigar"]
        self.h_sa_off = np.concatenate([[0], np.cumsum(d["sa_count"])]).astype(np.int64)
        self.h_sa, self.h_sa_nm = d["sa"], d["sa_nm"]
        self.h_nonacgt_rec, self.h_nonacgt_pos = d["nonacgt_rec"], d["nonacgt_pos"]
        self.n_names = int(d["n_names"])
        self.n_total = len(self.h_tid)
        self._names = names
        self._lazy_names = lazy_names


class DeviceRecords:
    """Records of one BAM (or one shard of it) in file order.

    Device tensors (HBM):  tid, pos, end, flagmq (flag | mapq<<16 | has_seq<<24), n_cigar : int32[n];
                           cigar_off : int64[n+1] (multiples of 4); cigar : int32[total] (BAM-packed, op-15 padded)
    Host mirrors (numpy):  the same per-record fields + qlen, nm, name_id, and the tokenised SA rows — of the WHOLE file on the
                           rank that runs the host logic (rank 0), absent on the other ranks.
    """

    def __init__(self, rec, device="cuda:0", rank=0, world=1, group=None, *, local_only=False, lo=None, n_total=None, host=None):
        """``rec``: the records this object takes its device arrays from.
        Default (``local_only`` False): ``rec`` is the whole file; with ``world`` > 1 this process keeps records [lo, hi) of it
        in HBM (a contiguous range balanced by CIGAR-op count) and, on rank 0 only, the host mirrors of everything.
        ``local_only`` True: ``rec`` IS this rank's shard (it decoded only its byte range of the BAM); ``lo`` / ``n_total`` place it
        in the file and ``host`` (rank 0: HostMirrors of the whole file; other ranks: None) comes from the gather."""
        self.device = torch.device(device)
        dev = self.device
        if dev.type == "cuda":
            torch.cuda.set_device(dev)          # libcoral_hip launches on this device's streams: it must be the thread's current GPU
        self.rank, self.world, self.group = int(rank), int(world), group
        self.header_chroms = list(rec.header_chroms)
        self.header_lens = list(rec.header_lens)
        off_all = rec.cigar_off
        if local_only:
            self.n_total = int(n_total)
            self.lo, self.hi = int(lo), int(lo) + int(rec.n)
            a, b = 0, int(rec.n)
        else:
            self.n_total = int(rec.n)
            if world > 1 and self.n_total > 0:
                total = int(off_all[-1])
                cuts = torch.searchsorted(off_all[:-1].contiguous(), torch.tensor([total * r // world for r in range(world + 1)],
                                                                                   dtype=off_all.dtype, device=off_all.device))
                cuts[0], cuts[-1] = 0, self.n_total
                self.lo, self.hi = int(cuts[rank]), int(cuts[rank + 1])
            else:
                self.lo, self.hi = 0, self.n_total
            a, b = self.lo, self.hi
        self.n = b - a
        i32 = lambda t: t[a:b].to(device=dev, dtype=torch.int32).contiguous()
        self.tid = i32(rec.tid)
        self.pos = i32(rec.pos)
        self.end = i32(rec.end)
        # the longest reference span of a local alignment: bounds the window coral_point_cover looks at per query point
        self.max_span = int((self.end - self.pos).max()) if self.n else 0
        flagmq = (rec.flag.to(torch.int64) & 0xFFFF) | ((rec.mapq.to(torch.int64) & 0xFF) << 16) | \
                 ((rec.has_seq.to(torch.int64) & 1) << 24)
        self.flagmq = i32(flagmq)
        self.n_cigar = i32(rec.n_cigar)
        have = int(rec.n) > 0
        c0 = int(off_all[a]) if have else 0
        c1 = int(off_all[b]) if have else 0
        self.cigar_off = (off_all[a:b + 1] - c0).to(device=dev, dtype=torch.int64).contiguous() if have else \
            torch.zeros(1, dtype=torch.int64, device=dev)
        self.cigar = rec.cigar[c0:c1].to(device=dev, dtype=torch.int32).contiguous()
        if self.cigar.numel() == 0:
            self.cigar = torch.zeros(4, dtype=torch.int32, device=dev)
        # layout contract of include/coral_hip.h — checked on the host before any kernel can touch the arrays
        assert self.cigar.data_ptr() % 16 == 0
        assert self.cigar_off.numel() == self.n + 1
        if self.n:
            assert int((self.cigar_off & 3).max()) == 0, "cigar_off must be multiples of 4 ops"
            assert int(self.cigar_off[-1]) <= self.cigar.numel(), "cigar array shorter than its offsets"
            span = self.cigar_off[1:] - self.cigar_off[:-1]
            assert bool((span >= ((self.n_cigar.to(torch.int64) + 3) // 4) * 4).all()), "record ops exceed their slot"
        self.total_ops = int(rec.n_cigar[a:b].to(torch.int64).sum()) if have else 0
        self.n_sa_local = int(rec.sa_off[b] - rec.sa_off[a]) if have else 0
        from .global_names import chr_idx
        self.chr_rank = np.array([chr_idx.get(c, -1) for c in self.header_chroms], dtype=np.int32)     # gn:13-18; -1 = other contig
        # ---- host mirrors of the whole file: only where the host logic runs
        if local_only:
            mirrors = host
        else:
            mirrors = HostMirrors(rec) if (self.rank == 0) else None
        self.has_host = mirrors is not None
        if mirrors is not None:
            for k, v in vars(mirrors).items():
                if k.startswith("h_"):
                    setattr(self, k, v)
            self.n_names = mirrors.n_names
            self._names: Optional[NameTable] = mirrors._names
            self._lazy_names = mirrors._lazy_names
            assert mirrors.n_total == self.n_total, "host mirrors do not describe the whole file"
            self.total_ops_all = int(self.h_n_cigar.astype(np.int64).sum())
            self.n_sa = int(self.h_sa.shape[0])
            # per-contig record ranges of the coordinate-sorted file; unplaced reads (refID -1) sit at the END of a sorted BAM,
            # so the binary search runs over the mapped prefix only
            n_mapped = int(np.count_nonzero(self.h_tid >= 0))
            mapped = self.h_tid[:n_mapped]
            if n_mapped and (bool((mapped < 0).any()) or bool((np.diff(mapped) < 0).any())):
                raise ValueError("records are not sorted by contig (mapped records first, in header order): sort the BAM by coordinate")
            t = np.arange(len(self.header_chroms))
            self.tid_lo = np.searchsorted(mapped, t, side="left")
            self.tid_hi = np.searchsorted(mapped, t, side="right")
            if dev.type == "cuda":
                self.sa_device_arrays()          # part of loading the records, not of the first build on them
        else:
            # a rank that only serves kernels: no per-record host data at all
            self.h_nonacgt_rec = np.zeros(0, dtype=np.int64)
            self.h_nonacgt_pos = np.zeros(0, dtype=np.int32)
            self.h_tid = np.zeros(0, dtype=np.int32)
            self.n_names, self._names, self._lazy_names = 0, NameTable.from_list([]), None
            self.total_ops_all, self.n_sa = self.total_ops, 0

    @property
    def names(self) -> NameTable:
        """Read names by name id: a sequence of ``str`` that keeps the bytes and makes strings on demand."""
        if self._names is None:
            self._names = self._lazy_names.name_table()
            self._lazy_names = None
        return self._names

    def algorithmic_bytes(self) -> int:
        """SURVEY.md §8(d):  Σ_rec (32 + 4·n_cigar) + 32·N_SA  over the records this process holds."""
        return 32 * self.n + 4 * self.total_ops + 32 * self.n_sa_local

    def c_struct(self) -> _lib.coral_records_t:
        return _lib.coral_records_t(self.n, self.tid.data_ptr(), self.pos.data_ptr(), self.end.data_ptr(),
                                    self.flagmq.data_ptr(), self.n_cigar.data_ptr(), self.cigar_off.data_ptr(),
                                    self.cigar.data_ptr())

    def sa_device_arrays(self):
        """Device copies of what coral_sa_table reads (whole file; uploaded once, ~40 B per SA row + 16 B per record)."""
        assert self.has_host, "the SA table is built where the host logic runs (rank 0)"
        if getattr(self, "_sa_dev", None) is None:
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            flagmq = (self.h_flag.astype(np.int64) & 0xFFFF) | ((self.h_mapq.astype(np.int64) & 0xFF) << 16)
            sa_rec = np.repeat(np.arange(self.n_total, dtype=np.int32), np.diff(self.h_sa_off))
            sa = self.h_sa if len(self.h_sa) else np.zeros((1, 8), dtype=np.int32)
            self._sa_dev = dict(tid=up(self.h_tid), flagmq=up(flagmq.astype(np.int32)), qlen=up(self.h_qlen),
                                name=up(self.h_name_id), sa=up(sa.astype(np.int32)),
                                sa_nm=up(self.h_sa_nm if len(self.h_sa_nm) else np.zeros(1, dtype=np.int32)),
                                sa_rec=up(sa_rec if len(sa_rec) else np.zeros(1, dtype=np.int32)))
        return self._sa_dev

    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def region(self, tid: int, start: int, stop: int) -> np.ndarray:  # whole file
        """Host-side record-level region query (htslib rule pos < stop and end > start), file order."""
        lo, hi = self.tid_lo[tid], self.tid_hi[tid]
        m = (self.h_pos[lo:hi] < stop) & (self.h_end[lo:hi] > start)
        return lo + np.nonzero(m)[0]
