"""Alignment records resident in HBM (structure of arrays) + the small host mirrors the host logic needs.

PyTorch is used for device memory and streams only (plumbing); the kernels are in csrc/ behind the C ABI.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import _lib


class DeviceRecords:
    """Records of one BAM (or one shard of it) in file order.

    Device tensors (HBM):  tid, pos, end, flagmq (flag | mapq<<16 | has_seq<<24), n_cigar : int32[n];
                           cigar_off : int64[n+1] (multiples of 4); cigar : int32[total] (BAM-packed, op-15 padded)
    Host mirrors (numpy):  the same per-record fields + qlen, nm, name_id, and the tokenised SA rows.
    """

    def __init__(self, rec, device="cuda:0", rank=0, world=1, group=None):
        """``rank``/``world``: this process keeps records [lo, hi) of the file in HBM (a contiguous range balanced
        by CIGAR-op count); the small host mirrors always describe the whole file."""
        self.device = torch.device(device)
        dev = self.device
        if dev.type == "cuda":
            torch.cuda.set_device(dev)          # libcoral_hip launches on this device's streams: it must be the thread's current GPU
        self.rank, self.world, self.group = int(rank), int(world), group
        self.n_total = int(rec.n)
        self.header_chroms = list(rec.header_chroms)
        self.header_lens = list(rec.header_lens)
        off_all = rec.cigar_off
        if world > 1 and self.n_total > 0:
            total = int(off_all[-1])
            cuts = torch.searchsorted(off_all[:-1].contiguous(), torch.tensor([total * r // world for r in range(world + 1)],
                                                                               dtype=off_all.dtype, device=off_all.device))
            cuts[0], cuts[-1] = 0, self.n_total
            self.lo, self.hi = int(cuts[rank]), int(cuts[rank + 1])
        else:
            self.lo, self.hi = 0, self.n_total
        lo, hi = self.lo, self.hi
        self.n = hi - lo
        i32 = lambda t: t[lo:hi].to(device=dev, dtype=torch.int32).contiguous()
        self.tid = i32(rec.tid)
        self.pos = i32(rec.pos)
        self.end = i32(rec.end)
        flagmq = (rec.flag.to(torch.int64) & 0xFFFF) | ((rec.mapq.to(torch.int64) & 0xFF) << 16) | \
                 ((rec.has_seq.to(torch.int64) & 1) << 24)
        self.flagmq = i32(flagmq)
        self.n_cigar = i32(rec.n_cigar)
        c0 = int(off_all[lo]) if self.n_total else 0
        c1 = int(off_all[hi]) if self.n_total else 0
        self.cigar_off = (off_all[lo:hi + 1] - c0).to(device=dev, dtype=torch.int64).contiguous()
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
        # host mirrors
        h = lambda t: t.detach().cpu().numpy()
        self.h_tid = h(rec.tid).astype(np.int32)
        self.h_pos = h(rec.pos).astype(np.int32)
        self.h_end = h(rec.end).astype(np.int32)
        self.h_flag = h(rec.flag).astype(np.int32)
        self.h_mapq = h(rec.mapq).astype(np.int32)
        self.h_has_seq = h(rec.has_seq).astype(bool)
        self.h_qlen = np.where(self.h_has_seq, h(rec.qlen), 0).astype(np.int32)    # pysam query_length
        self.h_nm = h(rec.nm).astype(np.int32)
        self.h_name_id = h(rec.name_id).astype(np.int32)
        self.h_n_cigar = h(rec.n_cigar).astype(np.int32)
        self.h_sa_off = h(rec.sa_off).astype(np.int64)
        self.h_sa = h(rec.sa).astype(np.int32)
        self.h_sa_nm = h(rec.sa_nm).astype(np.int32)
        self.h_nonacgt_rec = h(rec.nonacgt_rec).astype(np.int64)
        self.h_nonacgt_pos = h(rec.nonacgt_pos).astype(np.int32)
        self.n_names = int(rec.n_names)
        from .global_names import chr_idx
        self.chr_rank = np.array([chr_idx.get(c, -1) for c in self.header_chroms], dtype=np.int32)     # gn:13-18; -1 = other contig
        self._rec = rec
        self._names: Optional[List[str]] = rec.names
        self.total_ops_all = int(self.h_n_cigar.astype(np.int64).sum())
        self.total_ops = int(self.h_n_cigar[lo:hi].astype(np.int64).sum())
        self.n_sa_local = int(self.h_sa_off[hi] - self.h_sa_off[lo]) if self.n_total else 0
        self.n_sa = int(self.h_sa.shape[0])
        # per-contig record ranges of the coordinate-sorted file; unplaced reads (refID -1) sit at the END of a sorted BAM, so
        # the binary search runs over the mapped prefix only
        n_mapped = int(np.count_nonzero(self.h_tid >= 0))
        mapped = self.h_tid[:n_mapped]
        if n_mapped and (bool((mapped < 0).any()) or bool((np.diff(mapped) < 0).any())):
            raise ValueError("records are not sorted by contig (mapped records first, in header order): sort the BAM by coordinate")
        t = np.arange(len(self.header_chroms))
        self.tid_lo = np.searchsorted(mapped, t, side="left")
        self.tid_hi = np.searchsorted(mapped, t, side="right")

    @property
    def names(self) -> List[str]:
        if self._names is None:
            self._names = self._rec.materialise_names()
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
