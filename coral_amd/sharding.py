"""Multi-GPU sharding of the graph build: one process per GPU, torch.distributed over RCCL.

Input side: every rank decodes ITS byte range of the BAM (coral_bam_decode_range) and uploads only that shard; the per-record
host fields the order-sensitive host logic needs (32 bytes per record, no CIGARs) and the read names are gathered ONCE to
rank 0, which unifies the range-local read-name ids (``load_bam_sharded``).  Synthetic benchmarks instead slice records that
already sit on the GPU (``shard_records``).

Per build: every per-record kernel runs on the local shard only.  The path has two real exchange steps and only those use a
collective:
  * all-gather-v of the compacted candidate rows (large-gap rows, point-cover pairs), tagged with GLOBAL record
    ordinals and re-sorted into the single-GPU (= reference) iteration order;
  * all-reduce(sum, int64) of the per-segment (n_reads, n_bases) vectors — integer sums, identical for any world.
Rank 0 runs the small order-sensitive host logic and tells the other ranks which kernel to run next with a fixed-size int64
command tensor (+ one int64 payload tensor whose shape the command carries); ranks > 0 sit in ``serve``.  Works with backend
"nccl" (RCCL over xGMI) and, for the CPU tests, "gloo".
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

CMD_DONE, CMD_SCAN, CMD_COVERAGE, CMD_POINTS = 0, 1, 2, 3


def shard_records(rec, rank: int, world: int, device, group=None):
    from .records import DeviceRecords
    return DeviceRecords(rec, device, rank=rank, world=world, group=group)


def _comm_device(dr):
    """Where collective buffers live: the GPU with RCCL, host memory with gloo."""
    return torch.device("cpu") if dist.get_backend(dr.group) == "gloo" else dr.device


def command(dr, op: int, params=(), payload=None):
    """Rank 0 -> all: the next kernel to run (called from the public wrappers in coral_amd.kernels).  One int64[8] header
    (op, three parameters, payload rows, payload columns) and, if the command has one, one int64 payload tensor."""
    assert dr.rank == 0
    dev = _comm_device(dr)
    rows, cols = (payload.shape if payload is not None else (0, 0))
    head = torch.tensor([op] + [int(p) for p in params] + [0] * (3 - len(params)) + [rows, cols, 0, 0], dtype=torch.int64, device=dev)
    dist.broadcast(head, src=0, group=dr.group)
    if rows:
        dist.broadcast(torch.as_tensor(np.ascontiguousarray(payload, dtype=np.int64)).to(dev), src=0, group=dr.group)


def _receive(dr):
    dev = _comm_device(dr)
    head = torch.zeros(8, dtype=torch.int64, device=dev)
    dist.broadcast(head, src=0, group=dr.group)
    h = head.tolist()
    payload = None
    if h[4]:
        buf = torch.zeros((h[4], h[5]), dtype=torch.int64, device=dev)
        dist.broadcast(buf, src=0, group=dr.group)
        payload = buf.cpu().numpy()
    return h[0], h[1:4], payload


def _via_host(dr, t: torch.Tensor) -> bool:
    """gloo cannot all-gather device tensors: stage through host memory (CPU tests / single-GPU rehearsal only)."""
    return t.is_cuda and dist.get_backend(dr.group) == "gloo"


def allgather_rows(dr, rows: torch.Tensor) -> torch.Tensor:
    """All-gather-v of an int64 [k, c] row tensor (k differs per rank), concatenated in rank order: the row counts are
    exchanged first, then every rank contributes its rows padded to the largest count (the rows are a few KB)."""
    if _via_host(dr, rows):
        return allgather_rows(dr, rows.cpu()).to(rows.device)
    dev = rows.device
    k = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
    ks = [torch.zeros_like(k) for _ in range(dr.world)]
    dist.all_gather(ks, k, group=dr.group)
    ks = [int(x.item()) for x in ks]
    kmax = max(ks)
    c = rows.shape[1]
    if kmax == 0:
        return rows
    pad = torch.zeros((kmax, c), dtype=torch.int64, device=dev)
    pad[:rows.shape[0]] = rows
    bufs = [torch.empty_like(pad) for _ in range(dr.world)]
    dist.all_gather(bufs, pad, group=dr.group)
    return torch.cat([b[:n] for b, n in zip(bufs, ks)], dim=0)


def allreduce_sum(dr, t: torch.Tensor) -> torch.Tensor:
    if _via_host(dr, t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=dr.group)
        return h.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=dr.group)
    return t


def serve(dr):
    """Worker loop of ranks > 0: execute the kernels rank 0 asks for until it says done."""
    from . import kernels
    scan = None
    while True:
        op, params, payload = _receive(dr)
        if op == CMD_DONE:
            return
        if op == CMD_SCAN:
            scan = kernels.cigar_scan(dr, params[0], params[1], params[2], _worker=True)
        elif op == CMD_COVERAGE:
            kernels.segment_coverage(dr, scan, payload, _worker=True)
        elif op == CMD_POINTS:
            kernels.point_cover(dr, payload, params[0], _worker=True)
        else:
            raise RuntimeError("unknown command %r" % (op,))


def build_graph_sharded(dr, seedfile, cn_seg, output_prefix=None, min_bp_support=1.0, output_bp=False, gc_policy="pause"):
    """Graph build over sharded records: rank 0 returns the builder object, other ranks return None."""
    from . import infer_breakpoint_graph as ibg
    if dr.world == 1:
        return ibg.build_graph_from_records(dr, seedfile, cn_seg, output_prefix, min_bp_support, output_bp, gc_policy=gc_policy)
    if dr.rank == 0:
        try:
            b = ibg.build_graph_from_records(dr, seedfile, cn_seg, output_prefix, min_bp_support, output_bp, gc_policy=gc_policy)
        finally:
            command(dr, CMD_DONE)
        return b
    serve(dr)
    return None


# ----------------------------------------------------------------------------------------------
# input side: every rank decodes and uploads its own byte range of the BAM
# ----------------------------------------------------------------------------------------------
def _gather_to_rank0(dr, buf: np.ndarray):
    """Variable-length uint8 buffers of all ranks on rank 0 (None elsewhere): the sizes first (one small all-gather), then ONE
    gather of the buffers padded to the largest — only rank 0 receives (the other ranks have no use for the pieces)."""
    dev = _comm_device(dr)
    n = torch.tensor([len(buf)], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(dr.world)]
    dist.all_gather(ns, n, group=dr.group)
    ns = [int(x.item()) for x in ns]
    nmax = max(max(ns), 8)
    send = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    if len(buf):
        send[:len(buf)] = torch.from_numpy(buf).to(dev)
    outs = [torch.empty_like(send) for _ in range(dr.world)] if dr.rank == 0 else None
    dist.gather(send, outs, dst=0, group=dr.group)
    if dr.rank != 0:
        return None
    return [o[:k].cpu().numpy() for o, k in zip(outs, ns)]


LAST_LOAD = {}      # timings of the last load_bam_sharded call on this rank (bench.py): decode, gather, merge seconds


def load_bam_sharded(path: str, rank: int, world: int, device, group=None, n_threads=None):
    """Per-rank input: decode the rank's byte range of the BAM, upload only that shard, and gather — once — the per-record
    host fields and the read names to rank 0 (which unifies the range-local name ids).  Returns DeviceRecords; its host
    mirrors describe the whole file on rank 0 and are absent elsewhere.

    What travels: per rank ONE buffer of fixed-width columns + one read-name blob + its offsets (``HostMirrors.pack``) — no
    pickling, no text joins; rank 0 maps the columns in place and joins the name tables natively (``coral_names_unify``), so
    names stay bytes until something asks for a ``str``."""
    import time
    from . import bam
    from .records import DeviceRecords, HostMirrors
    t0 = time.perf_counter()
    rec = bam.load_bam(path, device, n_threads=n_threads, rank=rank, world=world)
    stats = dict(bam.LAST_DECODE)
    t1 = time.perf_counter()
    LAST_LOAD.clear()
    LAST_LOAD.update(decode_s=t1 - t0, gather_s=0.0, merge_s=0.0)
    if world == 1:
        dr = DeviceRecords(rec, device)
        dr.decode_stats = stats
        return dr

    class _G:                                  # what the gather helpers need before the DeviceRecords exists
        pass
    g = _G()
    g.rank, g.world, g.group, g.device = rank, world, group, torch.device(device)
    raw = _gather_to_rank0(g, HostMirrors.pack(HostMirrors.piece_of(rec)))
    counts = torch.tensor([rec.n], dtype=torch.int64, device=_comm_device(g))
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_counts = [int(c.item()) for c in all_counts]
    lo = sum(all_counts[:rank])
    t2 = time.perf_counter()
    host = None
    if rank == 0:
        host = HostMirrors.from_pieces([HostMirrors.unpack(r) for r in raw], n_threads=n_threads or bam.default_threads())
    t3 = time.perf_counter()
    LAST_LOAD.update(gather_s=t2 - t1, merge_s=t3 - t2)
    dr = DeviceRecords(rec, device, rank=rank, world=world, group=group, local_only=True, lo=lo, n_total=sum(all_counts), host=host)
    dr.decode_stats = stats
    return dr
