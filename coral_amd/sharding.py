"""Multi-GPU sharding of the graph build: one process per GPU, torch.distributed over RCCL.

Records are split into ``world`` contiguous (tid, pos)-ordered ranges balanced by CIGAR-op count; every per-record
kernel runs on the local range only.  The path has two real exchange steps and only those use a collective:
  * all-gather-v of the compacted candidate rows (large-gap rows, point-cover pairs), tagged with GLOBAL record
    ordinals and re-sorted into the single-GPU (= reference) iteration order;
  * all-reduce(sum, int64) of the per-segment (n_reads, n_bases) vectors — integer sums, identical for any world.
Rank 0 runs the small order-sensitive host logic and tells the other ranks which kernel to run next by broadcasting
a tiny command object; ranks > 0 sit in ``serve``.  Works with backend "nccl" (RCCL over xGMI) and, for the CPU
tests, "gloo".
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_records(rec, rank: int, world: int, device, group=None):
    from .records import DeviceRecords
    return DeviceRecords(rec, device, rank=rank, world=world, group=group)


def command(dr, cmd):
    """Rank 0 -> all: the next kernel to run (called from the public wrappers in coral_amd.kernels)."""
    assert dr.rank == 0
    dist.broadcast_object_list([cmd], src=0, group=dr.group)


def _via_host(dr, t: torch.Tensor) -> bool:
    """gloo cannot all-gather device tensors: stage through host memory (CPU tests / single-GPU rehearsal only)."""
    return t.is_cuda and dist.get_backend(dr.group) == "gloo"


def allgather_rows(dr, rows: torch.Tensor) -> torch.Tensor:
    """All-gather-v of an int64 [k, c] row tensor (k differs per rank), concatenated in rank order."""
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
        box = [None]
        dist.broadcast_object_list(box, src=0, group=dr.group)
        cmd = box[0]
        if cmd[0] == "done":
            return
        if cmd[0] == "scan":
            scan = kernels.cigar_scan(dr, cmd[1], cmd[2], cmd[3], _worker=True)
        elif cmd[0] == "coverage":
            kernels.segment_coverage(dr, scan, cmd[1], _worker=True)
        elif cmd[0] == "points":
            kernels.point_cover(dr, cmd[1], cmd[2], _worker=True)
        else:
            raise RuntimeError("unknown command %r" % (cmd,))


def build_graph_sharded(dr, seedfile, cn_seg, output_prefix=None, min_bp_support=1.0, output_bp=False, gc_policy="pause"):
    """Graph build over sharded records: rank 0 returns the builder object, other ranks return None."""
    from . import infer_breakpoint_graph as ibg
    if dr.world == 1:
        return ibg.build_graph_from_records(dr, seedfile, cn_seg, output_prefix, min_bp_support, output_bp, gc_policy=gc_policy)
    if dr.rank == 0:
        try:
            b = ibg.build_graph_from_records(dr, seedfile, cn_seg, output_prefix, min_bp_support, output_bp, gc_policy=gc_policy)
        finally:
            dist.broadcast_object_list([("done",)], src=0, group=dr.group)
        return b
    serve(dr)
    return None
