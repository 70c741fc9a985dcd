"""Rank body of the world_size-2 gloo test (CPU): sharded graph build with oracle-backed local kernels."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Patch:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def main():
    import torch
    import torch.distributed as dist
    case, outdir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from coral_amd import sharding, synth
    from coral_amd.breakpoint_graph import graph_text
    from tests.product_check import install_cpu_kernel_fakes
    install_cpu_kernel_fakes(_Patch())
    cfg, rec = synth.dataset(case, "cpu")
    cn, seeds = os.path.join(outdir, "cn%d.bed" % rank), os.path.join(outdir, "seeds%d.bed" % rank)
    synth.write_cn_bed(cfg, cn)
    synth.write_seed_bed(cfg, seeds)
    if len(sys.argv) > 3:          # per-rank input: every rank decodes only its byte range of the BAM written by the test
        dr = sharding.load_bam_sharded(sys.argv[3], rank, world, "cpu", n_threads=2)
        assert dr.has_host == (rank == 0)
        if rank == 0:           # unified read-name ids = first appearance over the whole file, as a one-process decode numbers them
            want = rec.materialise_names()
            ids = rec.name_id.tolist()
            assert [dr.names[i] for i in dr.h_name_id.tolist()] == [want[i] for i in ids]
            first_seen = list(dict.fromkeys(want[i] for i in ids))
            assert dr.names == first_seen
    else:
        dr = sharding.shard_records(rec, rank, world, "cpu")
    if world <= 3:
        assert 0 < dr.n < dr.n_total
    else:                      # many ranks on a small file: a byte range may hold no record start at all
        assert 0 <= dr.n < dr.n_total
    b = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(outdir, "sh") if rank == 0 else None)
    if rank == 0:
        files = {"out_amplicon%d_graph.txt" % (i + 1): graph_text(g) for i, g in enumerate(b.lr_graph)}
        with open(os.path.join(outdir, "result.json"), "w") as fp:
            json.dump({"files": files, "normal_cov": b.normal_cov, "shard": [dr.lo, dr.hi, dr.n_total], "world": world,
                       "large_indel": len(b.large_indel_alignments)}, fp)
    else:
        assert b is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
