"""One-off parity check at FULL config-3 size: product (HIP kernels + native host pieces + look-ahead threads) against the CPU
oracle on the same 2 M reads.  Takes ~6 minutes of CPU for the oracle; run with PYTHONHASHSEED=0 for the strict text comparison.

    PYTHONHASHSEED=0 python tools/validate_full_size.py [n_reads [config [bam]]] > gpurun_out/full_size_parity.txt

With a third argument `bam` the records first become a BAM FILE: the product decodes it on the GPU (coral_bamgpu_*) and builds from
that; the oracle works on the host pipeline's decode of the same file (both decodes are also compared field by field).
"""
import os, sys, time, tempfile
sys.path.insert(0, ".")
import torch
from coral_amd import synth, sharding
from coral_amd.breakpoint_graph import graph_text
from oracle import coral_oracle as O
from oracle.hostrecords import HostRecords
from tests.product_check import compare_graph_text

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
name = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
cfg = synth.scaled_config(name, n)
work = tempfile.mkdtemp()
cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
synth.write_cn_bed(cfg, cn); synth.write_seed_bed(cfg, seeds)
t = time.time(); rec = synth.generate(cfg, "cuda:0", chunk_pieces=200000); torch.cuda.synchronize()
print("generated %d records in %.1fs" % (rec.n, time.time() - t), flush=True)
from_bam = len(sys.argv) > 3 and sys.argv[3] == "bam"
if from_bam:
    import numpy as np
    from coral_amd import bam
    from coral_amd.records import DeviceRecords
    from tests.test_bam_io import FIELDS
    path = os.path.join(work, "input.bam")
    rec_cpu = rec.to("cpu"); del rec; torch.cuda.empty_cache()
    t = time.time(); bam.write_bam_native(rec_cpu, path, seed=1); print("BAM written: %.2f GB in %.1fs" % (os.path.getsize(path) / 1e9, time.time() - t), flush=True)
    del rec_cpu
    t = time.time(); rec = bam.decode_bam_gpu(path, "cuda:0"); torch.cuda.synchronize(); td = time.time() - t
    print("GPU decode: %.2fs (%.0f reads/s, %d batches)" % (td, n / td, bam.LAST_DECODE["batches"]), flush=True)
    t = time.time(); rec_cpu = bam.decode_bam(path); print("host decode: %.2fs" % (time.time() - t), flush=True)
    assert rec.n == rec_cpu.n and rec.names == rec_cpu.names
    for k in FIELDS:
        assert np.array_equal(getattr(rec, k).cpu().numpy(), getattr(rec_cpu, k).cpu().numpy()), k
    print("GPU decode == host decode: %d records, every field" % rec.n, flush=True)
    dr = DeviceRecords(rec, "cuda:0")
else:
    dr = sharding.shard_records(rec, 0, 1, "cuda:0")
t = time.time(); b = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(work, "gpu")); tg = time.time() - t
print("product: %.2fs, %d amplicons, %d breakpoints, %d chimeric reads" % (tg, len(b.lr_graph), len(b.new_bp_list), len(b.chimeric_alignments)), flush=True)
if not from_bam:
    rec_cpu = rec.to("cpu")
del rec, dr; torch.cuda.empty_cache()
t = time.time(); host = HostRecords(rec_cpu); print("host records %.1fs" % (time.time() - t), flush=True)
t = time.time(); ob, ofiles = O.reconstruct_graph(host, seeds, cn); to = time.time() - t
print("oracle: %.1fs (%.0f reads/s), %d amplicons, %d breakpoints" % (to, n / to, len(ob.lr_graph), len(ob.new_bp_list)), flush=True)
assert len(b.lr_graph) == len(ob.lr_graph) and b.normal_cov == ob.normal_cov
assert sorted(map(str, b.amplicon_intervals)) == sorted(map(str, ob.amplicon_intervals))
strict = os.environ.get("PYTHONHASHSEED") == "0"
for g, og in zip(b.lr_graph, ob.lr_graph):
    assert [e[:8] for e in g.sequence_edges] == [e[:8] for e in og.sequence_edges]
    assert [e[8] for e in g.concordant_edges] == [e[8] for e in og.concordant_edges]
    assert sorted(map(str, (e[:6] + [e[9]] for e in g.discordant_edges))) == sorted(map(str, (e[:6] + [e[9]] for e in og.discordant_edges)))
    assert sorted(map(sorted, (e[10] for e in g.discordant_edges))) == sorted(map(sorted, (e[10] for e in og.discordant_edges)))
    if strict:
        assert [e[:6] for e in g.discordant_edges] == [e[:6] for e in og.discordant_edges]      # the set-order dependent edge order
        compare_graph_text(graph_text(g), O.graph_text(og))
        assert b.new_bp_stats == ob.new_bp_stats
print("PARITY OK (" + name + ") at %d reads: sequence / concordant / discordant edges, supports and read sets identical%s" % (
    n, "; discordant-edge ORDER, breakpoint statistics and graph text identical (CN within 1e-6)" if strict else ""))
print("speed-up of this run: %.0fx (oracle %.1fs vs product %.2fs, first product call incl. warm-up)" % (to / tg, to, tg))
