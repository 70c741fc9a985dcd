#!/usr/bin/env python3
"""Headline benchmark: reads/s through the breakpoint-graph build on a synthetic long-read amplicon BAM.

    python bench.py --gpus 1 --steps K --warmup W [--config cfg3] [--reads N]

One "step" = one full pass of the hot path over the resident batch: decoded records already in HBM ->
BreakpointGraph objects + *_graph.txt written (Gurobi cycle step skipped), i.e. SURVEY.md §8(d)'s timed region.
Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: coral_cigar_scan, HIP-event timed on the launch
stream inside the timed steps) and `cpu_baseline` (the CPU oracle on a bounded sample of the same workload).
"""
import argparse
import json
import os

# the host side is one thread + a few native workers: per-core OpenMP / BLAS pools only spin (and, under a container CPU quota,
# get the process throttled); set before numpy / torch are imported
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1" if _v.startswith("OPENBLAS") else "4")
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# name of the kernel behind coral_cigar_scan per coral_set_scan_variant value (0 = library default = 15)
SCAN_KERNEL_NAME = {0: "k_cigar_scan_v2<8, false, true, 8>", 15: "k_cigar_scan_v2<8, false, true, 8>", 7: "k_cigar_scan_v2<8, false, true, 1>", 3: "k_cigar_scan_v2<8, false, false, 1>",
                    13: "k_cigar_scan_ring_asm<8, true>", 10: "k_cigar_scan_ring<8, true>", 8: "k_cigar_scan_packed<8>"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--reads", type=int, default=0, help="override the read count of the config (0 = as configured)")
    ap.add_argument("--cpu-sample", type=int, default=50000, help="reads in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU rehearsal of N > 1)")
    ap.add_argument("--shared-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--gc-policy", default="pause", choices=["pause", "freeze", "none"],
                    help="cyclic-GC handling around the build (build_graph_from_records); 'pause' is the library and CLI default")
    ap.add_argument("--mode", default="shard", choices=["shard", "samples"],
                    help="N > 1: 'shard' (default) = ONE sample, records sharded over the GPUs, exchange over RCCL (strong scaling); "
                         "'samples' = one independent sample per GPU, no collective on the data path (weak scaling, cohort use)")
    ap.add_argument("--scan-variant", type=int, default=0, help="A/B only: coral_set_scan_variant (0 = library default)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "launch with torch.distributed.run --nproc-per-node %d" % a.gpus
    if a.shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(a.backend)

    from coral_amd import synth, kernels
    from coral_amd import infer_breakpoint_graph as ibg
    from coral_amd import sharding

    if a.scan_variant:
        from coral_amd import _lib
        _lib.check(_lib.lib().coral_set_scan_variant(a.scan_variant), "coral_set_scan_variant")
    cfg = synth.named_config(a.config)
    if a.reads:
        cfg.n_reads = a.reads
    work = tempfile.mkdtemp(prefix="coral_bench_")
    cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
    synth.write_cn_bed(cfg, cn)
    synth.write_seed_bed(cfg, seeds)

    t0 = time.time()
    rec = synth.generate(cfg, dev, chunk_pieces=200000)
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    if a.mode == "samples" and world > 1:
        dr = sharding.shard_records(rec, 0, 1, dev)             # every rank holds (and builds) a whole sample of its own
        n_reads_total = cfg.n_reads * world
    else:
        dr = sharding.shard_records(rec, rank, world, dev)      # world == 1: all records on this GPU
        n_reads_total = cfg.n_reads
    alg_bytes_local = dr.algorithmic_bytes()
    del rec
    torch.cuda.empty_cache()

    def step(i):
        prefix = os.path.join(work, "r%d_s%d" % (rank, i))
        b = sharding.build_graph_sharded(dr, seeds, cn, prefix if (rank == 0 or a.mode == "samples") else None, gc_policy=a.gc_policy)
        return b

    kernels.PROFILE["scan_ms"] = []
    for i in range(a.warmup):
        step(-1 - i)
    kernels.PROFILE["scan_ms"] = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_ms, phases = [], []
    for i in range(a.steps):
        ts = time.perf_counter()
        b = step(i)
        step_ms.append(round((time.perf_counter() - ts) * 1e3, 1))
        phases.append(dict(ibg.PHASE_SECONDS))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    scan_ms = [e0.elapsed_time(e1) for e0, e1 in kernels.PROFILE["scan_ms"]]
    scan_ms_avg = sum(scan_ms) / max(1, len(scan_ms))

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = n_reads_total * a.steps / dt
        achieved = alg_bytes_local / (scan_ms_avg * 1e-3) / 1e9 if scan_ms_avg > 0 else 0.0
        out = {
            "metric": "reads/sec through breakpoint-graph build", "value": value, "unit": "reads/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak" if (a.mode == "samples" and world > 1) else "strong", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": "%s: %d reads x %d bp mean, %d seed intervals over %d chroms, full reconstruct incl. CN, "
                                   "cycle step skipped" % (cfg.name, cfg.n_reads, cfg.mean_len, len(cfg.seeds), len(cfg.windows)),
                       "records": int(dr.n_total), "cigar_ops": int(dr.total_ops_all), "amplicons": len(b.lr_graph),
                       "discordant_edges": sum(len(g.discordant_edges) for g in b.lr_graph),
                       "generate_s": round(gen_s, 2), "step_ms": step_ms, "gc_policy": a.gc_policy,
                       "parallelism": ("%d independent samples, one per GPU" % world) if (a.mode == "samples" and world > 1)
                       else "records sharded over %d GPU(s)" % world,
                       "phase_ms_median": {k: round(sorted(p.get(k, 0.0) for p in phases)[len(phases) // 2] * 1e3, 1)
                                           for k in (phases[-1] if phases else {})}},
            "roofline": {"bound": "hbm", "kernel": SCAN_KERNEL_NAME.get(a.scan_variant, "variant %d" % a.scan_variant), "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": pmc_traffic(cfg, world, SCAN_KERNEL_NAME.get(a.scan_variant)), "launch_ms": scan_ms_avg,
                         "algorithmic_bytes_per_launch": int(alg_bytes_local)},
        }
        if a.cpu_sample and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.config, min(a.cpu_sample, cfg.n_reads), work)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    shutil.rmtree(work, ignore_errors=True)


def pmc_traffic(cfg, world, kernel):
    """HBM bytes per scan launch from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json) when they
    were taken on exactly this workload on one GPU; counters cannot be collected from inside this process -> else None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fp:
            p = json.load(fp)
        if world == 1 and p["workload"] == cfg.name and p["n_reads"] == cfg.n_reads and p["scan_kernel"] == kernel:
            return p["kernels"][p["scan_kernel"]]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(config, n_sample, work):
    """The CPU oracle (a port of the reference's per-record Python loops; pysam itself is not installable) on the
    first ``n_sample`` reads of the same workload, one thread."""
    import torch
    from coral_amd import synth
    from oracle import coral_oracle as O
    from oracle.hostrecords import HostRecords
    torch.set_num_threads(1)
    cfg = synth.scaled_config(config, n_sample)
    rec = synth.generate(cfg, "cpu")
    cn, seeds = os.path.join(work, "cpu_cn.bed"), os.path.join(work, "cpu_seeds.bed")
    synth.write_cn_bed(cfg, cn)
    synth.write_seed_bed(cfg, seeds)
    host = HostRecords(rec)
    t0 = time.perf_counter()
    O.reconstruct_graph(host, seeds, cn, os.path.join(work, "cpu"))
    dt = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of %s (same generator and seed), decoded records in host memory -> graph files, %.1f s; "
                      "single-threaded like the reference (host has %d cores)" % (n_sample, config, dt, os.cpu_count() or 0)}


if __name__ == "__main__":
    main()
