#!/usr/bin/env python3
"""Headline benchmark: reads/s through the breakpoint-graph build on a synthetic long-read amplicon BAM.

    python bench.py --gpus 1 --steps K --warmup W [--config cfg3] [--reads N] [--bam-reads M]

One "step" = one full pass of the hot path over the resident batch: decoded records already in HBM ->
BreakpointGraph objects + *_graph.txt written (Gurobi cycle step skipped), i.e. the contract's timed region.
Prints ONE JSON line (rank 0) with
  `roofline`      dominant kernel (coral_cigar_scan), HIP-event timed on the launch stream inside the timed steps;
  `h2d_ms` / `value_incl_h2d`   pinned host -> HBM of the resident arrays, measured once, and the rate with it added;
  `first_build_ms` / `value_cold`   the FIRST build of the process on fresh records (what one `CoRAL.py reconstruct` run pays: library
                  load, first launches, pinned pools, every per-records one-off), before any warm-up; `value` is the steady state;
  `decode`        BGZF/BAM decode throughput of the product's GPU decoder (coral_bamgpu_*) on a BAM FILE of the whole workload
                  (--bam-reads 0 = the configuration's read count: 2 M reads, 18.7 GB at config 3); `decode_host` the host pipeline
                  (coral_bam_decode_*) on the first tenth of the same file (a byte range; the whole file would take 17 s);
  `end_to_end`    BAM file -> graph files (decode of every rank's byte range + gather / merge on rank 0 + build), same BAM;
  `cpu_baseline`  the CPU oracle on a bounded sample of the same workload, resident (`value`) and from its BAM (`end_to_end`).
"""
import argparse
import json
import os

# the host side is one thread + a few native workers: per-core OpenMP / BLAS pools only spin (and, under a container CPU quota,
# get the process throttled); set before numpy / torch are imported
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1" if _v.startswith("OPENBLAS") else "4")
# hardware queues per device: ROCm's default is 4, and streams beyond that share queues — the decoder's three streams, the build's
# side streams and torch's own then serialise work that is meant to overlap (measured: the full-file decode 2.5 s instead of 1.9 s in
# this process).  Must be set before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--reads", type=int, default=0, help="override the read count of the config (0 = as configured)")
    ap.add_argument("--cpu-sample", type=int, default=50000, help="reads in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--bam-reads", type=int, default=0,
                    help="reads of the BAM the decode / end-to-end legs run on (same generator and layout); 0 (default) = the "
                         "configuration's own read count, i.e. the size the metric is quoted on; -1 = skip those legs")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU rehearsal of N > 1)")
    ap.add_argument("--shared-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--gc-policy", default="pause", choices=["pause", "freeze", "none"],
                    help="cyclic-GC handling around the build (build_graph_from_records); 'pause' is the library and CLI default")
    ap.add_argument("--mode", default="shard", choices=["shard", "samples"],
                    help="N > 1: 'shard' (default) = ONE sample, records sharded over the GPUs, exchange over RCCL (strong scaling); "
                         "'samples' = one independent sample per GPU, no collective on the data path (weak scaling, cohort use)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="N > 1 without a launcher: the rank processes only print their RANK / LOCAL_RANK / WORLD_SIZE and exit (CPU test)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without torch.distributed.run: this process becomes the launcher.  Nothing has touched the
        # GPU yet (torch is not even imported), and it never will here: the ranks are fresh child processes.
        sys.exit(launch_ranks(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under torch.distributed.run "
                 "with --nproc-per-node equal to --gpus)" % (a.gpus, world))
    if a.dry_launch:
        if os.environ.get("CORAL_BENCH_DRY_FAIL_RANK") == os.environ.get("RANK", "0"):      # (test hook: a rank that dies)
            sys.exit(3)
        print(json.dumps({"dry_launch": True, "rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "world": int(os.environ.get("WORLD_SIZE", "1")), "master": "%s:%s" % (os.environ.get("MASTER_ADDR"),
                                                                                                 os.environ.get("MASTER_PORT"))}), flush=True)
        return

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not a.shared_gpu and torch.cuda.device_count() < world:          # (counting devices does not initialise the GPU)
        sys.exit("bench.py: --gpus %d but only %d GPU(s) are visible (a one-GPU rehearsal of N > 1 is --shared-gpu --backend gloo)"
                 % (world, torch.cuda.device_count()))
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

    cfg = synth.named_config(a.config)
    if a.reads:
        cfg.n_reads = a.reads
    work = tempfile.mkdtemp(prefix="coral_bench_")
    cn, seeds = os.path.join(work, "cn.bed"), os.path.join(work, "seeds.bed")
    synth.write_cn_bed(cfg, cn)
    synth.write_seed_bed(cfg, seeds)

    t0 = time.time()
    rec = synth.generate(cfg, dev, chunk_pieces=200000)
    rec.names = rec.name_table()              # input construction: a decoded BAM brings its read names (blob + offsets) with it
    torch.cuda.synchronize()
    gen_s = time.time() - t0
    if a.mode == "samples" and world > 1:
        dr = sharding.shard_records(rec, 0, 1, dev)             # every rank holds (and builds) a whole sample of its own
        n_reads_total = cfg.n_reads * world
    else:
        dr = sharding.shard_records(rec, rank, world, dev)      # world == 1: all records on this GPU
        n_reads_total = cfg.n_reads
    alg_bytes_local = dr.algorithmic_bytes()
    if not (rank == 0 and a.bam_reads == 0 and a.mode == "shard"):
        rec = None                                                  # (rank 0 keeps the records: they become the BAM file of the legs below)
    torch.cuda.empty_cache()

    def step(i):
        prefix = os.path.join(work, "r%d_s%d" % (rank, i))
        b = sharding.build_graph_sharded(dr, seeds, cn, prefix if (rank == 0 or a.mode == "samples") else None, gc_policy=a.gc_policy)
        return b

    # the FIRST build of this process, on records nothing has been built from yet: what a one-shot `reconstruct` run pays.  It is
    # also the first of the W warm-up steps (with --warmup 0 it is an extra, untimed step).
    kernels.PROFILE["scan_ms"] = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    tf = time.perf_counter()
    b = step(-1)
    torch.cuda.synchronize()
    first_build_ms = (time.perf_counter() - tf) * 1e3
    first_phases = {k: round(v * 1e3, 1) for k, v in ibg.PHASE_SECONDS.items()}
    # (the warm-up keeps each result until the next one exists, exactly as the timed loop does: a build whose predecessor is
    # still alive needs a second set of pinned staging buffers, and pinning ~30 MB costs 15-20 ms the one time it happens —
    # that belongs to the warm-up, not to the second timed step, where rounds 2-3 had it)
    for i in range(1, a.warmup):
        b = step(-1 - i)
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
    kernels.PROFILE.pop("scan_ms", None)
    b_last = b
    h2d_ms = measure_h2d(dr) if a.mode == "shard" else None          # every rank uploads its own shard: max over ranks
    if world > 1 and h2d_ms is not None:
        tt = torch.tensor([h2d_ms], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        h2d_ms = float(tt.item())
    legs = None
    if a.bam_reads >= 0 and a.mode == "shard":
        try:
            legs = bam_legs(a, rank, world, dev, work, rec, cfg.n_reads)
        except Exception as exc:                      # noqa: BLE001 — the headline above must not be lost to a full disk or the like
            if world > 1:
                raise                                 # (ranks are inside collectives: no partial result there)
            legs = {"bam_legs_error": "%s: %s" % (type(exc).__name__, exc)}
    rec = None
    b = b_last

    if rank == 0:
        from coral_amd import _lib
        ms_per_step = dt / a.steps * 1e3
        value = n_reads_total * a.steps / dt
        achieved = alg_bytes_local / (scan_ms_avg * 1e-3) / 1e9 if scan_ms_avg > 0 else 0.0
        scan_kernel = _lib.lib().coral_scan_kernel_name().decode()
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
                       # Amdahl: per step, only the per-record kernels (scan + coverage + point cover) shrink with more GPUs;
                       # the order-sensitive host logic of rank 0 does not
                       "serial_ms": round(ms_per_step - scan_ms_avg, 1),
                       "phase_ms_median": {k: round(sorted(p.get(k, 0.0) for p in phases)[len(phases) // 2] * 1e3, 1)
                                           for k in (phases[-1] if phases else {})},
                       # (where the slowest timed step lost its time: steps are host-bound, so scheduling noise of the box shows)
                       "slowest_step_phase_ms": {k: round(v * 1e3, 1) for k, v in
                                                 (phases[max(range(len(step_ms)), key=step_ms.__getitem__)] if phases else {}).items()}},
            "roofline": {"bound": "hbm", "kernel": scan_kernel, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": pmc_traffic(cfg, world, _lib.lib().coral_version().decode()), "launch_ms": scan_ms_avg,
                         "algorithmic_bytes_per_launch": int(alg_bytes_local)},
        }
        out["first_build_ms"] = round(first_build_ms, 1)
        out["value_cold"] = n_reads_total / (first_build_ms * 1e-3)
        out["config"]["first_build_phase_ms"] = first_phases
        out["config"]["host_threads"] = {"build": "1 Python thread + %s look-ahead threads + %s helper threads of the interval search" %
                                         (os.environ.get("CORAL_SEARCH_THREADS", "6"), os.environ.get("CORAL_SEARCH_HELPERS", "8" if (os.cpu_count() or 1) >= 16 else "3" if (os.cpu_count() or 1) >= 8 else "1")),
                                         "decode": "see decode.host_threads"}
        out["config"]["library"] = _lib.lib().coral_version().decode()
        if h2d_ms is not None:
            out["h2d_ms"] = round(h2d_ms, 1)
            out["value_incl_h2d"] = n_reads_total / (ms_per_step * 1e-3 + h2d_ms * 1e-3)
        if legs:
            out.update(legs)
        if a.cpu_sample and world == 1:
            out["cpu_baseline"] = cpu_baseline(a.config, min(a.cpu_sample, cfg.n_reads), work)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    shutil.rmtree(work, ignore_errors=True)


def launch_ranks(n: int) -> int:
    """The launcher side of `python bench.py --gpus N` (no torch.distributed.run around it): start N rank processes of this same
    script — one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment — relay their output
    (rank 0's stdout, which carries the ONE JSON line, to stdout; everything else to stderr) and return non-zero if any rank
    fails.  Children are started with Popen from a process that has made no GPU call; nothing is exec'ed over a process."""
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))

    def relay():
        for line in procs[0].stdout:
            sys.stdout.write(line.decode(errors="replace"))
            sys.stdout.flush()

    t = threading.Thread(target=relay, daemon=True)
    t.start()
    rc = 0
    try:
        # a rank that dies leaves the others waiting in a collective: poll, and end the rest as soon as one has failed
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is not None:
                    alive.discard(r)
                    if code != 0:
                        sys.stderr.write("bench.py launcher: rank %d exited with code %d\n" % (r, code))
                        rc = rc or (code if code > 0 else 1)
            if rc and alive:
                time.sleep(5.0)
                for r in alive:
                    procs[r].terminate()
                for r in alive:
                    try:
                        procs[r].wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        procs[r].kill()
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    t.join(timeout=10)
    return rc


def measure_h2d(dr):
    """Pinned host -> HBM time of this rank's resident arrays (what a loader pays once per sample, outside `value`): the
    arrays are streamed through a 256 MiB pinned staging buffer, chunk by chunk, into the very tensors they came from."""
    import torch
    staging = torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True)
    total_ms = 0.0
    for t in (dr.cigar, dr.cigar_off, dr.tid, dr.pos, dr.end, dr.flagmq, dr.n_cigar):
        flat = t.view(-1).view(torch.uint8)
        for a in range(0, flat.numel(), staging.numel()):
            n = min(staging.numel(), flat.numel() - a)
            staging[:n].copy_(flat[a:a + n])                      # fill the staging buffer (not timed)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            flat[a:a + n].copy_(staging[:n], non_blocking=True)
            e1.record()
            e1.synchronize()
            total_ms += e0.elapsed_time(e1)
    return total_ms


def bam_legs(a, rank, world, dev, work, rec=None, rec_reads=0):
    """`decode`, `decode_host` and `end_to_end` on a real BAM FILE of the whole workload (default: the configuration's own read
    count — 2 M reads, 18.7 GB at config 3, the size the metric is quoted on).  Rank 0 writes the file (``rec``: the records the
    timed steps ran on, when the sizes agree) and times the whole-file decodes; then EVERY rank decodes only its byte range of it,
    rank 0 merges the per-record host fields, and the graph is built: the end-to-end time is barrier to barrier."""
    import torch
    import torch.distributed as dist
    from coral_amd import bam, sharding, synth
    full = synth.named_config(a.config)
    if a.reads:
        full.n_reads = a.reads
    n_bam = full.n_reads if a.bam_reads == 0 else min(a.bam_reads, full.n_reads)
    tmp_root = os.environ.get("CORAL_BENCH_TMP", "/tmp")
    # the file needs ~0.47 byte per read base (18.7 GB at config 3): shrink the read count to what the scratch disk holds,
    # with room to spare, and say so in the line (never the case on the boxes seen so far)
    need = lambda n: int(0.47 * n * full.mean_len * 1.25) + (1 << 30)
    free = shutil.disk_usage(tmp_root).free
    capped = False
    while n_bam > 50000 and need(n_bam) > free:
        n_bam //= 2
        capped = True
    cfg = synth.scaled_config(a.config, n_bam)
    shared = os.path.join(tmp_root, "coral_bench_bam_%s" % os.environ.get("MASTER_PORT", str(os.getpid())))
    path = os.path.join(shared, "input.bam")
    cn, seeds = os.path.join(shared, "cn.bed"), os.path.join(shared, "seeds.bed")
    out = {}
    if rank == 0:
        os.makedirs(shared, exist_ok=True)
        if rec is None or rec_reads != cfg.n_reads:
            rec = synth.generate(cfg, dev, chunk_pieces=200000)
        synth.write_cn_bed(cfg, cn)
        synth.write_seed_bed(cfg, seeds)
        rec_cpu = rec.to("cpu")
        n_records = rec.n
        del rec
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        bam.write_bam_native(rec_cpu, path, seed=1)
        write_s = time.perf_counter() - t0
        del rec_cpu
        size = os.path.getsize(path)
        # the product's decoder: inflate + parse on the GPU (coral_bamgpu_*); twice, the better run counts (the first one
        # carries one-off costs of the process)
        runs = []
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            whole = bam.decode_bam_gpu(path, dev)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0, dict(bam.LAST_DECODE)))
            assert whole.n == n_records
            del whole                                  # (its blocks stay in torch's cache: the second run does not pay hipMalloc again)
        dec_s, st = min(runs, key=lambda r: r[0])
        out["decode"] = {"reads_per_s": cfg.n_reads / dec_s, "where": "gpu", "host_threads": st["threads"],
                         "GB_per_s_compressed": size / dec_s / 1e9, "GB_per_s_inflated": st["uncompressed_bytes"] / dec_s / 1e9,
                         "seconds": round(dec_s, 3), "seconds_all_runs": [round(r[0], 3) for r in runs], "batches": st["batches"],
                         "stage_seconds": {k: round(st[k], 3) for k in ("read_seconds", "setup_seconds", "waited_for_file_seconds",
                                                                         "waited_for_gpu_seconds", "host_seconds")},
                         # the file was written seconds ago: the decoder reads it from the page cache, not from the disk
                         "file_in_page_cache": True, "n_reads": cfg.n_reads, "n_records": n_records,
                         "capped_by_free_disk_space": capped,
                         "bam": "%d reads (%d records) of %s, %.2f GB BGZF (zlib level 1), written in %.1f s; one process, whole file" % (
                             cfg.n_reads, n_records, a.config, size / 1e9, write_s)}
        # the host pipeline (coral_bam_decode_*: zlib inflate on all host threads this process may use): the first tenth of the
        # same file as a byte range (the whole 2 M-read file takes it ~17 s; reads/s over the records of that range)
        parts = 10 if cfg.n_reads >= 1000000 else 1
        t0 = time.perf_counter()
        part = bam.decode_bam(path, rank=0, world=parts)
        hdec_s = time.perf_counter() - t0
        hst = dict(bam.LAST_DECODE)
        out["decode_host"] = {"reads_per_s": part.n_names / hdec_s, "threads": hst["threads"],
                              "GB_per_s_inflated": hst["uncompressed_bytes"] / hdec_s / 1e9, "seconds": round(hdec_s, 2),
                              "sample": "byte range 1 of %d of the same file: %d records, %d read names" % (parts, part.n, part.n_names)}
        del part
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dr = sharding.load_bam_sharded(path, rank, world, dev)
    torch.cuda.synchronize()
    load = dict(sharding.LAST_LOAD)
    t1 = time.perf_counter()
    b = sharding.build_graph_sharded(dr, seeds, cn, os.path.join(shared, "e2e") if rank == 0 else None, gc_policy=a.gc_policy)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t2 = time.perf_counter()
    if world > 1:          # the slowest rank's decode of its byte range
        tt = torch.tensor([load["decode_s"]], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        load["decode_s_max_over_ranks"] = float(tt.item())
    if rank == 0:
        out["end_to_end"] = {"reads_per_s": cfg.n_reads / (t2 - t0), "seconds": round(t2 - t0, 3), "load_s": round(t1 - t0, 3),
                             "build_s": round(t2 - t1, 3), "n_gpus": world, "n_reads": cfg.n_reads,
                             "decode_s": round(load.get("decode_s_max_over_ranks", load["decode_s"]), 3),
                             "gather_s": round(load["gather_s"], 3), "merge_s": round(load["merge_s"], 3), "file_in_page_cache": True,
                             "what": "BAM file -> GPU decode (every rank its byte range: compressed bytes over PCIe, inflate + parse in HBM) -> "
                                     "per-record host fields + name blobs gathered to rank 0 and merged natively -> graph files; the build is "
                                     "the first one on these records"}
        shutil.rmtree(shared, ignore_errors=True)
    return out if rank == 0 else None


def pmc_traffic(cfg, world, library_version):
    """HBM bytes per scan launch from a committed rocprofv3 --pmc run (profiles/r*_pmc_traffic*.json, tools/pmc_traffic.sh) — served
    only when it was taken on exactly this workload, on one GPU, with a library built from THIS coral_kernels.hip (the library's
    version string carries the source's hash, and so does the profile): a figure measured on another kernel yields None.
    Counters cannot be collected from inside this process."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json"))):
        try:
            with open(path) as fp:
                p = json.load(fp)
            if world == 1 and p["workload"] == cfg.name and p["n_reads"] == cfg.n_reads and p.get("library") == library_version:
                best = p["kernels"][p["scan_kernel"]]["hbm_bytes"]
        except (OSError, KeyError, ValueError, TypeError):
            pass
    return best


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
    # the same from a BAM file: one decoder thread (the reference's pysam reads with one thread) + the same build
    from coral_amd import bam
    path = os.path.join(work, "cpu_sample.bam")
    bam.write_bam_native(rec, path, seed=1)
    t0 = time.perf_counter()
    back = bam.decode_bam(path, n_threads=1)
    t1 = time.perf_counter()
    O.reconstruct_graph(HostRecords(back), seeds, cn, os.path.join(work, "cpu_e2e"))
    t2 = time.perf_counter()
    return {"value": n_sample / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "end_to_end": {"reads_per_s": n_sample / (t2 - t0), "decode_s": round(t1 - t0, 2), "build_s": round(t2 - t1, 2),
                           "what": "BAM file -> one-thread decode -> CPU oracle -> graph files"},
            "sample": "first %d reads of %s (same generator and seed), decoded records in host memory -> graph files, %.1f s; "
                      "single-threaded like the reference (host has %d cores)" % (n_sample, config, dt, os.cpu_count() or 0)}


if __name__ == "__main__":
    main()
