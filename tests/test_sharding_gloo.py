"""N > 1 path on CPU: two gloo ranks, records split in two, exchange = all-gather-v of candidate rows + all-reduce
of the per-segment sums.  The merged result must equal the reference golden (== the unsharded result)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.product_check import compare_graph_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(world, case, tmp_path, extra=()):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, PYTHONHASHSEED="0", RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2" if world <= 3 else "1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_shard_worker.py"), case, str(tmp_path)] + list(extra),
                                      env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]


@pytest.mark.parametrize("case,world", [("tiny_edge", 2), ("small", 3), ("small", 8)])
def test_per_rank_bam_decode_equals_reference(case, world, golden_dir, tmp_path):
    """The input side of the N > 1 path: the test writes a BAM, every rank decodes only ITS byte range of it and keeps only
    that shard; rank 0 gets the gathered per-record host fields with unified read-name ids.  Same golden as everywhere."""
    from coral_amd import bam, synth
    cfg, rec = synth.dataset(case, "cpu")
    path = str(tmp_path / "input.bam")
    bam.write_bam_native(rec, path, seed=7, n_threads=2)
    _run_ranks(world, case, tmp_path, extra=[path])
    with open(os.path.join(golden_dir, "e2e_%s.json" % case)) as fp:
        gold = json.load(fp)
    with open(tmp_path / "result.json") as fp:
        res = json.load(fp)
    assert res["normal_cov"] == gold["A2"]["normal_cov"]
    assert 0 < res["shard"][1] < res["shard"][2] == gold["n_records"] and res["world"] == world
    assert sorted(res["files"]) == sorted(gold["files"])
    for k in res["files"]:
        compare_graph_text(res["files"][k], gold["files"][k])


@pytest.mark.parametrize("case,world", [("tiny_edge", 2), ("small", 2), ("small", 8)])
def test_shard_merge_equals_reference(case, world, golden_dir, tmp_path):
    """Records already in memory, split over `world` ranks by CIGAR-op count (world 8 = one node's worth of ranks)."""
    _run_ranks(world, case, tmp_path)
    with open(os.path.join(golden_dir, "e2e_%s.json" % case)) as fp:
        gold = json.load(fp)
    with open(tmp_path / "result.json") as fp:
        res = json.load(fp)
    assert res["normal_cov"] == gold["A2"]["normal_cov"]
    assert 0 < res["shard"][1] < res["shard"][2]
    assert sorted(res["files"]) == sorted(gold["files"])
    for k in res["files"]:
        compare_graph_text(res["files"][k], gold["files"][k])
        assert open(str(tmp_path / ("sh" + k[3:]))).read() == res["files"][k]


@pytest.mark.gpu
def test_rccl_arms_at_world_one():
    """The RCCL arms of the exchange helpers (device tensors, backend "nccl") run — in the only RCCL world one GPU can form."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_world1.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl world-1 ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
