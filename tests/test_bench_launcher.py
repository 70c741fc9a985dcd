"""`python bench.py --gpus N` with no launcher around it (the way the driver starts every N) must start its own N rank
processes before anything touches a GPU, relay rank 0's JSON line on stdout and fail when a rank fails (VERDICT r02 missing #1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, cwd=ROOT, capture_output=True, text=True,
                          timeout=120)


def test_gpus_n_without_launcher_spawns_n_ranks():
    r = _run(["--gpus", "4", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    err = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["rank"] == 0 and out[0]["world"] == 4              # stdout carries rank 0's line only
    assert sorted(d["rank"] for d in err) == [1, 2, 3]
    assert all(d["world"] == 4 and d["local_rank"] == d["rank"] for d in out + err)
    assert len({d["master"] for d in out + err}) == 1 and out[0]["master"].startswith("127.0.0.1:")


def test_launcher_fails_when_a_rank_fails():
    r = _run(["--gpus", "2", "--dry-launch"], CORAL_BENCH_DRY_FAIL_RANK="1")
    assert r.returncode != 0
    assert "rank 1 exited with code 3" in r.stderr


def test_under_a_launcher_the_world_size_must_match():
    r = _run(["--gpus", "2", "--dry-launch"], WORLD_SIZE="4", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
