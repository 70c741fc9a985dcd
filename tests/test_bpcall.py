"""coral_call_breakpoints (one native call) against the step-by-step host functions cluster_bp_list / bpc2bp, which are
themselves pinned to the reference by tests/golden/unit_vectors.json (test_host_logic.py)."""
import numpy as np
import pytest

from coral_amd.bpcluster import bpc2bp, call_breakpoints, cluster_bp_list
from coral_amd.chimeric import Candidates


def _stepwise(c, mcc, dist, cut, floor, advance):
    sizes, calls = [], []
    for cl in cluster_bp_list(c, mcc, dist):
        sizes.append(len(cl))
        if len(cl) < mcc:
            continue
        sub, rest = 0, cl
        while len(rest) >= mcc and len(rest):
            head = rest[0]
            p1, p2, sup, st, rest = bpc2bp(c, rest, cut)
            n_sup = len(set(zip(c.read[sup].tolist(), c.i[sup].tolist(), c.j[sup].tolist())))
            if (sub == 0 and n_sup >= mcc) or n_sup >= floor:
                calls.append((int(head), p1, p2, sup.tolist(), st))
            if advance:
                sub += 1
    return sizes, calls


def _random_candidates(rng, n, spread):
    centres = rng.integers(1000, 5_000_000, size=(rng.integers(1, 5), 2))
    which = rng.integers(0, len(centres), n)
    jitter = lambda: np.where(rng.random(n) < 0.6, 0, rng.integers(-spread, spread + 1, n))
    kw = dict(c1=rng.integers(0, 3, n), c2=rng.integers(0, 3, n), o1=rng.integers(0, 2, n), o2=rng.integers(0, 2, n),
              p1=centres[which, 0] + jitter(), p2=centres[which, 1] + jitter(),
              read=rng.integers(0, max(2, n // 2), n), i=rng.integers(0, 3, n), j=rng.integers(0, 3, n),
              gap=np.where(rng.random(n) < 0.5, 0, rng.integers(-50, 400, n)), swapped=rng.integers(0, 2, n),
              mqa=rng.integers(0, 61, n), mqb=rng.integers(0, 61, n))
    if rng.random() < 0.5:                      # few groups: big clusters, ties between modes, sub-clusters
        kw["c1"][:] = 1; kw["c2"][:] = 2; kw["o1"][:] = rng.integers(0, 2); kw["o2"][:] = rng.integers(0, 2)
    return Candidates(**kw)


@pytest.mark.parametrize("advance", [False, True])
def test_native_call_equals_stepwise(advance):
    rng = np.random.default_rng(11 + advance)
    n_calls = 0
    for trial in range(300):
        n = int(rng.integers(1, 400))
        c = _random_candidates(rng, n, int(rng.choice([3, 60, 150, 900])))
        mcc = float(rng.choice([1.0, 3.0, 3.5, 8.0]))
        floor = max(3.0, float(rng.choice([1.0, 4.2, 9.0])))
        a = _stepwise(c, mcc, 2000, 100, floor, advance)
        b = call_breakpoints(c, mcc, 2000, 100, floor, advance)
        assert a[0] == b[0], trial
        assert len(a[1]) == len(b[1]), trial
        for x, y in zip(a[1], b[1]):
            assert x[:3] == y[:3], trial
            assert x[3] == y[3].tolist(), trial
            assert x[4] == y[4] and [type(v) for v in x[4]] == [type(v) for v in y[4]], (trial, x[4], y[4])
        n_calls += len(b[1])
    assert n_calls > 300


def test_strided_columns_and_empty():
    assert call_breakpoints(Candidates(), 3.0, 2000, 100, 3.0, True) == ([], [])
    rows = np.zeros((40, 13), dtype=np.int64)
    rows[:, 1] = 5000 + np.arange(40) % 3
    rows[:, 4] = 90000 - np.arange(40) % 2
    rows[:, 6] = np.arange(40)
    c = Candidates(**{k: rows[:, j] for j, k in enumerate(Candidates.FIELDS)})   # column views: stride 13
    sizes, calls = call_breakpoints(c, 3.0, 2000, 100, 3.0, True)
    assert sizes == [40] and len(calls) == 1 and calls[0][1:3] == (5000, 90000) and len(calls[0][3]) == 40


def test_contig_ids_beyond_64_do_not_collide():
    """BAM headers may list thousands of contigs (decoys, alts) before the chromosome of interest: candidates whose contig
    ids differ by multiples of 64 must stay in separate groups (bu:257-262 groups by the chromosome NAMES)."""
    rng = np.random.default_rng(5)
    n = 240
    tids = np.array([6, 70, 134, 200, 3000, 3064])                 # 70 = 6 + 64, 134 = 6 + 128, 3064 = 3000 + 64
    kw = dict(c1=rng.choice(tids, n), c2=rng.choice(tids, n), o1=rng.integers(0, 2, n), o2=rng.integers(0, 2, n),
              p1=50_000 + rng.integers(-40, 41, n), p2=900_000 + rng.integers(-40, 41, n),
              read=np.arange(n), i=np.zeros(n, dtype=np.int64), j=np.ones(n, dtype=np.int64), gap=np.zeros(n, dtype=np.int64),
              swapped=np.zeros(n, dtype=np.int64), mqa=np.full(n, 60), mqb=np.full(n, 60))
    c = Candidates(**kw)
    groups = {}
    for k in range(n):                                               # the reference's grouping, literally
        groups.setdefault((int(c.c1[k]), int(c.c2[k]), int(c.o1[k]), int(c.o2[k])), []).append(k)
    expected = [g for g in groups.values()]                          # every group is within 2000 bp: one cluster each
    got = [cl.tolist() for cl in cluster_bp_list(c, 1.0, 2000)]
    assert got == expected
    sizes, calls = call_breakpoints(c, 1.0, 2000, 100, 3.0, True)
    assert sizes == [len(g) for g in expected]
    for (head, p1, p2, sup, st) in calls:
        key = (int(c.c1[head]), int(c.c2[head]), int(c.o1[head]), int(c.o2[head]))
        assert all((int(c.c1[k]), int(c.c2[k]), int(c.o1[k]), int(c.o2[k])) == key for k in sup.tolist())
