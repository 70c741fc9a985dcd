"""coral_pyset_*: the replay of CPython's set algorithm must give the iteration order of REAL sets of str built with the
same operations the reference uses (set([x]) + .add per key, then `acc = set(); acc |= s1; acc |= s2 ...`, iterate)."""
import ctypes as C
import random
import sys

import numpy as np
import pytest

from coral_amd import _lib


def emulated_order(entries, n_keys, union_keys, names):
    L = _lib.lib()
    hashes = np.array([hash(nm) for nm in names], dtype=np.int64)
    key = np.array([k for k, _ in entries], dtype=np.int32)
    item = np.array([i for _, i in entries], dtype=np.int32)
    counts = np.zeros(n_keys, dtype=np.int32)
    h = L.coral_pyset_batch_create(len(entries), key.ctypes.data, item.ctypes.data, hashes.ctypes.data, n_keys, counts.ctypes.data)
    assert h
    uk = np.array(union_keys, dtype=np.int32)
    out = np.empty(int(counts.sum()) + 1, dtype=np.int32)
    n = C.c_int32(0)
    _lib.check(L.coral_pyset_union_order(h, len(uk), uk.ctypes.data, out.ctypes.data, C.byref(n)), "union")
    L.coral_pyset_batch_free(h)
    return counts, out[:n.value].tolist()


def real_order(entries, n_keys, union_keys, names):
    sets = {}
    for k, i in entries:
        if k in sets:
            sets[k].add(names[i])
        else:
            sets[k] = set([names[i]])
    acc = set([])
    for k in union_keys:
        acc |= sets.get(k, set())
    idx = {nm: i for i, nm in enumerate(names)}
    return [len(sets.get(k, ())) for k in range(n_keys)], [idx[nm] for nm in acc]


@pytest.mark.parametrize("scale", [1, 2, 3])
def test_replay_matches_real_sets(scale):
    assert sys.version_info[:2] == (3, 10), "the replay is pinned to CPython 3.10's setobject.c"
    rnd = random.Random(1234 + scale)
    for trial in range(120):
        n_names = rnd.choice([3, 10, 60, 400, 3000, 70000][: 3 + scale])
        names = ["read%08d" % rnd.randrange(10 ** 8) for _ in range(n_names)]
        names = list(dict.fromkeys(names))
        n_keys = rnd.randint(1, 6)
        n_entries = rnd.choice([1, 5, 40, 300, 2500, 60000, 200000][: 4 + scale])
        entries = [(rnd.randrange(n_keys), rnd.randrange(len(names))) for _ in range(n_entries)]
        union_keys = [k for k in range(n_keys) if rnd.random() < 0.8] or [0]
        rnd.shuffle(union_keys) if trial % 3 == 0 else None
        c1, o1 = emulated_order(entries, n_keys, union_keys, names)
        c2, o2 = real_order(entries, n_keys, union_keys, names)
        assert c1.tolist() == c2
        assert o1 == o2, (trial, n_names, n_entries)


def test_replay_crosses_the_50000_resize_rule():
    names = ["r%d" % i for i in range(130000)]
    entries = [(0, i) for i in range(0, 130000, 2)] + [(1, i) for i in range(1, 130000, 3)]
    c1, o1 = emulated_order(entries, 2, [0, 1], names)
    c2, o2 = real_order(entries, 2, [0, 1], names)
    assert c1.tolist() == c2 and o1 == o2
    c1, o1 = emulated_order(entries, 2, [1, 0], names)
    c2, o2 = real_order(entries, 2, [1, 0], names)
    assert o1 == o2
