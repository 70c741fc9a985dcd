"""Read names as a byte blob + offsets (coral_amd.names.NameTable), the interpreter's str hash computed from the bytes, and the
native join of per-rank name tables (coral_names_unify) — against plain Python str / dict on the same inputs."""
import ctypes as C
import random

import numpy as np
import pytest

from coral_amd.names import NameTable
from coral_amd.records import HostMirrors


def _rand_names(rnd, n, alphabet="abcdefgh0123456789_/:.-", lo=1, hi=40):
    out = set()
    while len(out) < n:
        out.add("".join(rnd.choice(alphabet) for _ in range(rnd.randint(lo, hi))))
    return sorted(out, key=lambda s: rnd.random())


def test_name_table_is_a_sequence_of_str():
    names = ["read%08d" % i for i in range(1000)] + ["", "x", "a/b:c.1", "naïve", "ß" * 7]
    t = NameTable.from_list(names)
    assert len(t) == len(names) and t == names and names == list(t) and not (t != names)
    assert t[3] == names[3] and t[-1] == names[-1] and t[10:13] == names[10:13]
    ids = np.array([5, 0, 1003, 1001, 5], dtype=np.int64)
    assert t.take(ids) == [names[i] for i in ids]
    assert t.tuples(ids, ids + 1, ids * 2) == [(names[i], int(i) + 1, int(i) * 2) for i in ids]
    with pytest.raises(IndexError):
        t.take(np.array([len(names)]))
    with pytest.raises(IndexError):
        t[len(names)]
    assert t.index_map()["x"] == 1001 and t.index("a/b:c.1") == 1002
    assert NameTable.from_list([]) == [] and len(NameTable.from_list([])) == 0
    # made once, then the same objects
    assert t.take(ids)[0] is t.tolist()[5]


def test_hashes_are_this_interpreters_str_hashes():
    rnd = random.Random(5)
    names = _rand_names(rnd, 3000) + ["", "ü", "日本語", "a" * 300, "read\t1", "é" + "x" * 20]
    t = NameTable.from_list(names)
    ids = np.array(rnd.sample(range(len(names)), 2000) + [len(names) - k for k in range(1, 7)], dtype=np.int64)
    assert t.hashes(ids).tolist() == [hash(names[i]) for i in ids]
    assert t.hashes(np.zeros(0, dtype=np.int64)).tolist() == []


def test_from_decimal_equals_percent_formatting():
    v = np.array([0, 7, 12345678, 99999999, 42], dtype=np.int64)
    assert NameTable.from_decimal("read", v, 8) == ["read%08d" % int(x) for x in v]
    assert NameTable.from_decimal("read", np.array([10 ** 8]), 8) is None           # more digits than the width: caller falls back


def _unify(pieces, threads):
    """pieces: lists of names in piece-local first-seen order -> (luts, global names) through HostMirrors.from_pieces."""
    ds = []
    for loc in pieces:
        t = NameTable.from_list(loc)
        z32, z64 = np.zeros(len(loc), dtype=np.int32), np.zeros(len(loc), dtype=np.int64)
        ds.append(dict(tid=z32, pos=z32, end=z32, flag=z32, mapq=z32, qlen=z32, has_seq=z32.astype(np.uint8), nm=z32,
                       name_id=np.arange(len(loc), dtype=np.int32), n_cigar=z32, sa_count=z64, sa=np.zeros((0, 8), dtype=np.int32),
                       sa_nm=z32[:0], nonacgt_rec=z64[:0], nonacgt_pos=z32[:0], name_blob=t.blob, name_off=t.off))
    host = HostMirrors.from_pieces([HostMirrors.unpack(HostMirrors.pack(d)) for d in ds], n_threads=threads)
    return host


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_unify_numbers_names_by_first_appearance_over_the_file(threads):
    rnd = random.Random(11 + threads)
    for trial in range(12):
        universe = _rand_names(rnd, rnd.choice([1, 5, 300, 70000 if trial < 2 else 2000]))
        n_pieces = rnd.choice([1, 2, 3, 8])
        pieces = []
        for p in range(n_pieces):
            k = rnd.randint(0, len(universe))
            loc = rnd.sample(universe, k)                     # unique within the piece, overlapping between pieces
            pieces.append(loc if rnd.random() > 0.15 else [])
        host = _unify(pieces, threads)
        want = list(dict.fromkeys(nm for loc in pieces for nm in loc))
        assert host._names == want and host.n_names == len(want)
        index = {nm: i for i, nm in enumerate(want)}
        assert host.h_name_id.tolist() == [index[nm] for loc in pieces for nm in loc]


def test_unify_compares_bytes_not_hashes():
    """Names that differ only in length / in a trailing NUL-like byte / by one byte far into the string stay distinct; equal ones join."""
    a = ["r", "r\x01", "rr", "x" * 100 + "a", "x" * 100 + "b", ""]
    b = ["x" * 100 + "b", "", "r\x01", "rrr", "r"]
    host = _unify([a, b], 2)
    assert host._names == a + ["rrr"]
    assert host.h_name_id.tolist() == [0, 1, 2, 3, 4, 5, 4, 5, 1, 6, 0]


def test_pack_unpack_round_trip_and_layout():
    rnd = np.random.default_rng(3)
    n = 1234
    t = NameTable.from_list(["q%d" % i for i in range(900)])
    d = dict(tid=rnd.integers(0, 24, n).astype(np.int32), pos=rnd.integers(0, 1 << 30, n).astype(np.int32),
             end=rnd.integers(0, 1 << 30, n).astype(np.int32), flag=rnd.integers(0, 4096, n).astype(np.int32),
             mapq=rnd.integers(0, 61, n).astype(np.int32), qlen=rnd.integers(0, 1 << 20, n).astype(np.int32),
             has_seq=rnd.integers(0, 2, n).astype(np.uint8), nm=rnd.integers(0, 999, n).astype(np.int32),
             name_id=rnd.integers(0, 900, n).astype(np.int32), n_cigar=rnd.integers(0, 70000, n).astype(np.int32),
             sa_count=rnd.integers(0, 3, n).astype(np.int64), sa=rnd.integers(0, 1 << 20, (77, 8)).astype(np.int32),
             sa_nm=rnd.integers(0, 99, 77).astype(np.int32), nonacgt_rec=rnd.integers(0, n, 13).astype(np.int64),
             nonacgt_pos=rnd.integers(0, 1 << 30, 13).astype(np.int32), name_blob=t.blob, name_off=t.off)
    buf = HostMirrors.pack(d)
    assert buf.dtype == np.uint8 and len(buf) % 8 == 0
    back = HostMirrors.unpack(buf)
    for k, dt in HostMirrors.PIECE_SPEC:
        assert back[k].dtype == np.dtype(dt) and np.array_equal(back[k], d[k]), k
    assert back["n_names"] == 900


def test_unify_rejects_bad_arguments():
    from coral_amd import _lib
    L = _lib.lib()
    n = C.c_int64(0)
    off = np.zeros(1, dtype=np.int64)
    assert L.coral_names_unify(0, None, None, None, None, None, off.ctypes.data, C.byref(n), 1) == 0 and n.value == 0
    assert L.coral_names_unify(1, None, None, None, None, None, off.ctypes.data, C.byref(n), 1) != 0
    assert L.coral_names_unify(-1, None, None, None, None, None, off.ctypes.data, C.byref(n), 1) != 0
