"""ctypes binding of libcoral_hip.so (include/coral_hip.h).  Fails loudly when the library is missing:
there is no CPU fallback on the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcoral_hip.so")


class CoralHipError(RuntimeError):
    pass


class coral_records_t(C.Structure):
    _fields_ = [("n_rec", C.c_int64), ("tid", C.c_void_p), ("pos", C.c_void_p), ("end", C.c_void_p),
                ("flagmq", C.c_void_p), ("n_cigar", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar", C.c_void_p)]


_lib = None

def lib():
    """Load (once) and return the shared library; raise CoralHipError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CoralHipError("libcoral_hip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "— the product path has no CPU fallback" % LIB_PATH)
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise CoralHipError("cannot load %s: %s" % (LIB_PATH, e))
    L.coral_version.restype = C.c_char_p
    L.coral_last_error.restype = C.c_char_p
    P = C.c_void_p
    R = C.POINTER(coral_records_t)
    L.coral_cigar_scan.argtypes = [R, C.c_int32, C.c_int32, P, P, P, C.c_uint32, P]
    L.coral_scan_kernel_name.restype = C.c_char_p
    L.coral_segment_coverage.argtypes = [R, P, C.c_int32, P, P, P, P, P, P, P, P]
    L.coral_point_cover.argtypes = [R, C.c_int32, P, P, C.c_int32, P, P, C.c_uint32, P]
    L.coral_read_counter.argtypes = [P, C.POINTER(C.c_uint32), P]
    L.coral_first_seen_rows.argtypes = [C.c_int64, C.c_int32, P, P]
    L.coral_first_seen_rows.restype = C.c_int
    L.coral_bp_pair_table.argtypes = [C.c_int32, C.c_int32, P, P, P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, P, P]
    L.coral_bp_pair_table.restype = C.c_int
    L.coral_search_create.argtypes = [C.c_int64, C.c_int64] + [P] * 9 + [C.c_int64, P, P, P, C.c_int32, P, P, P]
    L.coral_search_create.restype = C.c_void_p
    L.coral_search_free.argtypes = [C.c_void_p]
    L.coral_search_free.restype = C.c_int
    L.coral_search_error.argtypes = [C.c_void_p]
    L.coral_search_error.restype = C.c_char_p
    PI64, PF64, PI32 = C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_int32))
    L.coral_search_result.argtypes = [C.c_void_p, C.POINTER(C.c_int64), PI64, C.POINTER(C.c_int64), PI64, C.POINTER(C.c_int64), PI64,
                                      PF64, PI64, PI32]
    L.coral_search_result.restype = C.c_int
    L.coral_search_step.argtypes = [C.c_void_p] + [C.c_int64] * 5
    L.coral_search_step.restype = C.c_int
    L.coral_search_prefetch.argtypes = [C.c_void_p] + [C.c_int64] * 5
    L.coral_search_prefetch.restype = C.c_int
    L.coral_search_params.argtypes = [C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_int32]
    L.coral_search_params.restype = C.c_int
    L.coral_search_bfs.argtypes = [C.c_void_p, C.c_int32, P, P, P, P, P, C.c_double, C.c_int64, C.c_int32]
    L.coral_search_bfs.restype = C.c_int
    L.coral_search_bfs_get.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.coral_search_bfs_get.restype = C.c_int
    L.coral_search_within.argtypes = [C.c_void_p, C.c_int32, P, P, P]
    L.coral_search_within.restype = C.c_int
    L.coral_search_between.argtypes = [C.c_void_p, C.c_int64, P] + [C.c_int64] * 6
    L.coral_search_between.restype = C.c_int
    L.coral_sa_table.argtypes = [C.c_int32, P, P, P, P, C.c_int32, C.c_int32, P, P, P, P, C.c_int64, P, P, P, P, P,
                                 C.POINTER(C.c_int32), P]
    L.coral_sa_table.restype = C.c_int
    L.coral_sa_last_error.restype = C.c_char_p
    L.coral_hash_rows.argtypes = [C.c_int32, P, C.c_int32, P, P, P, P, P, C.c_int32, P, C.c_int64, P, P, P, P, C.POINTER(C.c_int32), P]
    L.coral_hash_rows.restype = C.c_int
    L.coral_pyset_batch_create.argtypes = [C.c_int64, P, P, P, C.c_int32, P]
    L.coral_pyset_batch_create.restype = C.c_void_p
    L.coral_pyset_union_order.argtypes = [C.c_void_p, C.c_int32, P, P, C.POINTER(C.c_int32)]
    L.coral_pyset_union_order.restype = C.c_int
    L.coral_pyset_batch_free.argtypes = [C.c_void_p]
    L.coral_pyset_batch_free.restype = C.c_int
    L.coral_call_breakpoints.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_int64, C.c_int64, C.c_double,
                                         C.c_int32] + [C.c_void_p] * 10
    L.coral_call_breakpoints.restype = C.c_int
    L.coral_nm_stats.argtypes = [C.c_int64] + [C.c_void_p] * 8
    L.coral_nm_stats.restype = C.c_int
    L.coral_reach_create.argtypes = [C.c_int64] + [C.c_void_p] * 6 + [C.c_int64] * 4 + [C.c_void_p, C.c_void_p]
    L.coral_reach_create.restype = C.c_void_p
    L.coral_reach_keys.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.coral_reach_keys.restype = C.c_int
    L.coral_concordant_counts.argtypes = [C.c_int32, P, P, P, C.c_int64, P, C.c_int64, C.c_int64, P, P, P]
    L.coral_concordant_counts.restype = C.c_int
    L.coral_independent_rows.argtypes = [C.c_int32, C.c_int32, P, P, C.c_double]
    L.coral_independent_rows.restype = C.c_int
    L.coral_cn_solve.argtypes = [C.c_int32, C.c_int32, P, P, P, P, C.c_int32, P, P, C.POINTER(C.c_int32)]
    L.coral_cn_solve.restype = C.c_int
    L.coral_cluster_first_fit.argtypes = [C.c_int64, P, P, C.c_int64, P, C.POINTER(C.c_int32)]
    L.coral_cluster_first_fit.restype = C.c_int
    L.coral_bam_decode_open.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]
    L.coral_bam_decode_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.coral_bam_decode_fill.argtypes = [C.c_void_p] + [P] * 21
    L.coral_bam_decode_close.argtypes = [C.c_void_p]
    L.coral_bam_last_error.restype = C.c_char_p
    L.coral_names_unify.argtypes = [C.c_int32, P, P, P, P, P, P, C.POINTER(C.c_int64), C.c_int32]
    L.coral_names_unify.restype = C.c_int
    L.coral_bam_decode_range.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.coral_bam_decode_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.coral_bam_write.argtypes = [C.c_char_p, C.c_int64] + [P] * 9 + [P, P, P, P, P, C.c_int64, P, P, P, C.c_int32, P, P, C.c_uint32,
                                                                        C.c_int32, C.c_int32]
    L.coral_bamgpu_open.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.coral_bamgpu_start.argtypes = [C.c_void_p, P, C.c_int64]
    L.coral_bamgpu_next.argtypes = [C.c_void_p, C.POINTER(C.c_int64), P]
    L.coral_bamgpu_emit.argtypes = [C.c_void_p, P, P, P]
    L.coral_bamgpu_host.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.coral_bamgpu_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.coral_bamgpu_close.argtypes = [C.c_void_p]
    L.coral_bgzf_inflate.argtypes = [P, P, C.c_int32, P, P, P]
    for name in ("coral_bamgpu_open", "coral_bamgpu_start", "coral_bamgpu_next", "coral_bamgpu_emit", "coral_bamgpu_host",
                 "coral_bamgpu_stats", "coral_bamgpu_close", "coral_bgzf_inflate"):
        getattr(L, name).restype = C.c_int
    for name in ("coral_bam_decode_open", "coral_bam_decode_sizes", "coral_bam_decode_fill", "coral_bam_decode_close",
                 "coral_bam_decode_range", "coral_bam_decode_stats", "coral_bam_write"):
        getattr(L, name).restype = C.c_int
    for name in ("coral_cigar_scan", "coral_segment_coverage", "coral_point_cover", "coral_read_counter"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        raise CoralHipError("%s failed (%d): %s" % (what, rc, lib().coral_last_error().decode()))


_pyset_checked = False


def check_pyset_replay():
    """The native replay of the interpreter's ``set`` (coral_pyset_* / coral_reach_*) reproduces CPython's table growth and
    probing rules; another interpreter (or a future CPython with a different set) would silently give a different — still
    valid, but not reference-identical — discordant-edge order.  So the replay is checked once per process against real sets
    of str on a few hundred seeded operations, and a mismatch is an error, not a fallback."""
    global _pyset_checked
    if _pyset_checked:
        return
    import random
    import numpy as np
    L = lib()
    rnd = random.Random(12345)
    names = ["read%07d_%d" % (rnd.randrange(10 ** 7), k) for k in range(900)]
    hashes = np.array([hash(nm) for nm in names], dtype=np.int64)
    n_keys = 7
    entries = [(rnd.randrange(n_keys), rnd.randrange(len(names))) for _ in range(2500)]
    key = np.array([k for k, _ in entries], dtype=np.int32)
    item = np.array([i for _, i in entries], dtype=np.int32)
    counts = np.zeros(n_keys, dtype=np.int32)
    h = L.coral_pyset_batch_create(len(entries), key.ctypes.data, item.ctypes.data, hashes.ctypes.data, n_keys, counts.ctypes.data)
    if not h:
        raise CoralHipError("coral_pyset_batch_create failed")
    try:
        sets = {}
        for k, i in entries:
            if k in sets:
                sets[k].add(names[i])
            else:
                sets[k] = set([names[i]])
        index = {nm: i for i, nm in enumerate(names)}
        for trial in range(6):
            ks = [rnd.randrange(n_keys) for _ in range(rnd.randrange(1, 5))]
            acc = set([])
            for k in ks:
                acc |= sets.get(k, set())
            uk = np.array(ks, dtype=np.int32)
            out = np.empty(len(names) + 1, dtype=np.int32)
            n = C.c_int32(0)
            check(L.coral_pyset_union_order(h, len(uk), uk.ctypes.data, out.ctypes.data, C.byref(n)), "coral_pyset_union_order")
            if out[:n.value].tolist() != [index[nm] for nm in acc] or [int(c) for c in counts] != [len(sets.get(k, ())) for k in range(n_keys)]:
                raise CoralHipError("the native replay of this interpreter's set iteration order does not match real sets "
                                    "(libcoral_hip replays CPython 3.10-3.12 sets); refusing to emit edges in a different order")
    finally:
        L.coral_pyset_batch_free(h)
    _pyset_checked = True
