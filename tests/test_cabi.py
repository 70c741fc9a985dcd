"""The C-ABI library builds, loads, and exports every symbol include/coral_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "coral_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(coral_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for s in ("coral_cigar_scan", "coral_segment_coverage", "coral_point_cover", "coral_cluster_first_fit"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from coral_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), "missing export: " + s
    L.coral_version.restype = ctypes.c_char_p
    assert L.coral_version().startswith(b"coral_hip")


def test_host_entry_point_first_fit_clustering():
    """coral_cluster_first_fit is a pure host function: exact greedy first-fit (bu:268-282) incl. chaining."""
    import numpy as np
    from coral_amd import _lib
    L = _lib.lib()
    p1 = np.array([0, 1500, 3000, 100000, 2999, 4000, 100001, 1499], dtype=np.int64)
    p2 = np.array([0, 1500, 3000, 100000, 2999, 6000, 100001, 4000], dtype=np.int64)
    out = np.empty(len(p1), dtype=np.int32)
    n = ctypes.c_int32(0)
    _lib.check(L.coral_cluster_first_fit(len(p1), p1.ctypes.data, p2.ctypes.data, 2000, out.ctypes.data, ctypes.byref(n)), "ff")
    # 0,1500 chain into cluster 0; 3000 joins through 1500; 100000 new; (4000,6000): within 2000 of (3000,3000)? |3000| no -> new
    assert out.tolist() == [0, 0, 0, 1, 0, 2, 1, 0] and n.value == 3   # the last one matches (3000, 3000) of cluster 0 first


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    from coral_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(_lib.CoralHipError):
        _lib.lib()


def test_first_fit_equals_quadratic_definition():
    """The grid-indexed coral_cluster_first_fit is EXACTLY the greedy loop of bu:268-282 (checked against a direct O(n^2)
    transcription of its definition on random inputs with chaining, several cutoffs and negative coordinates)."""
    import numpy as np
    from coral_amd import _lib
    L = _lib.lib()

    def definition(p1, p2, cut):
        clusters, out = [], []
        for x, y in zip(p1, p2):
            home = -1
            for ci, c in enumerate(clusters):
                if any(abs(x - a) < cut and abs(y - b) < cut for a, b in c):
                    home = ci
                    break
            if home < 0:
                clusters.append([])
                home = len(clusters) - 1
            clusters[home].append((x, y))
            out.append(home)
        return out
    rng = np.random.default_rng(3)
    for trial in range(150):
        n = int(rng.integers(1, 100))
        cut = int(rng.choice([5, 50, 2000]))
        centres = rng.integers(-3 * cut, 20 * cut, size=(int(rng.integers(1, 6)), 2))
        c = centres[rng.integers(0, len(centres), n)]
        sp = int(rng.choice([1, cut // 2 + 1, cut, 3 * cut]))
        p1 = (c[:, 0] + rng.integers(-sp, sp + 1, n)).astype(np.int64)
        p2 = (c[:, 1] + rng.integers(-sp, sp + 1, n)).astype(np.int64)
        out = np.empty(n, dtype=np.int32)
        k = ctypes.c_int32(0)
        _lib.check(L.coral_cluster_first_fit(n, p1.ctypes.data, p2.ctypes.data, cut, out.ctypes.data, ctypes.byref(k)), "ff")
        exp = definition(p1.tolist(), p2.tolist(), cut)
        assert out.tolist() == exp and k.value == max(exp) + 1


def test_pyobjects_extension_builds_the_reference_containers():
    """coral_amd._pyobjects (CPython C API): same lists as the plain Python expressions, and loud errors."""
    import numpy as np
    import pytest
    from coral_amd import _pyobjects as P
    names = ["r%d" % k for k in range(50)]
    ids = np.array([3, 3, 49, 0], dtype=np.int64)
    i, j = np.array([0, 1, 2, 300], dtype=np.int64), np.array([1, 0, 5, -7], dtype=np.int64)
    assert P.names_of(names, ids) == [names[k] for k in ids]
    assert P.read_tuples(names, ids, i, j) == [(names[a], int(b), int(c)) for a, b, c in zip(ids, i, j)]
    assert P.names_of(names, np.zeros(0, dtype=np.int64)) == []
    with pytest.raises(IndexError):
        P.names_of(names, np.array([50], dtype=np.int64))
    with pytest.raises(TypeError):
        P.read_tuples(names, ids.astype(np.int32), i, j)
    with pytest.raises(ValueError):
        P.read_tuples(names, ids, i[:2], j)


def test_nm_stats_equals_sequential_python_sums():
    """coral_nm_stats: the reference's `+=` loop over mapped, SA-less MAPQ-60 records (ibg:153-157), bit for bit."""
    import ctypes as C
    import numpy as np
    from coral_amd import _lib
    rng = np.random.default_rng(5)
    n = 5000
    tid = rng.integers(-1, 3, n).astype(np.int32)
    mapq = rng.choice([0, 20, 60, 60, 60], n).astype(np.int32)
    nm = rng.integers(0, 900, n).astype(np.int32)
    qlen = rng.integers(500, 40000, n).astype(np.int32)
    sa_cnt = (rng.random(n) < 0.2) * rng.integers(1, 4, n)
    sa_off = np.concatenate([[0], np.cumsum(sa_cnt)]).astype(np.int64)
    cnt, s0, s1 = C.c_int64(0), C.c_double(0), C.c_double(0)
    assert _lib.lib().coral_nm_stats(n, tid.ctypes.data, sa_off.ctypes.data, mapq.ctypes.data, nm.ctypes.data, qlen.ctypes.data,
                                     C.byref(cnt), C.byref(s0), C.byref(s1)) == 0
    k, a, b = 0, 0.0, 0.0
    for i in range(n):
        if tid[i] >= 0 and sa_off[i + 1] == sa_off[i] and mapq[i] == 60:
            e = int(nm[i]) / int(qlen[i])
            a += e
            b += e * e
            k += 1
    assert (cnt.value, s0.value, s1.value) == (k, a, b) and k > 1000


def test_concordant_counts_equal_python_sets():
    """coral_concordant_counts vs the reference's set algebra (ibg:1043-1055): |(rls & rrs & rls1 & rrs1) - rbps| per edge."""
    import ctypes as C
    import numpy as np
    from coral_amd import _lib
    rng = np.random.default_rng(9)
    n_rec, n_names, n_edges = 4000, 1500, 23
    rec_name = rng.integers(0, n_names, n_rec).astype(np.int32)
    cover, sup, expect = [], [], []
    for q in range(n_edges):
        lists = [rng.choice(n_rec, size=int(rng.integers(0, 900)), replace=False) for _ in range(4)]
        if q % 7 == 0:
            lists[2] = lists[0]
        s_ = rng.integers(0, n_names, int(rng.integers(0, 300)))
        cover += lists
        sup.append(s_)
        sets = [set(rec_name[l].tolist()) for l in lists]
        expect.append(len((sets[0] & sets[1] & sets[2] & sets[3]) - set(s_.tolist())))
    pt_off = np.concatenate([[0], np.cumsum([len(c) for c in cover])]).astype(np.int64)
    pt_rec = np.concatenate(cover).astype(np.int32)
    pt_begin, pt_end = pt_off[:-1].copy(), pt_off[1:].copy()
    for q in range(0, n_edges, 7):                 # equal points share ONE slice of the record array
        pt_begin[4 * q + 2], pt_end[4 * q + 2] = pt_begin[4 * q], pt_end[4 * q]
    sup_off = np.concatenate([[0], np.cumsum([len(x) for x in sup])]).astype(np.int64)
    sup_all = np.concatenate(sup).astype(np.int64)
    L = _lib.lib()
    for trial in range(3):                         # (the per-name marks live on between calls: repeated calls must not see stale ones)
        out = np.zeros(n_edges, dtype=np.int64)
        assert L.coral_concordant_counts(n_edges, pt_begin.ctypes.data, pt_end.ctypes.data, pt_rec.ctypes.data, len(pt_rec), rec_name.ctypes.data,
                                         n_rec, n_names, sup_off.ctypes.data, sup_all.ctypes.data, out.ctypes.data) == 0
        assert out.tolist() == expect
    bad = pt_end.copy()
    bad[5] = len(pt_rec) + 1
    assert L.coral_concordant_counts(n_edges, pt_begin.ctypes.data, bad.ctypes.data, pt_rec.ctypes.data, len(pt_rec), rec_name.ctypes.data,
                                     n_rec, n_names, sup_off.ctypes.data, sup_all.ctypes.data, out.ctypes.data) != 0
    assert out.tolist() == expect and sum(expect) > 50


def test_count_distinct3():
    import numpy as np
    from coral_amd import _pyobjects as P
    rng = np.random.default_rng(2)
    for n in (0, 1, 7, 5000):
        a, b, c = (rng.integers(0, 40, n).astype(np.int64) for _ in range(3))
        assert P.count_distinct3(a, b, c) == len(set(zip(a.tolist(), b.tolist(), c.tolist())))


# ---- INTEGRATION.md must not drift from the header -------------------------------------------------
def _split_args(text):
    """Top-level comma split of a call's / prototype's argument text (nested parentheses and brackets respected)."""
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _calls(text, pattern):
    """(name, argument text) of every `pattern`(...) in text, with balanced parentheses."""
    for m in re.finditer(pattern, text):
        i, depth = m.end(), 1
        while depth and i < len(text):
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        yield m.group(1), text[m.end():i - 1]


def header_prototypes():
    text = open(os.path.join(ROOT, "include", "coral_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for name, args in _calls(text, r"\b(coral_[a-z0-9_]+)\s*\("):
        a = _split_args(args)
        protos[name] = 0 if a == ["void"] else len(a)
    return protos


def test_integration_md_calls_match_the_header():
    """Every `L.coral_*(...)` call printed in INTEGRATION.md §2 has exactly as many arguments as the header's prototype (a call
    shown with `...` is an explicit pointer to the header and only has to name an existing entry point)."""
    protos = header_prototypes()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    doc = "\n".join(line.split("  #")[0] if "L.coral_" in line or line.startswith(" ") else line for line in doc.splitlines())
    seen = 0
    for name, args in _calls(doc, r"\bL\.(coral_[a-z0-9_]+)\("):
        assert name in protos, "INTEGRATION.md calls %s, which include/coral_hip.h does not declare" % name
        if args.strip() in ("...", "h, *array_pointers"):
            continue
        n = len(_split_args(re.sub(r"#[^\n]*", "", args)))
        assert n == protos[name], "INTEGRATION.md calls %s with %d arguments, the header declares %d" % (name, n, protos[name])
        seen += 1
    assert seen >= 20
    assert "MAX_SEGS" not in doc


def test_ctypes_binding_matches_the_header():
    """coral_amd/_lib.py: every argtypes list is as long as the header's prototype."""
    from coral_amd import _lib
    L = _lib.lib()
    protos = header_prototypes()
    checked = 0
    for name, n in protos.items():
        f = getattr(L, name)
        if f.argtypes is not None:
            assert len(f.argtypes) == n, "%s: _lib.py binds %d arguments, the header declares %d" % (name, len(f.argtypes), n)
            checked += 1
    assert checked >= 30
