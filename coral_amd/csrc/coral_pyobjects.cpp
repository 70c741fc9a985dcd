// coral_pyobjects.cpp — CPython extension `coral_amd._pyobjects`: builds the Python containers the reference's object
// surface requires (lists of read names, lists of (name, i, j) support tuples) straight from index arrays.
// These are the containers of /root/reference/src/breakpoint_utilities.py:81 / :294 (the `r` tuple of a candidate) and of
// infer_breakpoint_graph.py:1047-1049 (name sets of concordant edges); building them with the C API instead of
// itemgetter + zip costs a quarter of the time.  No algorithm lives here.
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

namespace {
struct I64View {
    Py_buffer buf;
    bool ok = false;
    const int64_t *p = nullptr;
    Py_ssize_t n = 0;
    bool get(PyObject *o, const char *what) {
        if (PyObject_GetBuffer(o, &buf, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return false;
        ok = true;
        if (buf.itemsize != 8 || buf.ndim != 1) {
            PyErr_Format(PyExc_TypeError, "%s must be a contiguous one-dimensional int64 array", what);
            return false;
        }
        p = static_cast<const int64_t *>(buf.buf);
        n = buf.shape[0];
        return true;
    }
    ~I64View() { if (ok) PyBuffer_Release(&buf); }
};

// names_of(names: list, ids) -> [names[k] for k in ids]
PyObject *names_of(PyObject *, PyObject *args) {
    PyObject *names, *ids_o;
    if (!PyArg_ParseTuple(args, "O!O", &PyList_Type, &names, &ids_o)) return nullptr;
    I64View ids;
    if (!ids.get(ids_o, "ids")) return nullptr;
    const Py_ssize_t n_names = PyList_GET_SIZE(names);
    PyObject *out = PyList_New(ids.n);
    if (!out) return nullptr;
    for (Py_ssize_t k = 0; k < ids.n; ++k) {
        const int64_t id = ids.p[k];
        if (id < 0 || id >= n_names) {
            Py_DECREF(out);
            PyErr_SetString(PyExc_IndexError, "name id out of range");
            return nullptr;
        }
        PyObject *s = PyList_GET_ITEM(names, id);
        Py_INCREF(s);
        PyList_SET_ITEM(out, k, s);
    }
    return out;
}

// read_tuples(names: list, ids, i, j) -> [(names[ids[k]], int(i[k]), int(j[k])) for k in range(len(ids))]
PyObject *read_tuples(PyObject *, PyObject *args) {
    PyObject *names, *ids_o, *i_o, *j_o;
    if (!PyArg_ParseTuple(args, "O!OOO", &PyList_Type, &names, &ids_o, &i_o, &j_o)) return nullptr;
    I64View ids, vi, vj;
    if (!ids.get(ids_o, "ids") || !vi.get(i_o, "i") || !vj.get(j_o, "j")) return nullptr;
    if (vi.n != ids.n || vj.n != ids.n) {
        PyErr_SetString(PyExc_ValueError, "ids, i and j must have the same length");
        return nullptr;
    }
    const Py_ssize_t n_names = PyList_GET_SIZE(names);
    PyObject *out = PyList_New(ids.n);
    if (!out) return nullptr;
    for (Py_ssize_t k = 0; k < ids.n; ++k) {
        const int64_t id = ids.p[k];
        if (id < 0 || id >= n_names) {
            Py_DECREF(out);
            PyErr_SetString(PyExc_IndexError, "name id out of range");
            return nullptr;
        }
        PyObject *a = PyLong_FromLongLong(vi.p[k]);
        PyObject *b = a ? PyLong_FromLongLong(vj.p[k]) : nullptr;
        PyObject *t = b ? PyTuple_New(3) : nullptr;
        if (!t) {
            Py_XDECREF(a);
            Py_XDECREF(b);
            Py_DECREF(out);
            return nullptr;
        }
        PyObject *s = PyList_GET_ITEM(names, id);
        Py_INCREF(s);
        PyTuple_SET_ITEM(t, 0, s);
        PyTuple_SET_ITEM(t, 1, a);
        PyTuple_SET_ITEM(t, 2, b);
        PyList_SET_ITEM(out, k, t);
    }
    return out;
}

// count_distinct3(a, b, c) -> number of distinct (a[k], b[k], c[k]) triples  (= len(set(zip(a, b, c))))
PyObject *count_distinct3(PyObject *, PyObject *args) {
    PyObject *a_o, *b_o, *c_o;
    if (!PyArg_ParseTuple(args, "OOO", &a_o, &b_o, &c_o)) return nullptr;
    I64View a, b, c;
    if (!a.get(a_o, "a") || !b.get(b_o, "b") || !c.get(c_o, "c")) return nullptr;
    if (b.n != a.n || c.n != a.n) {
        PyErr_SetString(PyExc_ValueError, "the three arrays must have the same length");
        return nullptr;
    }
    size_t cap = 16;
    while (cap < (size_t)a.n * 2) cap <<= 1;
    int64_t *slot = static_cast<int64_t *>(PyMem_Malloc(cap * sizeof(int64_t)));      // index of the triple stored there, -1 = empty
    if (!slot) return PyErr_NoMemory();
    for (size_t k = 0; k < cap; ++k) slot[k] = -1;
    Py_ssize_t distinct = 0;
    for (Py_ssize_t k = 0; k < a.n; ++k) {
        uint64_t h = (uint64_t)a.p[k] * 0x9E3779B97F4A7C15ull;
        h ^= ((uint64_t)b.p[k] + 0x7F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
        h ^= ((uint64_t)c.p[k] + 0x94D049BBull) * 0x94D049BB133111EBull;
        size_t at = (size_t)(h ^ (h >> 29)) & (cap - 1);
        for (;;) {
            const int64_t j = slot[at];
            if (j < 0) {
                slot[at] = k;
                ++distinct;
                break;
            }
            if (a.p[j] == a.p[k] && b.p[j] == b.p[k] && c.p[j] == c.p[k]) break;
            at = (at + 1) & (cap - 1);
        }
    }
    PyMem_Free(slot);
    return PyLong_FromSsize_t(distinct);
}

// ---- read names kept as ONE byte blob + offsets (coral_amd/names.py: NameTable) ----------------------------------------
struct BlobView {
    Py_buffer blob, off;
    bool ok_blob = false, ok_off = false;
    const char *b = nullptr;
    const int64_t *o = nullptr;
    Py_ssize_t n = 0;          // number of names
    bool get(PyObject *blob_o, PyObject *off_o) {
        if (PyObject_GetBuffer(blob_o, &blob, PyBUF_C_CONTIGUOUS) != 0) return false;
        ok_blob = true;
        if (PyObject_GetBuffer(off_o, &off, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) return false;
        ok_off = true;
        if (off.itemsize != 8 || off.ndim != 1 || off.shape[0] < 1) {
            PyErr_SetString(PyExc_TypeError, "off must be a contiguous one-dimensional int64 array of n + 1 offsets");
            return false;
        }
        b = static_cast<const char *>(blob.buf);
        o = static_cast<const int64_t *>(off.buf);
        n = off.shape[0] - 1;
        if (o[0] < 0 || o[n] > blob.len) {
            PyErr_SetString(PyExc_ValueError, "name offsets exceed the blob");
            return false;
        }
        return true;
    }
    // a new str for name id (UTF-8, which ASCII read names are)
    PyObject *str(int64_t id) const {
        if (id < 0 || id >= n || o[id + 1] < o[id]) {
            PyErr_SetString(PyExc_IndexError, "name id out of range");
            return nullptr;
        }
        return PyUnicode_DecodeUTF8(b + o[id], (Py_ssize_t)(o[id + 1] - o[id]), "strict");
    }
    ~BlobView() {
        if (ok_blob) PyBuffer_Release(&blob);
        if (ok_off) PyBuffer_Release(&off);
    }
};

// blob_names(blob, off, ids | None) -> [str(name ids[k]) ...]   (None: every name, in id order)
PyObject *blob_names(PyObject *, PyObject *args) {
    PyObject *blob_o, *off_o, *ids_o;
    if (!PyArg_ParseTuple(args, "OOO", &blob_o, &off_o, &ids_o)) return nullptr;
    BlobView B;
    if (!B.get(blob_o, off_o)) return nullptr;
    I64View ids;
    const bool all = ids_o == Py_None;
    if (!all && !ids.get(ids_o, "ids")) return nullptr;
    const Py_ssize_t n = all ? B.n : ids.n;
    PyObject *out = PyList_New(n);
    if (!out) return nullptr;
    for (Py_ssize_t k = 0; k < n; ++k) {
        PyObject *s = B.str(all ? (int64_t)k : ids.p[k]);
        if (!s) {
            Py_DECREF(out);
            return nullptr;
        }
        PyList_SET_ITEM(out, k, s);
    }
    return out;
}

// blob_tuples(blob, off, ids, i, j) -> [(name, int(i[k]), int(j[k])) ...]
PyObject *blob_tuples(PyObject *, PyObject *args) {
    PyObject *blob_o, *off_o, *ids_o, *i_o, *j_o;
    if (!PyArg_ParseTuple(args, "OOOOO", &blob_o, &off_o, &ids_o, &i_o, &j_o)) return nullptr;
    BlobView B;
    if (!B.get(blob_o, off_o)) return nullptr;
    I64View ids, vi, vj;
    if (!ids.get(ids_o, "ids") || !vi.get(i_o, "i") || !vj.get(j_o, "j")) return nullptr;
    if (vi.n != ids.n || vj.n != ids.n) {
        PyErr_SetString(PyExc_ValueError, "ids, i and j must have the same length");
        return nullptr;
    }
    PyObject *out = PyList_New(ids.n);
    if (!out) return nullptr;
    for (Py_ssize_t k = 0; k < ids.n; ++k) {
        PyObject *s = B.str(ids.p[k]);
        PyObject *a = s ? PyLong_FromLongLong(vi.p[k]) : nullptr;
        PyObject *b = a ? PyLong_FromLongLong(vj.p[k]) : nullptr;
        PyObject *t = b ? PyTuple_New(3) : nullptr;
        if (!t) {
            Py_XDECREF(s);
            Py_XDECREF(a);
            Py_XDECREF(b);
            Py_DECREF(out);
            return nullptr;
        }
        PyTuple_SET_ITEM(t, 0, s);
        PyTuple_SET_ITEM(t, 1, a);
        PyTuple_SET_ITEM(t, 2, b);
        PyList_SET_ITEM(out, k, t);
    }
    return out;
}

// blob_hashes(blob, off, ids, out) : out[k] = hash(str(name ids[k])) — THIS interpreter's str hash (its algorithm and its
// per-process key), which is what orders the reference's sets of read names (ibg:379-384, :412-418).  An ASCII str hashes
// its bytes (_Py_HashBytes over the one-byte-per-character buffer), so neither the str nor a Python-level loop is needed;
// a name with a byte >= 0x80 goes through a real str.
PyObject *blob_hashes(PyObject *, PyObject *args) {
    PyObject *blob_o, *off_o, *ids_o, *out_o;
    if (!PyArg_ParseTuple(args, "OOOO", &blob_o, &off_o, &ids_o, &out_o)) return nullptr;
    BlobView B;
    if (!B.get(blob_o, off_o)) return nullptr;
    I64View ids;
    if (!ids.get(ids_o, "ids")) return nullptr;
    Py_buffer ob;
    if (PyObject_GetBuffer(out_o, &ob, PyBUF_C_CONTIGUOUS | PyBUF_WRITABLE | PyBUF_FORMAT) != 0) return nullptr;
    if (ob.itemsize != 8 || ob.ndim != 1 || ob.shape[0] != ids.n) {
        PyBuffer_Release(&ob);
        PyErr_SetString(PyExc_TypeError, "out must be a writable int64 array as long as ids");
        return nullptr;
    }
    int64_t *out = static_cast<int64_t *>(ob.buf);
    // ASCII names (all of them, in practice) are hashed from their bytes WITHOUT the interpreter lock: _Py_HashBytes touches no
    // Python object, so this loop can run beside the build's main thread; the rare others are done afterwards through a str
    Py_ssize_t bad = -1, n_other = 0;
    Py_BEGIN_ALLOW_THREADS
    for (Py_ssize_t k = 0; k < ids.n; ++k) {
        const int64_t id = ids.p[k];
        if (id < 0 || id >= B.n || B.o[id + 1] < B.o[id]) {
            bad = k;
            break;
        }
        const char *s = B.b + B.o[id];
        const Py_ssize_t len = (Py_ssize_t)(B.o[id + 1] - B.o[id]);
        bool ascii = true;
        for (Py_ssize_t q = 0; q < len; ++q) ascii &= ((unsigned char)s[q] < 0x80);
        if (ascii) {
            out[k] = (int64_t)_Py_HashBytes(s, len);
        } else {
            out[k] = 0;
            ++n_other;
        }
    }
    Py_END_ALLOW_THREADS
    if (bad >= 0) {
        PyBuffer_Release(&ob);
        PyErr_SetString(PyExc_IndexError, "name id out of range");
        return nullptr;
    }
    for (Py_ssize_t k = 0; n_other > 0 && k < ids.n; ++k) {
        const int64_t id = ids.p[k];
        const char *s = B.b + B.o[id];
        const Py_ssize_t len = (Py_ssize_t)(B.o[id + 1] - B.o[id]);
        bool ascii = true;
        for (Py_ssize_t q = 0; q < len; ++q) ascii &= ((unsigned char)s[q] < 0x80);
        if (ascii) continue;
        PyObject *u = B.str(id);
        if (!u) {
            PyBuffer_Release(&ob);
            return nullptr;
        }
        out[k] = (int64_t)PyObject_Hash(u);
        Py_DECREF(u);
        --n_other;
    }
    PyBuffer_Release(&ob);
    Py_RETURN_NONE;
}

PyMethodDef methods[] = {
    {"names_of", names_of, METH_VARARGS, "names_of(names, ids) -> [names[k] for k in ids]  (ids: contiguous int64 array)"},
    {"read_tuples", read_tuples, METH_VARARGS, "read_tuples(names, ids, i, j) -> [(names[ids[k]], i[k], j[k]), ...]"},
    {"count_distinct3", count_distinct3, METH_VARARGS, "count_distinct3(a, b, c) -> len(set(zip(a, b, c)))  (contiguous int64 arrays)"},
    {"blob_names", blob_names, METH_VARARGS, "blob_names(blob, off, ids | None) -> list of str"},
    {"blob_tuples", blob_tuples, METH_VARARGS, "blob_tuples(blob, off, ids, i, j) -> [(name, i[k], j[k]), ...]"},
    {"blob_hashes", blob_hashes, METH_VARARGS, "blob_hashes(blob, off, ids, out): out[k] = hash(name ids[k]) of this interpreter"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pyobjects", "Python containers of the graph build, built with the C API.", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};
}  // namespace

PyMODINIT_FUNC PyInit__pyobjects(void) { return PyModule_Create(&module); }
