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

PyMethodDef methods[] = {
    {"names_of", names_of, METH_VARARGS, "names_of(names, ids) -> [names[k] for k in ids]  (ids: contiguous int64 array)"},
    {"read_tuples", read_tuples, METH_VARARGS, "read_tuples(names, ids, i, j) -> [(names[ids[k]], i[k], j[k]), ...]"},
    {"count_distinct3", count_distinct3, METH_VARARGS, "count_distinct3(a, b, c) -> len(set(zip(a, b, c)))  (contiguous int64 arrays)"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pyobjects", "Python containers of the graph build, built with the C API.", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};
}  // namespace

PyMODINIT_FUNC PyInit__pyobjects(void) { return PyModule_Create(&module); }
