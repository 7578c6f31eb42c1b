/* _pyhelp.c -- two loops over Python strings that cost tens of milliseconds per call in pure Python at 50 000 x 2.5 kb: the
 * addresses / lengths of a list of ASCII str objects (so that the C ABI's isocon_store_create_ptrs gathers them straight into its
 * pinned staging buffer: no 125 MB "".join), and the inverse, a list of str cut out of one ASCII buffer (the gapped alignments
 * isocon_sg_strings_batch returns).  Host glue of the Python wrappers only; the C ABI itself (include/isocon_hip.h) knows nothing
 * about Python objects.  Built by isocon_amd/_lib.py:build() with the system compiler; if it is missing the wrappers use their
 * pure-Python loops. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* str_pointers(seqs: list[str], ptrs_addr: int, lens_addr: int) -> total bytes; arrays of len(seqs) uint64 each */
static PyObject *str_pointers(PyObject *self, PyObject *args)
{
    PyObject *seqs;
    unsigned long long pa, la;
    if (!PyArg_ParseTuple(args, "OKK", &seqs, &pa, &la)) return NULL;
    if (!PyList_Check(seqs)) { PyErr_SetString(PyExc_TypeError, "str_pointers: a list is required"); return NULL; }
    uint64_t *ptrs = (uint64_t *)(uintptr_t)pa, *lens = (uint64_t *)(uintptr_t)la;
    const Py_ssize_t n = PyList_GET_SIZE(seqs);
    unsigned long long total = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *s = PyList_GET_ITEM(seqs, i);
        if (!PyUnicode_Check(s)) { PyErr_Format(PyExc_TypeError, "sequence %zd is not a str", i); return NULL; }
        if (PyUnicode_READY(s) < 0) return NULL;
        if (!PyUnicode_IS_ASCII(s)) { PyErr_Format(PyExc_ValueError, "sequence %zd contains a symbol outside ACGT (non-ASCII character)", i); return NULL; }
        ptrs[i] = (uint64_t)(uintptr_t)PyUnicode_1BYTE_DATA(s);
        lens[i] = (uint64_t)PyUnicode_GET_LENGTH(s);
        total += lens[i];
    }
    return PyLong_FromUnsignedLongLong(total);
}

/* split_ascii(buf_addr: int, ptr_addr: int, n: int) -> [str(buf[ptr[i]:ptr[i+1]]) for i < n]; ptr = int64[n + 1] */
static PyObject *split_ascii(PyObject *self, PyObject *args)
{
    unsigned long long ba, pa;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "KKn", &ba, &pa, &n)) return NULL;
    const char *buf = (const char *)(uintptr_t)ba;
    const int64_t *ptr = (const int64_t *)(uintptr_t)pa;
    PyObject *out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        const int64_t len = ptr[i + 1] - ptr[i];
        PyObject *s = len >= 0 ? PyUnicode_New((Py_ssize_t)len, 127) : NULL;
        if (!s) { Py_DECREF(out); if (len < 0) PyErr_SetString(PyExc_ValueError, "split_ascii: descending offsets"); return NULL; }
        if (len) memcpy(PyUnicode_1BYTE_DATA(s), buf + ptr[i], (size_t)len);
        PyList_SET_ITEM(out, i, s);
    }
    return out;
}

static PyMethodDef methods[] = {
    {"str_pointers", str_pointers, METH_VARARGS, "addresses and lengths of a list of ASCII str"},
    {"split_ascii", split_ascii, METH_VARARGS, "list of str cut out of an ASCII buffer"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyhelp", NULL, -1, methods};
PyMODINIT_FUNC PyInit__pyhelp(void) { return PyModule_Create(&moddef); }
