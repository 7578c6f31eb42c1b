/* _pyhelp.c -- loops over Python objects that cost tens of milliseconds per call in pure Python at 50 000 x 2.5 kb: the
 * addresses / lengths of a list of ASCII str objects (so that the C ABI's isocon_store_create_ptrs gathers them straight into its
 * pinned staging buffer: no 125 MB "".join), and the inverse, a list of str cut out of one ASCII buffer (the gapped alignments
 * isocon_sg_strings_batch returns).  Host glue of the Python wrappers only; the C ABI itself (include/isocon_hip.h) knows nothing
 * about Python objects.  Built by isocon_amd/_lib.py:build() with the system compiler; if it is missing the wrappers use their
 * pure-Python loops. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

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

/* split_ascii(buf_addr: int, ptr_addr: int, n: int) -> [str(buf[ptr[i]:ptr[i+1]]) for i < n]; ptr = int64[n + 1].
 * The str objects are created first; their bytes are then copied by up to four threads (no Python API inside them): at 50 000 x 2.9 kb
 * the copy and the first touch of 146 MB of fresh heap are what the call costs. */
typedef struct { char **dst; const char *buf; const int64_t *ptr; Py_ssize_t i0, i1; } split_job;

static void *split_worker(void *arg)
{
    const split_job *j = (const split_job *)arg;
    for (Py_ssize_t i = j->i0; i < j->i1; ++i) {
        const int64_t len = j->ptr[i + 1] - j->ptr[i];
        if (len > 0) memcpy(j->dst[i], j->buf + j->ptr[i], (size_t)len);
    }
    return NULL;
}

static PyObject *split_ascii(PyObject *self, PyObject *args)
{
    unsigned long long ba, pa;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "KKn", &ba, &pa, &n)) return NULL;
    const char *buf = (const char *)(uintptr_t)ba;
    const int64_t *ptr = (const int64_t *)(uintptr_t)pa;
    PyObject *out = PyList_New(n);
    if (!out) return NULL;
    char **dst = (char **)malloc((size_t)(n > 0 ? n : 1) * sizeof(char *));
    if (!dst) { Py_DECREF(out); return PyErr_NoMemory(); }
    for (Py_ssize_t i = 0; i < n; ++i) {
        const int64_t len = ptr[i + 1] - ptr[i];
        PyObject *s = len >= 0 ? PyUnicode_New((Py_ssize_t)len, 127) : NULL;
        if (!s) { Py_DECREF(out); free(dst); if (len < 0) PyErr_SetString(PyExc_ValueError, "split_ascii: descending offsets"); return NULL; }
        dst[i] = (char *)PyUnicode_1BYTE_DATA(s);
        PyList_SET_ITEM(out, i, s);
    }
    const int64_t total = n > 0 ? ptr[n] - ptr[0] : 0;
    const int n_thr = total >= ((int64_t)8 << 20) ? 4 : 1;
    split_job jobs[4];
    for (int t = 0; t < n_thr; ++t) {
        jobs[t].dst = dst; jobs[t].buf = buf; jobs[t].ptr = ptr;
        jobs[t].i0 = n * t / n_thr; jobs[t].i1 = n * (t + 1) / n_thr;
    }
    if (n_thr == 1) split_worker(&jobs[0]);
    else {
        pthread_t th[4];
        int started[4] = {0, 0, 0, 0};
        Py_BEGIN_ALLOW_THREADS
        for (int t = 0; t < n_thr; ++t) started[t] = pthread_create(&th[t], NULL, split_worker, &jobs[t]) == 0;
        for (int t = 0; t < n_thr; ++t) {
            if (started[t]) pthread_join(th[t], NULL);
            else split_worker(&jobs[t]);
        }
        Py_END_ALLOW_THREADS
    }
    free(dst);
    return out;
}

/* csr_to_dict(keys: list, is_query_addr: int (uint8[n] or 0 = all), best_addr: int (int32[n]), row_ptr_addr: int (int64[n + 1]),
 *             cols_addr: int (uint32[]), n: int) -> {keys[i]: {keys[c]: best[i] for c in row i}} for the query entries, in entry order and,
 * inside a row, in column order (the insertion order the reference's loop produces: nearest_neighbor_graph.py:145-178) */
static PyObject *csr_to_dict(PyObject *self, PyObject *args)
{
    PyObject *keys;
    unsigned long long qa, ba, ra, ca;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "OKKKKn", &keys, &qa, &ba, &ra, &ca, &n)) return NULL;
    if (!PyList_Check(keys) || PyList_GET_SIZE(keys) < n) { PyErr_SetString(PyExc_TypeError, "csr_to_dict: a list of n keys is required"); return NULL; }
    const uint8_t *isq = (const uint8_t *)(uintptr_t)qa;
    const int32_t *best = (const int32_t *)(uintptr_t)ba;
    const int64_t *row_ptr = (const int64_t *)(uintptr_t)ra;
    const uint32_t *cols = (const uint32_t *)(uintptr_t)ca;
    PyObject *out = PyDict_New();
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (isq && !isq[i]) continue;
        PyObject *row = PyDict_New(), *val = NULL;
        if (!row) { Py_DECREF(out); return NULL; }
        if (row_ptr[i + 1] > row_ptr[i]) {
            val = PyLong_FromLong((long)best[i]);
            if (!val) { Py_DECREF(row); Py_DECREF(out); return NULL; }
        }
        for (int64_t e = row_ptr[i]; e < row_ptr[i + 1]; ++e) {
            if ((Py_ssize_t)cols[e] >= n || PyDict_SetItem(row, PyList_GET_ITEM(keys, (Py_ssize_t)cols[e]), val) < 0) {
                if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "csr_to_dict: column out of range");
                Py_XDECREF(val); Py_DECREF(row); Py_DECREF(out);
                return NULL;
            }
        }
        Py_XDECREF(val);
        const int rc = PyDict_SetItem(out, PyList_GET_ITEM(keys, i), row);
        Py_DECREF(row);
        if (rc < 0) { Py_DECREF(out); return NULL; }
    }
    return out;
}

/* pair_ids(index: dict[str, int], pairs: list[tuple], a_addr: int, b_addr: int) -> number of pairs looked up (len(pairs)), or -1 - i
 * when a member of pair i is not a key: a[i], b[i] = index[pairs[i][0]], index[pairs[i][1]] as uint32 */
static PyObject *pair_ids(PyObject *self, PyObject *args)
{
    PyObject *index, *pairs;
    unsigned long long aa, ba;
    if (!PyArg_ParseTuple(args, "OOKK", &index, &pairs, &aa, &ba)) return NULL;
    if (!PyDict_Check(index) || !PyList_Check(pairs)) { PyErr_SetString(PyExc_TypeError, "pair_ids: a dict and a list are required"); return NULL; }
    uint32_t *a = (uint32_t *)(uintptr_t)aa, *b = (uint32_t *)(uintptr_t)ba;
    const Py_ssize_t n = PyList_GET_SIZE(pairs);
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *p = PyList_GET_ITEM(pairs, i);
        if (!PyTuple_Check(p) || PyTuple_GET_SIZE(p) < 2) { PyErr_Format(PyExc_TypeError, "pair %zd is not a 2-tuple", i); return NULL; }
        PyObject *va = PyDict_GetItemWithError(index, PyTuple_GET_ITEM(p, 0));
        PyObject *vb = va ? PyDict_GetItemWithError(index, PyTuple_GET_ITEM(p, 1)) : NULL;
        if (!va || !vb) {
            if (PyErr_Occurred()) return NULL;
            return PyLong_FromSsize_t(-1 - i);
        }
        const unsigned long xa = PyLong_AsUnsignedLong(va), xb = PyLong_AsUnsignedLong(vb);
        if ((xa == (unsigned long)-1 || xb == (unsigned long)-1) && PyErr_Occurred()) return NULL;
        a[i] = (uint32_t)xa; b[i] = (uint32_t)xb;
    }
    return PyLong_FromSsize_t(n);
}

/* rank_strings(seqs: list[str] (ASCII), out_addr: int (uint32[len(seqs)])) -> None: out[i] = position of seqs[i] in sorted(seqs)
 * (Python's str order: bytewise, a proper prefix first; equal strings in index order, like the stable sorted()).  The partition of the
 * nearest-neighbour graph breaks its ties with the reference's `m < centre` on the sequences (partitions.py:346-361): 50 000 x 2.5 kb
 * strings sorted here in ~10 ms instead of 80 ms of key-function calls. */
typedef struct { const char *p; const char *key; Py_ssize_t len; uint32_t idx; } rank_item;

/* The strings sit all over the heap: a comparison of two of them is two cache misses.  Their first RANK_KEY bytes are copied into one
 * contiguous block first (in the threads), most comparisons are decided there (reads of one isoform differ at their first error). */
#define RANK_KEY 128

static int rank_cmp(const void *a, const void *b)
{
    const rank_item *x = (const rank_item *)a, *y = (const rank_item *)b;
    const Py_ssize_t m = x->len < y->len ? x->len : y->len;
    if (x->key && y->key) {
        const Py_ssize_t mk = m < RANK_KEY ? m : RANK_KEY;
        const int ck = mk ? memcmp(x->key, y->key, (size_t)mk) : 0;
        if (ck) return ck;
        if (m <= RANK_KEY) {
            if (x->len != y->len) return x->len < y->len ? -1 : 1;
            return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
        }
    }
    const int c = m ? memcmp(x->p, y->p, (size_t)m) : 0;
    if (c) return c;
    if (x->len != y->len) return x->len < y->len ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

typedef struct { rank_item *items; size_t n; } rank_job;

static void *rank_sort_worker(void *arg)
{
    const rank_job *j = (const rank_job *)arg;
    for (size_t i = 0; i < j->n; ++i)          /* (key points into the block; the bytes behind a short string stay unread) */
        if (j->items[i].key) memcpy((char *)j->items[i].key, j->items[i].p, (size_t)(j->items[i].len < RANK_KEY ? j->items[i].len : RANK_KEY));
    qsort(j->items, j->n, sizeof(rank_item), rank_cmp);
    return NULL;
}

static void rank_merge(const rank_item *a, size_t na, const rank_item *b, size_t nb, rank_item *out)
{
    size_t i = 0, j = 0, k = 0;
    while (i < na && j < nb) out[k++] = rank_cmp(&b[j], &a[i]) < 0 ? b[j++] : a[i++];
    while (i < na) out[k++] = a[i++];
    while (j < nb) out[k++] = b[j++];
}

static PyObject *rank_strings(PyObject *self, PyObject *args)
{
    PyObject *seqs;
    unsigned long long oa;
    if (!PyArg_ParseTuple(args, "OK", &seqs, &oa)) return NULL;
    if (!PyList_Check(seqs)) { PyErr_SetString(PyExc_TypeError, "rank_strings: a list is required"); return NULL; }
    const Py_ssize_t n = PyList_GET_SIZE(seqs);
    uint32_t *out = (uint32_t *)(uintptr_t)oa;
    rank_item *items = (rank_item *)malloc((size_t)(n > 0 ? n : 1) * sizeof(rank_item));
    if (!items) return PyErr_NoMemory();
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *s = PyList_GET_ITEM(seqs, i);
        if (!PyUnicode_Check(s) || PyUnicode_READY(s) < 0 || !PyUnicode_IS_ASCII(s)) {
            free(items);
            if (!PyErr_Occurred()) PyErr_Format(PyExc_TypeError, "rank_strings: element %zd is not an ASCII str", i);
            return NULL;
        }
        items[i].p = (const char *)PyUnicode_1BYTE_DATA(s);
        items[i].len = PyUnicode_GET_LENGTH(s);
        items[i].idx = (uint32_t)i;
        items[i].key = NULL;
    }
    rank_item *tmp = NULL;
    char *keys = NULL;
    Py_BEGIN_ALLOW_THREADS
    const int n_thr = n >= 8192 ? 4 : 1;          /* four sorted runs in four threads, then two rounds of merging */
    if (n_thr > 1) {
        tmp = (rank_item *)malloc((size_t)n * sizeof(rank_item));
        keys = (char *)malloc((size_t)n * RANK_KEY);
        if (tmp && keys) for (Py_ssize_t i = 0; i < n; ++i) items[i].key = keys + (size_t)i * RANK_KEY;          /* (filled by the workers) */
    }
    if (n_thr == 1 || !tmp || !keys) qsort(items, (size_t)n, sizeof(rank_item), rank_cmp);
    else {
        rank_job jobs[4];
        pthread_t th[4];
        int started[4];
        Py_ssize_t cut[5];
        for (int t = 0; t <= 4; ++t) cut[t] = n * t / 4;
        for (int t = 0; t < 4; ++t) {
            jobs[t].items = items + cut[t]; jobs[t].n = (size_t)(cut[t + 1] - cut[t]);
            started[t] = pthread_create(&th[t], NULL, rank_sort_worker, &jobs[t]) == 0;
        }
        for (int t = 0; t < 4; ++t) {
            if (started[t]) pthread_join(th[t], NULL);
            else rank_sort_worker(&jobs[t]);
        }
        rank_merge(items + cut[0], (size_t)(cut[1] - cut[0]), items + cut[1], (size_t)(cut[2] - cut[1]), tmp + cut[0]);
        rank_merge(items + cut[2], (size_t)(cut[3] - cut[2]), items + cut[3], (size_t)(cut[4] - cut[3]), tmp + cut[2]);
        rank_merge(tmp + cut[0], (size_t)(cut[2] - cut[0]), tmp + cut[2], (size_t)(cut[4] - cut[2]), items);
    }
    for (Py_ssize_t r = 0; r < n; ++r) out[items[r].idx] = (uint32_t)r;
    Py_END_ALLOW_THREADS
    free(tmp);
    free(keys);
    free(items);
    Py_RETURN_NONE;
}

/* group_keys_by_value(d: dict) -> {value: [keys with that value, in d's order]}, values in first-appearance order: what
 * isocon_get_candidates.get_unique_seq_accessions builds with a setdefault loop over 50 000 reads in every correction step
 * (modules/isocon_get_candidates.py:22-35: {sequence: [accessions]}). */
static PyObject *group_keys_by_value(PyObject *self, PyObject *args)
{
    PyObject *d;
    if (!PyArg_ParseTuple(args, "O", &d)) return NULL;
    if (!PyDict_Check(d)) { PyErr_SetString(PyExc_TypeError, "group_keys_by_value: a dict is required"); return NULL; }
    PyObject *out = PyDict_New();
    if (!out) return NULL;
    Py_ssize_t pos = 0;
    PyObject *key, *value;
    while (PyDict_Next(d, &pos, &key, &value)) {
        PyObject *lst = PyDict_GetItemWithError(out, value);          /* borrowed */
        if (!lst) {
            if (PyErr_Occurred()) { Py_DECREF(out); return NULL; }
            lst = PyList_New(1);
            if (!lst) { Py_DECREF(out); return NULL; }
            Py_INCREF(key);
            PyList_SET_ITEM(lst, 0, key);
            const int rc = PyDict_SetItem(out, value, lst);
            Py_DECREF(lst);
            if (rc < 0) { Py_DECREF(out); return NULL; }
        } else if (PyList_Append(lst, key) < 0) { Py_DECREF(out); return NULL; }
    }
    return out;
}

static PyMethodDef methods[] = {
    {"group_keys_by_value", group_keys_by_value, METH_VARARGS, "{value: [keys]} of a dict, in insertion order"},
    {"rank_strings", rank_strings, METH_VARARGS, "rank of every string of a list in the sorted order of the list"},
    {"pair_ids", pair_ids, METH_VARARGS, "ids of the members of a list of pairs"},
    {"csr_to_dict", csr_to_dict, METH_VARARGS, "dict of dicts from the CSR arrays of a nearest-neighbour graph"},
    {"str_pointers", str_pointers, METH_VARARGS, "addresses and lengths of a list of ASCII str"},
    {"split_ascii", split_ascii, METH_VARARGS, "list of str cut out of an ASCII buffer"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyhelp", NULL, -1, methods};
PyMODINIT_FUNC PyInit__pyhelp(void) { return PyModule_Create(&moddef); }
