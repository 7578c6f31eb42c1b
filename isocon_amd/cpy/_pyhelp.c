/* _pyhelp.c -- loops over Python objects that cost tens of milliseconds per call in pure Python at 50 000 x 2.5 kb: the
 * addresses / lengths of a list of ASCII str objects (so that the C ABI's isocon_store_create_ptrs gathers them straight into its
 * pinned staging buffer: no 125 MB "".join), and the inverse, a list of str cut out of one ASCII buffer (the gapped alignments
 * isocon_sg_strings_batch returns).  Host glue of the Python wrappers only; the C ABI itself (include/isocon_hip.h) knows nothing
 * about Python objects.  Built by isocon_amd/_lib.py:build() with the system compiler; if it is missing the wrappers use their
 * pure-Python loops. */
#define PY_SSIZE_T_CLEAN
#ifndef _GNU_SOURCE
#define _GNU_SOURCE          /* memmem */
#endif
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

/* split_ascii_rows(buf_addr: int, ptr_addr: int, rows_addr: int, k: int) -> [str(buf[ptr[r]:ptr[r + 1]]) for r in rows[0 .. k)], rows = int64[k],
 * with EQUAL rows sharing ONE str object.  The corrected reads of an iteration converge on their consensus: 50 000 rows are a few thousand
 * distinct sequences, and every dict the pipeline then builds over them (accessions per sequence, multiplicities, the NN graph's keys)
 * hashes a str object once and compares by identity first -- 125 MB of hashing and as much of memcmp per pass over 50 000 separate objects.
 * Row hashes are computed by up to four threads (no Python API inside them); the table walk and the str objects are single-threaded. */
typedef struct { const char *buf; const int64_t *ptr; const int64_t *rows; uint64_t *hash; Py_ssize_t i0, i1; } rowhash_job;

static uint64_t row_hash64(const char *p, size_t len)
{
    uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t)len;
    while (len >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
        p += 8; len -= 8;
    }
    uint64_t w = 0;
    if (len) memcpy(&w, p, len);
    h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
    return h ^ (h >> 29);
}

static void *rowhash_worker(void *arg)
{
    const rowhash_job *j = (const rowhash_job *)arg;
    for (Py_ssize_t i = j->i0; i < j->i1; ++i) {
        const int64_t r = j->rows[i], len = j->ptr[r + 1] - j->ptr[r];
        j->hash[i] = len > 0 ? row_hash64(j->buf + j->ptr[r], (size_t)len) : 0;
    }
    return NULL;
}

static PyObject *split_ascii_rows(PyObject *self, PyObject *args)
{
    unsigned long long ba, pa, ra;
    Py_ssize_t k, n_rows;
    if (!PyArg_ParseTuple(args, "KKnKn", &ba, &pa, &n_rows, &ra, &k)) return NULL;
    const char *buf = (const char *)(uintptr_t)ba;
    const int64_t *ptr = (const int64_t *)(uintptr_t)pa;          /* n_rows + 1 offsets */
    const int64_t *rows = (const int64_t *)(uintptr_t)ra;
    if (k < 0 || n_rows < 0) { PyErr_SetString(PyExc_ValueError, "split_ascii_rows: negative count"); return NULL; }
    int64_t total = 0;
    for (Py_ssize_t i = 0; i < k; ++i) {
        const int64_t len = (rows[i] < 0 || rows[i] >= (int64_t)n_rows) ? -1 : ptr[rows[i] + 1] - ptr[rows[i]];
        if (len < 0) { PyErr_SetString(PyExc_ValueError, "split_ascii_rows: bad row or descending offsets"); return NULL; }
        total += len;
    }
    size_t cap = 16;
    while (cap < (size_t)k * 2) cap <<= 1;
    uint64_t *hash = (uint64_t *)malloc((size_t)(k > 0 ? k : 1) * sizeof(uint64_t));
    Py_ssize_t *table = (Py_ssize_t *)malloc(cap * sizeof(Py_ssize_t));          /* slot -> index of the first row with that content, -1 = free */
    PyObject *out = PyList_New(k);
    if (!hash || !table || !out) { free(hash); free(table); Py_XDECREF(out); return PyErr_NoMemory(); }
    const int n_thr = total >= ((int64_t)8 << 20) ? 4 : 1;
    rowhash_job jobs[4];
    for (int t = 0; t < n_thr; ++t) {
        jobs[t].buf = buf; jobs[t].ptr = ptr; jobs[t].rows = rows; jobs[t].hash = hash;
        jobs[t].i0 = k * t / n_thr; jobs[t].i1 = k * (t + 1) / n_thr;
    }
    if (n_thr == 1) rowhash_worker(&jobs[0]);
    else {
        pthread_t th[4];
        int started[4] = {0, 0, 0, 0};
        Py_BEGIN_ALLOW_THREADS
        for (int t = 0; t < n_thr; ++t) started[t] = pthread_create(&th[t], NULL, rowhash_worker, &jobs[t]) == 0;
        for (int t = 0; t < n_thr; ++t) {
            if (started[t]) pthread_join(th[t], NULL);
            else rowhash_worker(&jobs[t]);
        }
        Py_END_ALLOW_THREADS
    }
    for (size_t x = 0; x < cap; ++x) table[x] = -1;
    for (Py_ssize_t i = 0; i < k; ++i) {
        const int64_t r = rows[i], len = ptr[r + 1] - ptr[r];
        size_t slot = (size_t)hash[i] & (cap - 1);
        PyObject *s = NULL;
        for (;; slot = (slot + 1) & (cap - 1)) {
            const Py_ssize_t f = table[slot];
            if (f < 0) break;
            const int64_t rf = rows[f];
            if (hash[f] == hash[i] && ptr[rf + 1] - ptr[rf] == len && (len == 0 || memcmp(buf + ptr[rf], buf + ptr[r], (size_t)len) == 0)) {
                s = PyList_GET_ITEM(out, f);
                Py_INCREF(s);
                break;
            }
        }
        if (!s) {
            s = PyUnicode_New((Py_ssize_t)len, 127);
            if (!s) { free(hash); free(table); Py_DECREF(out); return NULL; }
            if (len > 0) memcpy(PyUnicode_1BYTE_DATA(s), buf + ptr[r], (size_t)len);
            table[slot] = i;
        }
        PyList_SET_ITEM(out, i, s);
    }
    free(hash); free(table);
    return out;
}

/* invariant_partners(seq1: str, seqs: list[str], thr: int) -> [i for i, seq2 in enumerate(seqs) if _pair_is_invariant(seq1, seq2, thr)]
 * -- end_invariant_functions._pair_is_invariant / is_overlap (the reference's end_invariant_functions.py:933-946, :884-918) on ASCII strings:
 *   seq2 inside seq1: its FIRST occurrence decides (at most thr characters of seq1 before and after it), nothing else is tried;
 *   otherwise a suffix of one equal to a prefix of the other (after cutting the longer to the length of the shorter, as the reference does)
 *   that leaves at most thr characters of either original string uncovered -- in either direction.
 * One call per candidate against its partner list: the Python loop made 3.6 million calls of three functions for 9 000 candidates. */
static int overlap_holds(const char *t1, Py_ssize_t l1, const char *t2, Py_ssize_t l2, Py_ssize_t thr)
{
    if (l1 == 0 || l2 == 0) return 0;
    const char *a = l1 > l2 ? t1 + (l1 - l2) : t1;          /* text1[-len2:] */
    const Py_ssize_t n = l1 < l2 ? l1 : l2;                 /* (text2[:len1]: the first n characters of t2) */
    if (memcmp(a, t2, (size_t)n) == 0) return 1;
    const Py_ssize_t need = (l1 > l2 ? l1 : l2) - thr;
    if (need <= 0) return 1;
    for (Py_ssize_t L = n; L >= need; --L)
        if (memcmp(a + (n - L), t2, (size_t)L) == 0) return 1;
    return 0;
}

static PyObject *invariant_partners(PyObject *self, PyObject *args)
{
    PyObject *seq1, *seqs;
    Py_ssize_t thr;
    if (!PyArg_ParseTuple(args, "UO!n", &seq1, &PyList_Type, &seqs, &thr)) return NULL;
    if (PyUnicode_READY(seq1) < 0) return NULL;
    if (PyUnicode_KIND(seq1) != PyUnicode_1BYTE_KIND) { PyErr_SetString(PyExc_TypeError, "invariant_partners: 1-byte strings only"); return NULL; }
    const char *s1 = (const char *)PyUnicode_1BYTE_DATA(seq1);
    const Py_ssize_t l1 = PyUnicode_GET_LENGTH(seq1), n = PyList_GET_SIZE(seqs);
    PyObject *out = PyList_New(0);
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *o = PyList_GET_ITEM(seqs, i);
        if (!PyUnicode_Check(o) || PyUnicode_READY(o) < 0 || PyUnicode_KIND(o) != PyUnicode_1BYTE_KIND) {
            Py_DECREF(out);
            PyErr_SetString(PyExc_TypeError, "invariant_partners: a list of 1-byte strings is required");
            return NULL;
        }
        const char *s2 = (const char *)PyUnicode_1BYTE_DATA(o);
        const Py_ssize_t l2 = PyUnicode_GET_LENGTH(o);
        int holds;
        /* first occurrence of seq2 in seq1: the caller's pairs differ by at most 2 thr in length -- a memcmp per offset (each fails within a
         * few bytes) beats memmem, whose preprocessing of a 3 kb needle was the whole cost of the call; memmem for the general case */
        const char *hit = NULL;
        if (l2 == 0) hit = s1;
        else if (l2 <= l1 && l1 - l2 <= 64) {
            for (Py_ssize_t off = 0; off + l2 <= l1; ++off)
                if (s1[off] == s2[0] && memcmp(s1 + off, s2, (size_t)l2) == 0) { hit = s1 + off; break; }
        } else if (l2 <= l1) hit = (const char *)memmem(s1, (size_t)l1, s2, (size_t)l2);
        if (hit) {
            const Py_ssize_t start = (Py_ssize_t)(hit - s1), end = l1 - (start + l2);
            holds = start <= thr && end <= thr;
        } else holds = overlap_holds(s1, l1, s2, l2, thr) || overlap_holds(s2, l2, s1, l1, thr);
        if (holds) {
            PyObject *idx = PyLong_FromSsize_t(i);
            if (!idx || PyList_Append(out, idx) < 0) { Py_XDECREF(idx); Py_DECREF(out); return NULL; }
            Py_DECREF(idx);
        }
    }
    return out;
}

/* best_solution(max_insertion: str, q_ins: str) -> bytes(len(max_insertion)): functions.get_best_solution (the reference's functions.py:635-676
 * with min_ed :771-799 and the unit-cost global alignment that stands where it calls edlib, functions.nw_path_cigar) on ASCII strings:
 *   "-" -> all gaps; q_ins found in max_insertion -> at its first occurrence; else threaded along ONE optimal global alignment that deletes
 *   nothing from q_ins (backtracking from the end prefers a step in max_insertion alone, then one in q_ins alone, then the diagonal);
 *   else at the offset with the most matching characters (first maximum; offset 0: q_ins left-aligned, cut to the width).
 * The placement of every distinct insertion of a wide slot: 177 000 calls of the Python dynamic programme on 50 000 ONT-profile reads. */
#define BS_MAX 255
static PyObject *best_solution(PyObject *self, PyObject *args)
{
    PyObject *mxo, *qo;
    if (!PyArg_ParseTuple(args, "UU", &mxo, &qo)) return NULL;
    if (PyUnicode_READY(mxo) < 0 || PyUnicode_READY(qo) < 0) return NULL;
    if (PyUnicode_KIND(mxo) != PyUnicode_1BYTE_KIND || PyUnicode_KIND(qo) != PyUnicode_1BYTE_KIND) {
        PyErr_SetString(PyExc_TypeError, "best_solution: 1-byte strings only");
        return NULL;
    }
    const char *mx = (const char *)PyUnicode_1BYTE_DATA(mxo), *q = (const char *)PyUnicode_1BYTE_DATA(qo);
    const Py_ssize_t L = PyUnicode_GET_LENGTH(mxo), m = PyUnicode_GET_LENGTH(qo);
    if (L > BS_MAX || m > BS_MAX) { PyErr_SetString(PyExc_ValueError, "best_solution: strings of at most 255 characters"); return NULL; }
    PyObject *out = PyBytes_FromStringAndSize(NULL, L);
    if (!out) return NULL;
    char *o = PyBytes_AS_STRING(out);
    memset(o, '-', (size_t)L);
    if (m == 1 && q[0] == '-') return out;
    /* first occurrence of q_ins in max_insertion ("" is found at 0) */
    for (Py_ssize_t p = 0; p + m <= L; ++p)
        if (memcmp(mx + p, q, (size_t)m) == 0) { memcpy(o + p, q, (size_t)m); return out; }
    /* min_ed: global unit-cost alignment of max_insertion (rows) against q_ins (columns) */
    if (L > 0) {
        static _Thread_local uint16_t D[BS_MAX + 1][BS_MAX + 1];
        for (Py_ssize_t i = 0; i <= L; ++i) D[i][0] = (uint16_t)i;
        for (Py_ssize_t j = 0; j <= m; ++j) D[0][j] = (uint16_t)j;
        for (Py_ssize_t i = 1; i <= L; ++i)
            for (Py_ssize_t j = 1; j <= m; ++j) {
                uint16_t best = (uint16_t)(D[i - 1][j - 1] + (mx[i - 1] != q[j - 1]));
                if (D[i - 1][j] + 1 < best) best = (uint16_t)(D[i - 1][j] + 1);
                if (D[i][j - 1] + 1 < best) best = (uint16_t)(D[i][j - 1] + 1);
                D[i][j] = best;
            }
        /* backtrack; the threaded string is written from its end: an 'I' (row only) is a gap, anything else takes the next character of q_ins */
        Py_ssize_t i = L, j = m;
        int deleted = 0;
        while (i > 0 || j > 0) {
            if (i > 0 && D[i - 1][j] + 1 == D[i][j]) { o[i - 1] = '-'; --i; }
            else if (j > 0 && D[i][j - 1] + 1 == D[i][j]) { deleted = 1; break; }
            else { o[i - 1] = q[j - 1]; --i; --j; }
        }
        if (!deleted) return out;
        memset(o, '-', (size_t)L);
    }
    /* the offset with the most matching characters (the first maximum above 0 matches) */
    Py_ssize_t max_p = 0, max_matches = 0;
    for (Py_ssize_t p = 0; p + m <= L; ++p) {
        Py_ssize_t nr = 0;
        for (Py_ssize_t t = 0; t < m; ++t) nr += q[t] == mx[p + t];
        if (nr > max_matches) { max_p = p; max_matches = nr; }
    }
    if (max_p > 0) memcpy(o + max_p, q, (size_t)m);
    else memcpy(o, q, (size_t)(m < L ? m : L));
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
    /* The sort runs without the GIL on the strings' own buffers: a snapshot of the list with a strong reference to every element keeps
     * them alive whatever another thread does to the caller's list meanwhile. */
    PyObject *hold = PyList_GetSlice(seqs, 0, n);
    if (!hold) { free(items); return NULL; }
    for (Py_ssize_t i = 0; i < n; ++i) {
        if (PyList_GET_ITEM(hold, i) != PyList_GET_ITEM(seqs, i) || (const char *)PyUnicode_1BYTE_DATA(PyList_GET_ITEM(hold, i)) != items[i].p) {
            Py_DECREF(hold); free(items);
            PyErr_SetString(PyExc_RuntimeError, "rank_strings: the list changed during the call");
            return NULL;
        }
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
    Py_DECREF(hold);
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
        if (!PyUnicode_CheckExact(value)) {          /* (a __hash__ / __eq__ written in Python could change d under PyDict_Next) */
            Py_DECREF(out);
            PyErr_SetString(PyExc_TypeError, "group_keys_by_value: every value must be a str");
            return NULL;
        }
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

/* unique_values_by_length(S: dict[acc, str]) -> (seqs: list[str], accs: list): the unique VALUES of S -- a value keeps the position of its
 * first appearance and the key of its LAST (what {seq: acc for acc, seq in S.items()} gives, nearest_neighbor_graph.py:243) -- stably sorted
 * by length (:246), and the keys that go with them.  One pass and a counting sort instead of a dict comprehension, a list of items, a
 * key-function sort and two list comprehensions over 50 000 entries. */
static PyObject *unique_values_by_length(PyObject *self, PyObject *args)
{
    PyObject *S;
    if (!PyArg_ParseTuple(args, "O", &S)) return NULL;
    if (!PyDict_CheckExact(S)) { PyErr_SetString(PyExc_TypeError, "unique_values_by_length: a dict is required"); return NULL; }
    PyObject *inv = PyDict_New();
    if (!inv) return NULL;
    Py_ssize_t pos = 0, maxlen = 0;
    PyObject *key, *value;
    while (PyDict_Next(S, &pos, &key, &value)) {
        if (!PyUnicode_CheckExact(value) || PyUnicode_READY(value) < 0) {
            Py_DECREF(inv);
            if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "unique_values_by_length: every value must be a str");
            return NULL;
        }
        if (PyDict_SetItem(inv, value, key) < 0) { Py_DECREF(inv); return NULL; }
        if (PyUnicode_GET_LENGTH(value) > maxlen) maxlen = PyUnicode_GET_LENGTH(value);
    }
    const Py_ssize_t n = PyDict_GET_SIZE(inv);
    Py_ssize_t *start = (Py_ssize_t *)calloc((size_t)maxlen + 2, sizeof(Py_ssize_t));
    PyObject *seqs = PyList_New(n), *accs = PyList_New(n);
    if (!start || !seqs || !accs) { free(start); Py_XDECREF(seqs); Py_XDECREF(accs); Py_DECREF(inv); return PyErr_NoMemory(); }
    pos = 0;
    while (PyDict_Next(inv, &pos, &key, &value)) start[PyUnicode_GET_LENGTH(key) + 1] += 1;
    for (Py_ssize_t l = 0; l <= maxlen; ++l) start[l + 1] += start[l];
    pos = 0;
    while (PyDict_Next(inv, &pos, &key, &value)) {          /* insertion order inside a length: stable */
        const Py_ssize_t at = start[PyUnicode_GET_LENGTH(key)]++;
        Py_INCREF(key); Py_INCREF(value);
        PyList_SET_ITEM(seqs, at, key);
        PyList_SET_ITEM(accs, at, value);
    }
    free(start);
    Py_DECREF(inv);
    PyObject *out = PyTuple_Pack(2, seqs, accs);
    Py_DECREF(seqs); Py_DECREF(accs);
    return out;
}

/* flatten_pairs(matches: dict) -> (pairs: list[(k1, k2)], values: list): the (outer key, inner key) pairs of a dict of dicts -- or of a dict
 * of sets / lists / tuples (then values = None) -- in iteration order: the task lists of edlib_align_sequences / sw_align_sequences
 * (edlib_alignment_module.py:17-24, SW_alignment_module.py:96-118) without 50 000 rounds of the interpreter loop. */
static PyObject *flatten_pairs(PyObject *self, PyObject *args)
{
    PyObject *matches;
    if (!PyArg_ParseTuple(args, "O", &matches)) return NULL;
    if (!PyDict_CheckExact(matches)) { PyErr_SetString(PyExc_TypeError, "flatten_pairs: a dict is required"); return NULL; }
    PyObject *pairs = PyList_New(0), *values = PyList_New(0);
    if (!pairs || !values) { Py_XDECREF(pairs); Py_XDECREF(values); return NULL; }
    int have_values = -1;          /* -1 unknown, 1 inner dicts, 0 other iterables */
    Py_ssize_t pos = 0;
    PyObject *k1, *inner;
    while (PyDict_Next(matches, &pos, &k1, &inner)) {
        if (PyDict_CheckExact(inner)) {
            if (have_values == 0) goto mixed;
            have_values = 1;
            Py_ssize_t p2 = 0;
            PyObject *k2, *v;
            while (PyDict_Next(inner, &p2, &k2, &v)) {
                PyObject *t = PyTuple_Pack(2, k1, k2);
                if (!t || PyList_Append(pairs, t) < 0 || PyList_Append(values, v) < 0) { Py_XDECREF(t); goto fail; }
                Py_DECREF(t);
            }
        } else {
            if (have_values == 1) goto mixed;
            have_values = 0;
            /* only containers whose iteration runs no Python code (it could change `matches` under PyDict_Next); anything else: the caller's loop */
            if (!PyAnySet_CheckExact(inner) && !PyList_CheckExact(inner) && !PyTuple_CheckExact(inner)) {
                PyErr_SetString(PyExc_TypeError, "flatten_pairs: inner values must be dict, set, frozenset, list or tuple");
                goto fail;
            }
            PyObject *it = PyObject_GetIter(inner);
            if (!it) goto fail;
            PyObject *k2;
            while ((k2 = PyIter_Next(it)) != NULL) {
                PyObject *t = PyTuple_Pack(2, k1, k2);
                Py_DECREF(k2);
                if (!t || PyList_Append(pairs, t) < 0) { Py_XDECREF(t); Py_DECREF(it); goto fail; }
                Py_DECREF(t);
            }
            Py_DECREF(it);
            if (PyErr_Occurred()) goto fail;
        }
    }
    {
        PyObject *out = PyTuple_Pack(2, pairs, have_values == 1 ? values : Py_None);
        Py_DECREF(pairs); Py_DECREF(values);
        return out;
    }
mixed:
    PyErr_SetString(PyExc_TypeError, "flatten_pairs: inner values are dicts for some keys and not for others");
fail:
    Py_DECREF(pairs); Py_DECREF(values);
    return NULL;
}

/* distance_dict(pairs: list[(k1, k2)], ed_addr: int (int32[len(pairs)])) -> {k1: {k2: ed}} in the pairs' order: the return value of
 * edlib_align_sequences (edlib_alignment_module.py:42-49: every pair's distance filed under its two keys) without 50 000 rounds of the
 * interpreter loop (13 of the call's 19 ms at C3). */
static PyObject *distance_dict(PyObject *self, PyObject *args)
{
    PyObject *pairs;
    unsigned long long ea;
    if (!PyArg_ParseTuple(args, "OK", &pairs, &ea)) return NULL;
    if (!PyList_CheckExact(pairs)) { PyErr_SetString(PyExc_TypeError, "distance_dict: a list of pairs is required"); return NULL; }
    const int32_t *ed = (const int32_t *)(uintptr_t)ea;
    const Py_ssize_t n = PyList_GET_SIZE(pairs);
    PyObject *out = PyDict_New();
    if (!out) return NULL;
    PyObject *last_key = NULL, *last_row = NULL;          /* (consecutive pairs of one outer key: no second lookup) */
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject *p = PyList_GET_ITEM(pairs, i);
        if (!PyTuple_CheckExact(p) || PyTuple_GET_SIZE(p) != 2) { PyErr_Format(PyExc_TypeError, "pair %zd is not a 2-tuple", i); Py_DECREF(out); return NULL; }
        PyObject *k1 = PyTuple_GET_ITEM(p, 0), *k2 = PyTuple_GET_ITEM(p, 1);
        PyObject *row = k1 == last_key ? last_row : PyDict_GetItemWithError(out, k1);
        if (!row) {
            if (PyErr_Occurred()) { Py_DECREF(out); return NULL; }
            row = PyDict_New();
            if (!row || PyDict_SetItem(out, k1, row) < 0) { Py_XDECREF(row); Py_DECREF(out); return NULL; }
            Py_DECREF(row);          /* (the outer dict holds it) */
        }
        last_key = k1; last_row = row;
        PyObject *v = PyLong_FromLong((long)ed[i]);
        if (!v || PyDict_SetItem(row, k2, v) < 0) { Py_XDECREF(v); Py_DECREF(out); return NULL; }
        Py_DECREF(v);
    }
    return out;
}

/* pairs_of(matches: dict of (dict | set | frozenset | list | tuple), index: dict[str, int], a_addr, b_addr, cap) ->
 * (outer: list, counts_addr-less: list[int], inner: list) or None: the (outer key, inner key) pairs of `matches` in iteration order WITHOUT a tuple
 * per pair -- outer keys with at least one member, how many members each has, the members flat -- and their ids under `index` written to the
 * uint32 arrays a / b (capacity cap pairs).  None when a key is not in `index` (the caller then packs its own store) or cap is too small.
 * With distance_rows below: edlib_align_sequences (edlib_alignment_module.py:10-49) on 49 990 pairs in 10 instead of 16 ms. */
static PyObject *pairs_of(PyObject *self, PyObject *args)
{
    PyObject *matches, *index;
    unsigned long long aa, ba;
    Py_ssize_t cap;
    if (!PyArg_ParseTuple(args, "OOKKn", &matches, &index, &aa, &ba, &cap)) return NULL;
    if (!PyDict_CheckExact(matches) || !PyDict_CheckExact(index)) { PyErr_SetString(PyExc_TypeError, "pairs_of: two dicts are required"); return NULL; }
    uint32_t *a = (uint32_t *)(uintptr_t)aa, *b = (uint32_t *)(uintptr_t)ba;
    PyObject *outer = PyList_New(0), *counts = PyList_New(0), *inner_keys = PyList_New(0);
    if (!outer || !counts || !inner_keys) goto fail;
    {
        Py_ssize_t pos = 0, n = 0;
        PyObject *k1, *inner;
        while (PyDict_Next(matches, &pos, &k1, &inner)) {
            /* only containers whose iteration runs no Python code (it could change `matches` under PyDict_Next) */
            if (!PyDict_CheckExact(inner) && !PyAnySet_CheckExact(inner) && !PyList_CheckExact(inner) && !PyTuple_CheckExact(inner)) {
                PyErr_SetString(PyExc_TypeError, "pairs_of: inner values must be dict, set, frozenset, list or tuple");
                goto fail;
            }
            PyObject *v1 = NULL;
            Py_ssize_t cnt = 0;
            PyObject *it = PyObject_GetIter(inner);
            if (!it) goto fail;
            PyObject *k2;
            while ((k2 = PyIter_Next(it)) != NULL) {
                if (!v1) v1 = PyDict_GetItemWithError(index, k1);
                PyObject *v2 = v1 ? PyDict_GetItemWithError(index, k2) : NULL;
                if (!v1 || !v2 || n >= cap) {
                    Py_DECREF(k2); Py_DECREF(it);
                    if (PyErr_Occurred()) goto fail;
                    Py_DECREF(outer); Py_DECREF(counts); Py_DECREF(inner_keys);
                    Py_RETURN_NONE;
                }
                const unsigned long x1 = PyLong_AsUnsignedLong(v1), x2 = PyLong_AsUnsignedLong(v2);
                if ((x1 == (unsigned long)-1 || x2 == (unsigned long)-1) && PyErr_Occurred()) { Py_DECREF(k2); Py_DECREF(it); goto fail; }
                a[n] = (uint32_t)x1; b[n] = (uint32_t)x2; ++n; ++cnt;
                const int rc = PyList_Append(inner_keys, k2);
                Py_DECREF(k2);
                if (rc < 0) { Py_DECREF(it); goto fail; }
            }
            Py_DECREF(it);
            if (PyErr_Occurred()) goto fail;
            if (cnt) {
                PyObject *c = PyLong_FromSsize_t(cnt);
                if (!c || PyList_Append(outer, k1) < 0 || PyList_Append(counts, c) < 0) { Py_XDECREF(c); goto fail; }
                Py_DECREF(c);
            }
        }
    }
    {
        PyObject *out = PyTuple_Pack(3, outer, counts, inner_keys);
        Py_DECREF(outer); Py_DECREF(counts); Py_DECREF(inner_keys);
        return out;
    }
fail:
    Py_XDECREF(outer); Py_XDECREF(counts); Py_XDECREF(inner_keys);
    return NULL;
}

/* distance_rows(outer: list, counts: list[int], inner: list, ed_addr: int (int32[len(inner)])) -> {outer[r]: {inner[p]: ed[p]}}: the rows of
 * pairs_of with their distances (a key that occurs again keeps its dict and gets the later values, as the reference's loop would) */
static PyObject *distance_rows(PyObject *self, PyObject *args)
{
    PyObject *outer, *counts, *inner;
    unsigned long long ea;
    if (!PyArg_ParseTuple(args, "OOOK", &outer, &counts, &inner, &ea)) return NULL;
    if (!PyList_CheckExact(outer) || !PyList_CheckExact(counts) || !PyList_CheckExact(inner) || PyList_GET_SIZE(outer) != PyList_GET_SIZE(counts)) {
        PyErr_SetString(PyExc_TypeError, "distance_rows: three lists (outer keys, their counts, inner keys) are required");
        return NULL;
    }
    const int32_t *ed = (const int32_t *)(uintptr_t)ea;
    const Py_ssize_t n_in = PyList_GET_SIZE(inner);
    PyObject *out = PyDict_New();
    if (!out) return NULL;
    Py_ssize_t p = 0;
    for (Py_ssize_t r = 0; r < PyList_GET_SIZE(outer); ++r) {
        const Py_ssize_t cnt = PyLong_AsSsize_t(PyList_GET_ITEM(counts, r));
        if (cnt < 0 || p + cnt > n_in) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "distance_rows: counts do not match the inner keys"); Py_DECREF(out); return NULL; }
        PyObject *k1 = PyList_GET_ITEM(outer, r);
        PyObject *row = PyDict_GetItemWithError(out, k1);
        if (!row) {
            if (PyErr_Occurred()) { Py_DECREF(out); return NULL; }
            row = PyDict_New();
            if (!row || PyDict_SetItem(out, k1, row) < 0) { Py_XDECREF(row); Py_DECREF(out); return NULL; }
            Py_DECREF(row);
        }
        for (Py_ssize_t e = 0; e < cnt; ++e, ++p) {
            PyObject *v = PyLong_FromLong((long)ed[p]);
            if (!v || PyDict_SetItem(row, PyList_GET_ITEM(inner, p), v) < 0) { Py_XDECREF(v); Py_DECREF(out); return NULL; }
            Py_DECREF(v);
        }
    }
    if (p != n_in) { PyErr_SetString(PyExc_ValueError, "distance_rows: counts do not match the inner keys"); Py_DECREF(out); return NULL; }
    return out;
}

/* alignment_dict(pairs: list[(k1, k2)], aln_a: list[str], aln_b: list[str], res_addr: int (int32[n][6]: .., matches, mismatches, indels)) ->
 * (out: list of (aln_a[p], aln_b[p], (matches, mismatches, indels)), d: {k1: {k2: out[p]}}): the return value of sw_align_sequences
 * (SW_alignment_module.py:146-164: every pair's stats filed under its two keys) and the flat list of the same tuple objects. */
static PyObject *alignment_dict(PyObject *self, PyObject *args)
{
    PyObject *pairs, *la, *lb;
    unsigned long long ra;
    Py_ssize_t n_res;
    if (!PyArg_ParseTuple(args, "OOOKn", &pairs, &la, &lb, &ra, &n_res)) return NULL;
    if (!PyList_Check(pairs) || !PyList_Check(la) || !PyList_Check(lb) || PyList_GET_SIZE(la) != PyList_GET_SIZE(pairs) ||
        PyList_GET_SIZE(lb) != PyList_GET_SIZE(pairs)) {
        PyErr_SetString(PyExc_TypeError, "alignment_dict: three lists of one length are required");
        return NULL;
    }
    if (n_res < PyList_GET_SIZE(pairs)) { PyErr_SetString(PyExc_ValueError, "alignment_dict: fewer result rows than pairs"); return NULL; }
    const int32_t *res = (const int32_t *)(uintptr_t)ra;
    const Py_ssize_t n = PyList_GET_SIZE(pairs);
    PyObject *out = PyList_New(n), *d = PyDict_New();
    if (!out || !d) { Py_XDECREF(out); Py_XDECREF(d); return NULL; }
    for (Py_ssize_t p = 0; p < n; ++p) {
        PyObject *pr = PyList_GET_ITEM(pairs, p);
        if (!PyTuple_Check(pr) || PyTuple_GET_SIZE(pr) < 2) { PyErr_Format(PyExc_TypeError, "pair %zd is not a 2-tuple", p); goto fail; }
        PyObject *cnt = PyTuple_New(3);          /* (matches, mismatches, indels) */
        if (!cnt) goto fail;
        for (int k = 0; k < 3; ++k) {
            PyObject *v = PyLong_FromLong((long)res[p * 6 + 3 + k]);
            if (!v) { Py_DECREF(cnt); goto fail; }
            PyTuple_SET_ITEM(cnt, k, v);
        }
        PyObject *t = PyTuple_Pack(3, PyList_GET_ITEM(la, p), PyList_GET_ITEM(lb, p), cnt);
        Py_DECREF(cnt);
        if (!t) goto fail;
        PyList_SET_ITEM(out, p, t);          /* (steals t; the dict below takes its own reference) */
        PyObject *k1 = PyTuple_GET_ITEM(pr, 0), *k2 = PyTuple_GET_ITEM(pr, 1);
        PyObject *row = PyDict_GetItemWithError(d, k1);          /* borrowed */
        if (!row) {
            if (PyErr_Occurred()) goto fail;
            row = PyDict_New();
            if (!row) goto fail;
            const int rc = PyDict_SetItem(d, k1, row);
            Py_DECREF(row);
            if (rc < 0) goto fail;
        }
        if (PyDict_SetItem(row, k2, t) < 0) goto fail;
    }
    {
        PyObject *r = PyTuple_Pack(2, out, d);
        Py_DECREF(out); Py_DECREF(d);
        return r;
    }
fail:
    Py_DECREF(out); Py_DECREF(d);
    return NULL;
}

/* lazy_rows(cls, batch, pairs: list[(m, s)], keep_addr: int (uint8[n]: 1 = the pair stays), edit_addr: int (int32[n]), out: dict[m, dict],
 *           rows_of: dict) -> number of values made: out[m][s] = cls(batch, p, edit[p]) and rows_of[m].append(p) for every kept pair p, in
 * pair order -- the loop that files 50 000 alignments under their centres in get_partition_alignments (isocon_get_candidates.py:66-76). */
static PyObject *lazy_rows(PyObject *self, PyObject *args)
{
    PyObject *cls, *batch, *pairs, *out, *rows_of;
    unsigned long long ka, ea;
    Py_ssize_t n_arr;
    if (!PyArg_ParseTuple(args, "OOOKKnOO", &cls, &batch, &pairs, &ka, &ea, &n_arr, &out, &rows_of)) return NULL;
    if (!PyList_Check(pairs) || !PyDict_Check(out) || !PyDict_Check(rows_of)) { PyErr_SetString(PyExc_TypeError, "lazy_rows: list, dict, dict"); return NULL; }
    const uint8_t *keep = (const uint8_t *)(uintptr_t)ka;
    const int32_t *edit = (const int32_t *)(uintptr_t)ea;
    const Py_ssize_t n = PyList_GET_SIZE(pairs);
    if (n_arr != n) { PyErr_SetString(PyExc_ValueError, "lazy_rows: keep / edit hold another number of entries than pairs"); return NULL; }
    Py_ssize_t made = 0;
    for (Py_ssize_t p = 0; p < n; ++p) {
        /* (cls(...) below runs Python code: it must not have shortened the list) */
        if (PyList_GET_SIZE(pairs) != n) { PyErr_SetString(PyExc_RuntimeError, "lazy_rows: pairs changed size during the call"); return NULL; }
        if (!keep[p]) continue;
        PyObject *pr = PyList_GET_ITEM(pairs, p);
        if (!PyTuple_Check(pr) || PyTuple_GET_SIZE(pr) < 2) { PyErr_Format(PyExc_TypeError, "pair %zd is not a 2-tuple", p); return NULL; }
        PyObject *m = PyTuple_GET_ITEM(pr, 0), *s = PyTuple_GET_ITEM(pr, 1);
        PyObject *row = PyDict_GetItemWithError(out, m);          /* borrowed */
        if (!row) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_KeyError, "lazy_rows: a centre without a row"); return NULL; }
        PyObject *pi = PyLong_FromSsize_t(p), *ei = PyLong_FromLong((long)edit[p]);
        PyObject *v = (pi && ei) ? PyObject_CallFunctionObjArgs(cls, batch, pi, ei, NULL) : NULL;
        Py_XDECREF(ei);
        if (!v) { Py_XDECREF(pi); return NULL; }
        int rc = PyDict_SetItem(row, s, v);
        Py_DECREF(v);
        if (rc < 0) { Py_DECREF(pi); return NULL; }
        PyObject *lst = PyDict_GetItemWithError(rows_of, m);          /* borrowed */
        if (!lst) {
            if (PyErr_Occurred()) { Py_DECREF(pi); return NULL; }
            lst = PyList_New(0);
            if (!lst) { Py_DECREF(pi); return NULL; }
            rc = PyDict_SetItem(rows_of, m, lst);
            Py_DECREF(lst);
            if (rc < 0) { Py_DECREF(pi); return NULL; }
        }
        rc = PyList_Append(lst, pi);
        Py_DECREF(pi);
        if (rc < 0) return NULL;
        ++made;
    }
    return PyLong_FromSsize_t(made);
}

/* lazy_rows_intact(partition: dict, centre, cls, batch, pairs: list) -> bool: every value of `partition` except the centre's own is an
 * instance of exactly `cls` whose `_batch` is `batch` and whose `_p` names the pair (centre, that key) of `pairs` -- i.e. the dict still
 * holds what lazy_rows filed there and nothing else, so the correction may read members and alignments from the batch's arrays instead
 * of from the dict (correction_module.correct_strings). */
static PyObject *lazy_rows_intact(PyObject *self, PyObject *args)
{
    PyObject *partition, *centre, *cls, *batch, *pairs;
    if (!PyArg_ParseTuple(args, "OOOOO", &partition, &centre, &cls, &batch, &pairs)) return NULL;
    if (!PyDict_Check(partition) || !PyList_Check(pairs)) { PyErr_SetString(PyExc_TypeError, "lazy_rows_intact: a dict and a list are required"); return NULL; }
    static PyObject *s_batch = NULL, *s_p = NULL;
    if (!s_batch) { s_batch = PyUnicode_InternFromString("_batch"); s_p = PyUnicode_InternFromString("_p"); }
    if (!s_batch || !s_p) return NULL;
    const Py_ssize_t n = PyList_GET_SIZE(pairs);
    Py_ssize_t pos = 0;
    PyObject *key, *value;
    while (PyDict_Next(partition, &pos, &key, &value)) {
        if (key == centre) continue;
        if ((PyObject *)Py_TYPE(value) != cls) Py_RETURN_FALSE;
        PyObject *b = PyObject_GetAttr(value, s_batch);
        if (!b) return NULL;
        Py_DECREF(b);
        if (b != batch) Py_RETURN_FALSE;
        PyObject *pi = PyObject_GetAttr(value, s_p);
        if (!pi) return NULL;
        const Py_ssize_t p = PyLong_AsSsize_t(pi);
        Py_DECREF(pi);
        if (p == -1 && PyErr_Occurred()) return NULL;
        if (p < 0 || p >= n) Py_RETURN_FALSE;
        PyObject *pr = PyList_GET_ITEM(pairs, p);
        if (!PyTuple_Check(pr) || PyTuple_GET_SIZE(pr) < 2 || PyTuple_GET_ITEM(pr, 0) != centre || PyTuple_GET_ITEM(pr, 1) != key) Py_RETURN_FALSE;
    }
    Py_RETURN_TRUE;
}

static PyMethodDef methods[] = {
    {"lazy_rows_intact", lazy_rows_intact, METH_VARARGS, "the values of a partition are still the lazily expanded alignments filed there"},
    {"unique_values_by_length", unique_values_by_length, METH_VARARGS, "unique values of a dict, stably sorted by length, with their last keys"},
    {"flatten_pairs", flatten_pairs, METH_VARARGS, "(outer key, inner key) pairs of a dict of dicts / sets, and the inner values"},
    {"distance_dict", distance_dict, METH_VARARGS, "dict of dicts of the distances of a list of pairs"},
    {"pairs_of", pairs_of, METH_VARARGS, "pairs of a dict of containers without a tuple per pair, and their ids"},
    {"distance_rows", distance_rows, METH_VARARGS, "dict of dicts of the distances of the rows of pairs_of"},
    {"alignment_dict", alignment_dict, METH_VARARGS, "alignment tuples and the dict of dicts that files them under their pairs"},
    {"lazy_rows", lazy_rows, METH_VARARGS, "files lazily expanded alignment values under their centres"},
    {"group_keys_by_value", group_keys_by_value, METH_VARARGS, "{value: [keys]} of a dict, in insertion order"},
    {"rank_strings", rank_strings, METH_VARARGS, "rank of every string of a list in the sorted order of the list"},
    {"pair_ids", pair_ids, METH_VARARGS, "ids of the members of a list of pairs"},
    {"csr_to_dict", csr_to_dict, METH_VARARGS, "dict of dicts from the CSR arrays of a nearest-neighbour graph"},
    {"str_pointers", str_pointers, METH_VARARGS, "addresses and lengths of a list of ASCII str"},
    {"split_ascii", split_ascii, METH_VARARGS, "list of str cut out of an ASCII buffer"},
    {"best_solution", best_solution, METH_VARARGS, "functions.get_best_solution on ASCII strings, as bytes"},
    {"invariant_partners", invariant_partners, METH_VARARGS, "indices of the strings of a list that equal a string up to their ends"},
    {"split_ascii_rows", split_ascii_rows, METH_VARARGS, "list of str for selected rows of an ASCII buffer, equal rows sharing one object"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_pyhelp", NULL, -1, methods};
PyMODINIT_FUNC PyInit__pyhelp(void) { return PyModule_Create(&moddef); }
