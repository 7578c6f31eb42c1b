"""Packed sequence set resident in HBM (isocon_store of include/isocon_hip.h)."""
from __future__ import annotations

import ctypes
import weakref
from collections import OrderedDict

import numpy as np

from . import _lib


def _ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None


_utf8 = ctypes.pythonapi.PyUnicode_AsUTF8AndSize
_utf8.restype = ctypes.c_void_p
_utf8.argtypes = [ctypes.py_object, ctypes.POINTER(ctypes.c_ssize_t)]


class SeqStore(object):
    """Uploads a list of ACGT strings once; ids are positions in that list."""

    def __init__(self, seqs, private_pool=False):
        """private_pool: the store keeps its own scratch (bound matrix, held edges, counters) instead of the process-wide pool
        (ISOCON_STORE_PRIVATE_SCRATCH): several ranks of the sharded search emulated inside one process, tests/baton_dist.py"""
        L = _lib.lib()
        self.n = len(seqs)
        h = ctypes.c_void_p()
        H = _lib.pyhelp()
        if H is not None and isinstance(seqs, list):
            # the strings' own buffers go to the library, which gathers them into its pinned staging buffer (no 125 MB join)
            ptrs = np.empty(max(self.n, 1), dtype=np.uint64)
            lens = np.empty(max(self.n, 1), dtype=np.uint64)
            try:
                H.str_pointers(seqs, ptrs.ctypes.data, lens.ctypes.data)
            except ValueError as e:
                raise _lib.IsoconError("isocon_store_create failed: %s" % e)
            self.lens = lens[:self.n].astype(np.int64)
            if private_pool:
                _lib.check(L.isocon_store_create_ptrs_ex(_ptr(ptrs, _lib.u64p), _ptr(lens, _lib.u64p), self.n, 1, ctypes.byref(h)), "isocon_store_create")
            else:
                _lib.check(L.isocon_store_create_ptrs(_ptr(ptrs, _lib.u64p), _ptr(lens, _lib.u64p), self.n, ctypes.byref(h)), "isocon_store_create")
        else:
            lens = np.fromiter(map(len, seqs), dtype=np.uint64, count=self.n)
            self.lens = lens.astype(np.int64)
            off = np.zeros(self.n + 1, dtype=np.uint64)
            np.cumsum(lens, out=off[1:])
            # One join, no second copy: CPython keeps an all-ASCII str as one byte per character, and PyUnicode_AsUTF8AndSize hands
            # out that very buffer (any non-ASCII symbol makes the sizes differ -- and is outside the alphabet anyway).
            joined = "".join(seqs)
            size = ctypes.c_ssize_t(0)
            addr = _utf8(ctypes.py_object(joined), ctypes.byref(size)) if joined else None
            if joined and (not addr or size.value != int(off[self.n])):
                raise _lib.IsoconError("isocon_store_create failed: sequence contains a symbol outside ACGT (non-ASCII character)")
            dummy = (ctypes.c_uint8 * 1)()
            if private_pool:
                ptrs = (off[:-1] + np.uint64(addr or 0)).astype(np.uint64) if self.n else np.zeros(1, np.uint64)
                lens64 = np.ascontiguousarray(lens if self.n else np.zeros(1, np.uint64), dtype=np.uint64)
                _lib.check(L.isocon_store_create_ptrs_ex(_ptr(ptrs, _lib.u64p), _ptr(lens64, _lib.u64p), self.n, 1, ctypes.byref(h)), "isocon_store_create")
            else:
                _lib.check(L.isocon_store_create(ctypes.cast(addr, _lib.u8p) if addr else dummy, _ptr(off, _lib.u64p), self.n, ctypes.byref(h)),
                           "isocon_store_create")
            del joined
        self._h = h
        self._L = L
        self._fingerprint = None

    @property
    def fingerprint(self):
        """Identity of the packed set, order included (isocon_store_digest: every plane word and length, hashed on the
        device): the ranks of a sharded run compare it.  Computed on first use."""
        if self._fingerprint is None:
            out = np.zeros(1, dtype=np.uint64)
            _lib.check(self._L.isocon_store_digest(self._h, _ptr(out, _lib.u64p)), "isocon_store_digest")
            self._fingerprint = int(out[0] >> np.uint64(1))          # 63 bits: travels as int64 through the collectives
        return self._fingerprint

    @property
    def handle(self):
        return self._h

    def device_bytes(self) -> int:
        return int(self._L.isocon_store_device_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._L.isocon_store_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- batched edit distance over explicit pairs ------------------------------------------------------------
    def ed_pairs(self, a, b, k=None, return_ms=False):
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        if len(a) != len(b):
            raise ValueError("pair arrays differ in length")
        kk = None if k is None else np.ascontiguousarray(k, dtype=np.int32)
        out = np.full(len(a), -1, dtype=np.int32)
        ms = ctypes.c_float(0)
        _lib.check(self._L.isocon_ed_pairs(self._h, _ptr(a, _lib.u32p), _ptr(b, _lib.u32p), _ptr(kk, _lib.i32p), len(a),
                                           _ptr(out, _lib.i32p), ctypes.byref(ms)), "isocon_ed_pairs")
        return (out, ms.value) if return_ms else out

    def qgram_bound_pairs(self, a, b):
        """Lower bounds of ed(a[i], b[i]) from q-gram count profiles (csrc/qgram_mm.hpp) -- the pre-filter of the NN main pass."""
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        if len(a) != len(b):
            raise ValueError("pair arrays differ in length")
        out = np.zeros(len(a), dtype=np.int32)
        _lib.check(self._L.isocon_qgram_bound_pairs(self._h, _ptr(a, _lib.u32p), _ptr(b, _lib.u32p), len(a), _ptr(out, _lib.i32p)),
                   "isocon_qgram_bound_pairs")
        return out

    def block_bound_pairs(self, owner, partner, probe_stride=4):
        """Second lower bound of ed(owner[i], partner[i]) (csrc/nn_filter.hpp): the greedy count of disjoint 8-grams of the partner that
        occur nowhere in the owner -- what the main pass tests on the survivors of the q-gram bound (isocon_block_bound_pairs; tests)."""
        a = np.ascontiguousarray(owner, dtype=np.uint32)
        b = np.ascontiguousarray(partner, dtype=np.uint32)
        if len(a) != len(b):
            raise ValueError("pair arrays differ in length")
        out = np.zeros(len(a), dtype=np.int32)
        _lib.check(self._L.isocon_block_bound_pairs(self._h, _ptr(a, _lib.u32p), _ptr(b, _lib.u32p), len(a), probe_stride, _ptr(out, _lib.i32p)), "isocon_block_bound_pairs")
        return out

    def qgram_bound_matrix(self, q_begin=0, q_end=None, q_stride=1, depth=2 ** 32, q_block=1):
        """The bound matrix the NN main pass consults for the shard (q_begin, q_end, q_stride, q_block) (isocon_qgram_bound_matrix; tests):
        (row_ptr[rows + 1], bytes) -- row r = the shard's r-th entry (shard_entries) against the entries behind it within 63 of its length."""
        q_end = self.n if q_end is None else min(int(q_end), self.n)
        rows = len(shard_entries(q_begin, q_end, q_stride, q_block))
        row_ptr = np.zeros(rows + 1, dtype=np.uint64)
        needed = ctypes.c_uint64(0)
        cap = 1 << 20
        depth = int(min(depth, 2 ** 63 - 1))
        while True:
            out = np.empty(cap, dtype=np.uint8)
            rc = self._L.isocon_qgram_bound_matrix(self._h, q_begin, q_end, q_stride, q_block, depth, _ptr(row_ptr, _lib.u64p), _ptr(out, _lib.u8p), cap,
                                                   ctypes.byref(needed))
            if rc == _lib.ISOCON_E_CAPACITY:
                cap = int(needed.value)
                continue
            _lib.check(rc, "isocon_qgram_bound_matrix")
            return row_ptr, out[:int(row_ptr[-1])]

    def hw_pairs(self, q, t, k, return_ms=False, reuse_buffer=False):
        """Infix (edlib "HW", task="path") alignment of sequence q[p] inside sequence t[p] with threshold k[p]:
        int32 [n, 5] = distance (-1 if > k), start, end, leading insertion run, trailing insertion run
        (isocon_hw_pairs; reference call site end_invariant_functions.py:594)."""
        q = np.ascontiguousarray(q, dtype=np.uint32)
        t = np.ascontiguousarray(t, dtype=np.uint32)
        kk = np.ascontiguousarray(np.broadcast_to(np.asarray(k, dtype=np.int32), q.shape))
        if len(q) != len(t):
            raise ValueError("pair arrays differ in length")
        # (every row is written by the call)  reuse_buffer: the rows land in a pinned buffer that the NEXT such call overwrites -- for
        # callers that digest the result at once (millions of pairs: 100 MB come down without a staged copy)
        if reuse_buffer and len(q):
            out = _host_buffer("hw_out", len(q) * 20, self._L).view(np.int32).reshape(len(q), 5)
        else:
            out = np.empty((len(q), 5), dtype=np.int32)
        ms = ctypes.c_float(0)
        _lib.check(self._L.isocon_hw_pairs(self._h, _ptr(q, _lib.u32p), _ptr(t, _lib.u32p), _ptr(kk, _lib.i32p), len(q),
                                           _ptr(out, _lib.i32p), ctypes.byref(ms)), "isocon_hw_pairs")
        return (out, ms.value) if return_ms else out

    # ---- nearest-neighbour graph (store must be length-sorted) ------------------------------------------------
    def nn_graph(self, is_converged=None, is_target=None, depth=2 ** 32):
        """Returns (best[n], row_ptr[n+1], cols, stats dict)."""
        n = self.n
        conv = None if is_converged is None else np.ascontiguousarray(is_converged, dtype=np.uint8)
        targ = None if is_target is None else np.ascontiguousarray(is_target, dtype=np.uint8)
        best = np.empty(max(n, 1), dtype=np.int32)
        row_ptr = np.zeros(n + 1, dtype=np.uint64)
        cap = max(4 * n, 1024)
        needed = ctypes.c_uint64(0)
        stats = _lib.NNStats()
        depth = int(min(depth, 2 ** 63 - 1))
        while True:
            cols = np.empty(cap, dtype=np.uint32)
            rc = self._L.isocon_nn_graph(self._h, _ptr(conv, _lib.u8p), _ptr(targ, _lib.u8p), depth, _ptr(best, _lib.i32p),
                                         _ptr(row_ptr, _lib.u64p), _ptr(cols, _lib.u32p), cap, ctypes.byref(needed),
                                         ctypes.byref(stats))
            if rc == _lib.ISOCON_E_CAPACITY:
                cap = int(needed.value) + 16
                continue
            _lib.check(rc, "isocon_nn_graph")
            return best[:n], row_ptr.astype(np.int64), cols[:int(row_ptr[n])], stats.as_dict()

    def nn_partial(self, q_begin, q_end, phase, best, is_converged=None, is_target=None, depth=2 ** 32, q_stride=1, q_block=1, wide_queries=None):
        """One shard / one phase (see include/isocon_hip.h).  `best` is updated in place; returns (hits[k,3], stats)."""
        conv = None if is_converged is None else np.ascontiguousarray(is_converged, dtype=np.uint8)
        targ = None if is_target is None else np.ascontiguousarray(is_target, dtype=np.uint8)
        assert best.dtype == np.int32 and best.flags.c_contiguous and len(best) >= self.n
        cap = 48 * self.n + (1 << 20)          # generous (np.empty costs nothing until touched): a retry repeats the phase
        n_hits = ctypes.c_uint64(0)
        stats = _lib.NNStats()
        depth = int(min(depth, 2 ** 63 - 1))
        best0 = best.copy()
        wq = None if wide_queries is None else np.ascontiguousarray(wide_queries, dtype=np.uint8)
        while True:
            hits = np.empty((cap, 3), dtype=np.int32)
            rc = self._L.isocon_nn_partial(self._h, _ptr(conv, _lib.u8p), _ptr(targ, _lib.u8p), depth, q_begin, q_end, q_stride, q_block, phase,
                                           _ptr(best, _lib.i32p), _ptr(hits, _lib.i32p), cap, ctypes.byref(n_hits),
                                           ctypes.byref(stats), _ptr(wq, _lib.u8p))
            if rc == _lib.ISOCON_E_CAPACITY:
                cap = int(n_hits.value) + 16
                best[:] = best0
                continue
            _lib.check(rc, "isocon_nn_partial")
            return hits[:int(n_hits.value)], stats.as_dict()

    # the same protocol with best[] and the candidate edges resident in device memory (isocon_amd/dist.py on a GPU)
    def nn_partial_dev(self, q_begin, q_end, phase, best_dev_ptr, keep_hits, is_converged=None, is_target=None, depth=2 ** 32, q_stride=1, q_block=1,
                       wide_queries=None):
        """One shard / one phase on best[] in device memory (an int: the device address of n int32).  The phase's candidate edges join
        the list the library holds on the device (keep_hits False: a new list).  Returns (edges held, stats)."""
        conv = None if is_converged is None else np.ascontiguousarray(is_converged, dtype=np.uint8)
        targ = None if is_target is None else np.ascontiguousarray(is_target, dtype=np.uint8)
        held = ctypes.c_uint64(0)
        stats = _lib.NNStats()
        wq = None if wide_queries is None else np.ascontiguousarray(wide_queries, dtype=np.uint8)
        _lib.check(self._L.isocon_nn_partial_dev(self._h, _ptr(conv, _lib.u8p), _ptr(targ, _lib.u8p), int(min(depth, 2 ** 63 - 1)), q_begin, q_end,
                                                 q_stride, q_block, phase, ctypes.c_void_p(int(best_dev_ptr)), 1 if keep_hits else 0, ctypes.byref(held),
                                                 ctypes.byref(stats), _ptr(wq, _lib.u8p)), "isocon_nn_partial_dev")
        return int(held.value), stats.as_dict()

    def nn_hits_dev(self, best_dev_ptr, out_dev_ptr, cap_rows):
        """The held edges that attain best[] of their endpoint -> out (device, cap_rows x 3 int32; unused rows -1)."""
        _lib.check(self._L.isocon_nn_hits_dev(self._h, ctypes.c_void_p(int(best_dev_ptr)), ctypes.c_void_p(int(out_dev_ptr)), int(cap_rows), None),
                   "isocon_nn_hits_dev")

    def nn_finalize_dev(self, best_dev_ptr, hits_dev_ptr, n_rows):
        """(best, row_ptr, cols) from reduced best[] and gathered edges in device memory (isocon_nn_finalize_dev)."""
        n = self.n
        out_best = np.empty(max(n, 1), dtype=np.int32)
        row_ptr = np.zeros(n + 1, dtype=np.uint64)
        cap = max(4 * n, 1024)
        needed = ctypes.c_uint64(0)
        while True:
            cols = np.empty(cap, dtype=np.uint32)
            rc = self._L.isocon_nn_finalize_dev(self._h, ctypes.c_void_p(int(best_dev_ptr)), ctypes.c_void_p(int(hits_dev_ptr)), int(n_rows),
                                                _ptr(out_best, _lib.i32p), _ptr(row_ptr, _lib.u64p), _ptr(cols, _lib.u32p), cap, ctypes.byref(needed))
            if rc == _lib.ISOCON_E_CAPACITY:
                cap = int(needed.value) + 16
                continue
            _lib.check(rc, "isocon_nn_finalize_dev")
            return out_best[:n], row_ptr.astype(np.int64), cols[:int(row_ptr[n])]

    # ---- consensus correction with the multi-alignment matrix built on the device (csrc/msa_build.hpp) ----------
    def msa_build_ops(self, row_ids, ops, ops_ptr):
        """The partition row_ids[0] (centre) + row_ids[1:] (members) with the members' CIGAR ops against the centre (ops_ptr[0] = ops_ptr[1] = 0)
        -> (n_cols, col_slot uint32[Lm + 1], longest uint32[Lm + 1], wide uint32[k, 8] = row, slot, position in the member, length, 2-bit codes
        of the first 32 bases (two words), two spare words).
        The matrix stays in device memory for msa_correct_built."""
        row_ids = np.ascontiguousarray(row_ids, dtype=np.uint32)
        ops = np.ascontiguousarray(ops, dtype=np.uint32)
        ops_ptr = np.ascontiguousarray(ops_ptr, dtype=np.uint64)
        nr = len(row_ids)
        Lm = int(self.lens[int(row_ids[0])])
        n_cols = ctypes.c_uint32(0)
        col_slot = np.zeros(Lm + 1, dtype=np.uint32)
        longest = np.zeros(Lm + 1, dtype=np.uint32)
        cap = 65536
        n_wide = ctypes.c_uint64(0)
        while True:
            wide = np.empty((cap, 8), dtype=np.uint32)
            rc = self._L.isocon_msa_build_ops(self._h, nr, _ptr(row_ids, _lib.u32p), _ptr(ops if len(ops) else None, _lib.u32p), _ptr(ops_ptr, _lib.u64p),
                                              ctypes.byref(n_cols), _ptr(col_slot, _lib.u32p), _ptr(longest, _lib.u32p), _ptr(wide, _lib.u32p), cap,
                                              ctypes.byref(n_wide), None)
            if rc == _lib.ISOCON_E_CAPACITY:
                cap = int(n_wide.value) + 16
                continue
            _lib.check(rc, "isocon_msa_build_ops")
            return int(n_cols.value), col_slot, longest, wide[:int(n_wide.value)]

    def msa_correct_built(self, n_rows, n_cols, degree, patch_row=None, patch_col=None, patch_ptr=None, patch_bytes=None):
        """Patches into the built matrix, then the correction (isocon_msa_correct_built) -> (packed uint8[], offsets int64[n_rows + 1], n_cand)."""
        deg = np.ascontiguousarray(degree, dtype=np.int32)
        n_p = 0 if patch_row is None else len(patch_row)
        if n_p:
            patch_row = np.ascontiguousarray(patch_row, dtype=np.uint32)
            patch_col = np.ascontiguousarray(patch_col, dtype=np.uint32)
            patch_ptr = np.ascontiguousarray(patch_ptr, dtype=np.uint32)
            patch_bytes = np.ascontiguousarray(patch_bytes, dtype=np.uint8)
        cap = int(n_rows) * int(n_cols)
        packed = np.empty(max(cap, 1), dtype=np.uint8)
        offsets = np.zeros(n_rows + 1, dtype=np.uint64)
        n_cand = np.zeros(n_rows, dtype=np.int32)
        _lib.check(self._L.isocon_msa_correct_built(self._h, n_rows, n_cols, _ptr(patch_row if n_p else None, _lib.u32p), _ptr(patch_col if n_p else None, _lib.u32p),
                                                    _ptr(patch_ptr if n_p else None, _lib.u32p), _ptr(patch_bytes if n_p else None, _lib.u8p), n_p,
                                                    _ptr(deg, _lib.i32p), _ptr(packed, _lib.u8p), cap, _ptr(offsets, _lib.u64p), _ptr(n_cand, _lib.i32p),
                                                    None, None), "isocon_msa_correct_built")
        return packed, offsets.astype(np.int64), n_cand

    def msa_build_ops_batch(self, first_row, row_ids, ops, ops_ptr):
        """All partitions of a correction step at once (isocon_msa_build_ops_batch): partition p = rows first_row[p] .. first_row[p + 1] of
        the concatenation (its first row the centre).  -> (n_cols uint32[P], slot_base int64[P + 1], col_slot, longest (concatenated slot
        arrays), wide uint32[k, 8] with the row of the concatenation in word 0 and the partition in word 6)."""
        first_row = np.ascontiguousarray(first_row, dtype=np.uint32)
        row_ids = np.ascontiguousarray(row_ids, dtype=np.uint32)
        ops = np.ascontiguousarray(ops, dtype=np.uint32)
        ops_ptr = np.ascontiguousarray(ops_ptr, dtype=np.uint64)
        n_parts = len(first_row) - 1
        Lm = self.lens[row_ids[first_row[:-1]]].astype(np.int64)
        slot_base = np.zeros(n_parts + 1, dtype=np.int64)
        np.cumsum(Lm + 1, out=slot_base[1:])
        n_cols = np.zeros(n_parts, dtype=np.uint32)
        col_slot = np.zeros(int(slot_base[-1]), dtype=np.uint32)
        longest = np.zeros(int(slot_base[-1]), dtype=np.uint32)
        # room for the wide-slot records: a too-small buffer means building everything again (two 3.7 ms fill launches per correction step at
        # C3 with the old fixed 2^17); what the previous call of this store needed is the best guess for the next one
        cap = max(1 << 17, 8 * len(row_ids), int(1.25 * getattr(self, "_msa_wide_seen", 0)) + 16)
        n_wide = ctypes.c_uint64(0)
        while True:
            wide = np.empty((cap, 8), dtype=np.uint32)
            rc = self._L.isocon_msa_build_ops_batch(self._h, n_parts, _ptr(first_row, _lib.u32p), _ptr(row_ids, _lib.u32p), _ptr(ops if len(ops) else None, _lib.u32p),
                                                    _ptr(ops_ptr, _lib.u64p), _ptr(n_cols, _lib.u32p), _ptr(col_slot, _lib.u32p), _ptr(longest, _lib.u32p),
                                                    _ptr(wide, _lib.u32p), cap, ctypes.byref(n_wide), None)
            if rc == _lib.ISOCON_E_CAPACITY:
                cap = int(n_wide.value) + 16
                continue
            _lib.check(rc, "isocon_msa_build_ops_batch")
            self._msa_wide_seen = int(n_wide.value)
            return n_cols, slot_base, col_slot, longest, wide[:int(n_wide.value)]

    def msa_correct_built_batch(self, n_parts, n_rows, degree, packed_cap, patch_row=None, patch_col=None, patch_ptr=None, patch_bytes=None):
        """Patches + correction of the batch built by msa_build_ops_batch -> (packed uint8[], offsets int64[n_rows + 1], n_cand int32[n_rows])."""
        deg = np.ascontiguousarray(degree, dtype=np.int32)
        n_p = 0 if patch_row is None else len(patch_row)
        if n_p:
            patch_row = np.ascontiguousarray(patch_row, dtype=np.uint32)
            patch_col = np.ascontiguousarray(patch_col, dtype=np.uint32)
            patch_ptr = np.ascontiguousarray(patch_ptr, dtype=np.uint32)
            patch_bytes = np.ascontiguousarray(patch_bytes, dtype=np.uint8)
        offsets = np.zeros(n_rows + 1, dtype=np.uint64)
        n_cand = np.zeros(n_rows, dtype=np.int32)
        cap = int(packed_cap)
        while True:
            packed = _host_buffer("msa_packed", max(cap, 1), self._L)          # (pinned and reused: 125 MB per step at 50 000 x 2.5 kb)
            rc = self._L.isocon_msa_correct_built_batch(self._h, n_parts, n_rows, _ptr(patch_row if n_p else None, _lib.u32p), _ptr(patch_col if n_p else None, _lib.u32p),
                                                        _ptr(patch_ptr if n_p else None, _lib.u32p), _ptr(patch_bytes if n_p else None, _lib.u8p), n_p,
                                                        _ptr(deg, _lib.i32p), _ptr(packed, _lib.u8p), cap, _ptr(offsets, _lib.u64p), _ptr(n_cand, _lib.i32p), None)
            if rc == _lib.ISOCON_E_CAPACITY:
                cap = int(offsets[n_rows]) + 16
                continue
            _lib.check(rc, "isocon_msa_correct_built_batch")
            return packed, offsets.astype(np.int64), n_cand

    def msa_read_built(self, n_rows, n_cols):
        M = np.zeros((n_rows, n_cols), dtype=np.uint8)
        _lib.check(self._L.isocon_msa_read_built(self._h, n_rows, n_cols, _ptr(M, _lib.u8p)), "isocon_msa_read_built")
        return M

    # ---- semi-global affine alignment with traceback ----------------------------------------------------------
    def sg_trace(self, a, b, mismatch, match=2, open_=2, ext=0, tie_policy=0, return_ms=False, ed_upper=None):
        """Returns (ops uint32[], ops_ptr int64[n+1], res int32[n,6]) -- see include/isocon_hip.h."""
        a = np.ascontiguousarray(a, dtype=np.uint32)
        b = np.ascontiguousarray(b, dtype=np.uint32)
        n = len(a)
        mm = np.ascontiguousarray(np.broadcast_to(np.asarray(mismatch, dtype=np.int8), (n,)))
        res = np.zeros((max(n, 1), 6), dtype=np.int32)
        ops_ptr = np.zeros(n + 1, dtype=np.uint64)
        tot = int((self.lens[a].astype(np.int64) + self.lens[b]).sum()) if n else 0
        cap = max(64 * n + tot // 4, 1024)     # generous: a too-small buffer means running the batch again
        edu = None if ed_upper is None else np.ascontiguousarray(np.broadcast_to(np.asarray(ed_upper, dtype=np.int32), (n,)))
        needed = ctypes.c_uint64(0)
        ms = ctypes.c_float(0)
        while True:
            ops = np.empty(cap, dtype=np.uint32)
            rc = self._L.isocon_sg_trace_batch(self._h, _ptr(a, _lib.u32p), _ptr(b, _lib.u32p), n, match, _ptr(mm, _lib.i8p),
                                               open_, ext, tie_policy, _ptr(ops, _lib.u32p), _ptr(ops_ptr, _lib.u64p), cap,
                                               ctypes.byref(needed), _ptr(res, _lib.i32p), ctypes.byref(ms), _ptr(edu, _lib.i32p))
            if rc == _lib.ISOCON_E_CAPACITY:
                cap = int(needed.value) + 16
                continue
            _lib.check(rc, "isocon_sg_trace_batch")
            out = (ops[:int(ops_ptr[n])], ops_ptr.astype(np.int64), res[:n])
            return out + (ms.value,) if return_ms else out


def sg_last_stats():
    """where the kernel time of this thread's most recent alignment batch went (isocon_sg_last_stats, include/isocon_hip.h)"""
    out = (ctypes.c_double * 11)()
    _lib.load().isocon_sg_last_stats(out, 11)
    keys = ("forward_ms", "walk_ms", "compact_ms", "expand_ms", "pairs_band", "pairs_strips", "pairs_redone", "trace_bytes", "pairs_band_narrow",
            "pairs_tried_narrow", "pairs_retried_wider")
    return dict(zip(keys, list(out)))


def shard_entries(q_begin, q_end, q_stride=1, q_block=1):
    """The entries a shard owns, in slot order (include/isocon_hip.h, isocon_nn_partial): q_begin <= x < q_end with
    (x - q_begin) mod q_stride < q_block."""
    if q_begin >= q_end:
        return np.zeros(0, dtype=np.int64)
    x = np.arange(q_begin, q_end, dtype=np.int64)
    return x[(x - q_begin) % q_stride < q_block]


_HOST_BUFFERS = {}


def _host_buffer(name, nbytes, L):
    """A grow-only uint8 buffer in pinned host memory (isocon_host_alloc) that the NEXT call of the same wrapper overwrites;
    ordinary memory if pinning fails.  A block that is outgrown is NOT freed here: views handed out earlier (the result rows of
    hw_pairs(reuse_buffer=True), the gapped strings of sg_strings) may still be alive, so the block is released by a finalizer
    once the last of them is gone."""
    cur = _HOST_BUFFERS.get(name)
    if cur is not None and cur[0] >= nbytes:
        return cur[1][:nbytes]
    cap = int(nbytes + nbytes // 2 + 4096)
    addr = L.isocon_host_alloc(cap)
    if addr:
        block = (ctypes.c_uint8 * cap).from_address(addr)
        weakref.finalize(block, L.isocon_host_free, addr)          # runs when every numpy view / memoryview of the block is gone
        arr = np.frombuffer(block, dtype=np.uint8)
    else:
        arr = np.empty(cap, dtype=np.uint8)
    _HOST_BUFFERS[name] = (cap, arr)
    return arr[:nbytes]


def _sg_strings(self, a, b, mismatch, match=2, open_=2, ext=0, tie_policy=0, return_ops=False, ed_upper=None):
    """Gapped strings straight from the device: returns (aln_a bytes, aln_b bytes, aln_ptr int64[n+1], res int32[n,6]).
    The two byte buffers are views of buffers the next call reuses: decode them before calling again."""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    n = len(a)
    mm = np.ascontiguousarray(np.broadcast_to(np.asarray(mismatch, dtype=np.int8), (n,)))
    res = np.zeros((max(n, 1), 6), dtype=np.int32)
    ops_ptr = np.zeros(n + 1, dtype=np.uint64)
    aln_ptr = np.zeros(n + 1, dtype=np.uint64)
    # generous first guesses (untouched pages of np.empty cost nothing): a too-small buffer means running the batch again
    tot = int((self.lens[a].astype(np.int64) + self.lens[b]).sum()) if n else 0
    ops_cap = max(64 * n + tot // 4, 1024)
    aln_cap = tot + 1024                      # an alignment is never longer than both sequences together
    edu = None if ed_upper is None else np.ascontiguousarray(np.broadcast_to(np.asarray(ed_upper, dtype=np.int32), (n,)))
    need_ops, need_aln = ctypes.c_uint64(0), ctypes.c_uint64(0)
    while True:
        ops = np.empty(ops_cap, dtype=np.uint32)
        aln_a = _host_buffer("aln_a", aln_cap, self._L)         # pinned and reused: no page faults, no staged copy (2 x 130 MB at C3)
        aln_b = _host_buffer("aln_b", aln_cap, self._L)
        rc = self._L.isocon_sg_strings_batch(self._h, _ptr(a, _lib.u32p), _ptr(b, _lib.u32p), n, match, _ptr(mm, _lib.i8p), open_, ext,
                                             tie_policy, _ptr(ops, _lib.u32p), _ptr(ops_ptr, _lib.u64p), ops_cap, ctypes.byref(need_ops),
                                             _ptr(res, _lib.i32p), _ptr(aln_a, _lib.u8p), _ptr(aln_b, _lib.u8p), _ptr(aln_ptr, _lib.u64p),
                                             aln_cap, ctypes.byref(need_aln), None, _ptr(edu, _lib.i32p))
        if rc == _lib.ISOCON_E_CAPACITY:
            ops_cap = max(ops_cap, int(need_ops.value) + 16)
            aln_cap = max(aln_cap, int(need_aln.value) + 16)
            continue
        _lib.check(rc, "isocon_sg_strings_batch")
        end = int(aln_ptr[n])
        # (views, not copies: the caller decodes them once -- a 50 k x 2.5 kb batch is 2 x 130 MB)
        out = (memoryview(aln_a[:end]), memoryview(aln_b[:end]), aln_ptr.astype(np.int64), res[:n])
        if return_ops:
            out = out + (ops[:int(ops_ptr[n])].copy(), ops_ptr.astype(np.int64))
        return out


SeqStore.sg_strings = _sg_strings


def nn_finalize(n, best, hits):
    """Host-side CSR assembly from reduced best[] and gathered hit triples (isocon_nn_finalize)."""
    L = _lib.load()
    best = np.ascontiguousarray(best, dtype=np.int32)
    hits = np.ascontiguousarray(hits, dtype=np.int32).reshape(-1, 3)
    out_best = np.empty(max(n, 1), dtype=np.int32)
    row_ptr = np.zeros(n + 1, dtype=np.uint64)
    cap = max(len(hits), 16)
    cols = np.empty(cap, dtype=np.uint32)
    needed = ctypes.c_uint64(0)
    _lib.check(L.isocon_nn_finalize(n, _ptr(best, _lib.i32p), _ptr(hits, _lib.i32p), len(hits), _ptr(out_best, _lib.i32p),
                                    _ptr(row_ptr, _lib.u64p), _ptr(cols, _lib.u32p), cap, ctypes.byref(needed)),
               "isocon_nn_finalize")
    return out_best[:n], row_ptr.astype(np.int64), cols[:int(row_ptr[n])]


# The reference's callers hand the SAME sequence set to NNG, then EAM, then SWM within one correction iteration
# (isocon_get_candidates.py:129-130,38,47).  The most recent store is kept (with a sequence -> id index) so the pair-list
# wrappers can address it instead of packing and uploading the strings again.
_RECENT = {"store": None, "index": None}


class _LazyIndex(object):
    """sequence -> id of the remembered store, built on first use (a caller that only wants the graph never pays for it)"""

    def __init__(self, seqs):
        self._seqs = seqs
        self._d = None

    def __getitem__(self, s):
        return self.mapping()[s]

    def mapping(self):
        if self._d is None:
            self._d = {x: i for i, x in enumerate(self._seqs)}
        return self._d


def remember(store, seqs):
    old = _RECENT["store"]
    if old is not None and old is not store:
        old.close()
    _RECENT["store"] = store
    _RECENT["index"] = _LazyIndex(seqs)


def forget():
    if _RECENT["store"] is not None:
        _RECENT["store"].close()
    _RECENT["store"] = None
    _RECENT["index"] = None


def store_for_pairs(pairs, only_remembered=False):
    """[(x, y), ...] -> (store, a ids, b ids, owned).  Reuses the remembered store when it holds every sequence of the
    pair list; otherwise packs a private one (owned = True: the caller closes it) -- or, with only_remembered, packs nothing and
    returns (None, None, None, False): a caller that can only use the remembered store finds out before anything is uploaded."""
    index = _RECENT["index"]
    n = len(pairs)
    a = np.empty(n, dtype=np.uint32)
    b = np.empty(n, dtype=np.uint32)
    if index is not None and _RECENT["store"] is not None and getattr(_RECENT["store"], "_h", None):
        try:
            d = index.mapping()
            H = _lib.pyhelp()
            if H is not None and hasattr(H, "pair_ids") and isinstance(pairs, list) and isinstance(d, dict):
                if H.pair_ids(d, pairs, a.ctypes.data, b.ctypes.data) == n:          # (50 000 pairs: the lookups in C, 25 -> 5 ms)
                    return _RECENT["store"], a, b, False
                raise KeyError("a sequence of the pair list is not in the remembered store")
            a = np.fromiter((d[x] for x, _ in pairs), dtype=np.uint32, count=n)
            b = np.fromiter((d[y] for _, y in pairs), dtype=np.uint32, count=n)
            return _RECENT["store"], a, b, False
        except KeyError:
            a = np.empty(n, dtype=np.uint32)
            b = np.empty(n, dtype=np.uint32)
    if only_remembered:
        return None, None, None, False
    index, seqs = {}, []
    for p, (x, y) in enumerate(pairs):
        ia = index.get(x)
        if ia is None:
            ia = index[x] = len(seqs)
            seqs.append(x)
        ib = index.get(y)
        if ib is None:
            ib = index[y] = len(seqs)
            seqs.append(y)
        a[p], b[p] = ia, ib
    return SeqStore(seqs), a, b, True
