"""GPU drop-in for /root/reference/modules/SW_alignment_module.py (SWM): same names, arguments and return shapes.

The semi-global affine alignments + tracebacks come from isocon_sg_trace_batch (include/isocon_hip.h), which
replaces parasail.sg_trace_scan_16/32 and the CIGAR decode of parasail_alignment (SWM:64-86).  TIE_POLICY selects
the trace-back tie rules (0 = parasail's, as restated in oracle/isocon_oracle.c).  `nr_cores` is accepted and ignored.
"""
from __future__ import annotations

import sys

import numpy as np

from .store import store_for_pairs

TIE_POLICY = 0
_OPS = "=XID"
# id(result tuple) -> (the tuple, the batch's CIGAR ops as uint32 array, begin, end) for the alignments of the most recent batch call; lets
# isocon_amd.functions.filter_exon_differences work on run-length ops instead of re-scanning the gapped strings.
class _OpsCache(object):
    """the result tuples of the most recent batch with their run-length ops; the id -> position index is built when first asked for"""

    def __init__(self):
        self.clear()

    def clear(self):
        self.out, self.ops, self.ptr, self._index = None, None, None, None

    def set(self, out, ops, ptr):
        self.out, self.ops, self.ptr, self._index = out, ops, ptr, None

    def _idx(self):
        if self._index is None:
            self._index = {id(t): p for p, t in enumerate(self.out)} if self.out else {}
        return self._index

    def __contains__(self, key):
        return key in self._idx()

    def get(self, key):
        p = self._idx().get(key)
        return None if p is None else (self.out[p], self.ops, self.ptr[p], self.ptr[p + 1])


_OPS_CACHE = _OpsCache()


def cigar_to_seq(cigar, query, ref):
    """SWM:15-56: CIGAR string -> two gapped strings ('I' consumes the query, 'D' the reference)."""
    q_aln, r_aln = [], []
    q_index = r_index = 0
    length_ = 0
    for ch in cigar:
        if "0" <= ch <= "9":
            length_ = length_ * 10 + ord(ch) - 48
            continue
        if ch == "=" or ch == "X":
            q_aln.append(query[q_index:q_index + length_])
            r_aln.append(ref[r_index:r_index + length_])
            q_index += length_
            r_index += length_
        elif ch == "I":
            q_aln.append(query[q_index:q_index + length_])
            r_aln.append("-" * length_)
            q_index += length_
        elif ch == "D":
            q_aln.append("-" * length_)
            r_aln.append(ref[r_index:r_index + length_])
            r_index += length_
        else:
            print("error")
            print(cigar)
            sys.exit()  # SWM:51-54
        length_ = 0
    return "".join(q_aln), "".join(r_aln)


def _ops_to_alignment(ops, query, ref):
    """(len << 4 | code) ops -> gapped strings, without the detour over a CIGAR string."""
    q_aln, r_aln = [], []
    qi = ri = 0
    for op in ops:
        ln, code = op >> 4, op & 15
        if code < 2:
            q_aln.append(query[qi:qi + ln]); r_aln.append(ref[ri:ri + ln]); qi += ln; ri += ln
        elif code == 2:
            q_aln.append(query[qi:qi + ln]); r_aln.append("-" * ln); qi += ln
        else:
            q_aln.append("-" * ln); r_aln.append(ref[ri:ri + ln]); ri += ln
    return "".join(q_aln), "".join(r_aln)


def ops_to_cigar(ops):
    return "".join("%d%s" % (op >> 4, _OPS[op & 15]) for op in ops)


def _align_pairs(pairs, mismatch, match_score=2, opening_penalty=2, gap_ext=0, ed_upper=None, want_dict=False):
    """[(s1, s2)], per-pair mismatch penalties -> [(s1_aln, s2_aln, (matches, mismatches, indels))].
    mismatch None: the penalty of every pair from the error-rate bucket of its ed_upper (SWM:102-109).
    ed_upper: the pairs' edit distances where the caller has them (they only narrow the computed part of the matrix;
    the device re-aligns in full whatever it cannot certify, see include/isocon_hip.h)."""
    if not pairs:
        return ([], {}) if want_dict else []
    from . import perf_log
    with perf_log.call("SW_alignment_module.alignments", pairs=len(pairs), open=opening_penalty, ext=gap_ext, hints=ed_upper is not None):
        return _align_pairs_impl(pairs, mismatch, match_score, opening_penalty, gap_ext, ed_upper, want_dict)


def _align_pairs_impl(pairs, mismatch, match_score, opening_penalty, gap_ext, ed_upper, want_dict=False):
    """want_dict: returns (alignments, {pairs[p][0]: {pairs[p][1]: alignments[p]}} or None when the helper module is missing)"""
    st, a, b, owned = store_for_pairs(pairs)
    try:
        la, lb = st.lens[a], st.lens[b]
        if bool((la == 0).any() or (lb == 0).any()):
            raise ValueError("empty sequence in an alignment pair")
        if mismatch is None:
            # SWM:102-109 for the whole list: error_rate = ed / min(len) as IEEE doubles, like the reference's float division
            rate = np.asarray(ed_upper, dtype=np.float64) / np.minimum(la, lb).astype(np.float64)
            mismatch = np.where(rate <= 0.01, -1, np.where(rate <= 0.09, -2, -4)).astype(np.int8)
            ed_upper = np.where((np.asarray(ed_upper) >= 0) & (np.asarray(ed_upper) < 2 ** 30), ed_upper, -1).astype(np.int32)
        if ed_upper is None and len(pairs) >= 64:
            # callers without distances (hypothesis_test_module's candidate pairs, parasail_alignment batches): the device
            # computes them first (isocon_ed_pairs, ~1e7 pairs/s) -- with a bound the alignment runs inside its certified band
            # (4-9x less work for related sequences) and comes out identical; unrelated pairs fall back to the full matrix
            ed_upper = st.ed_pairs(a, b, None)
        aln_a, aln_b, ptr, res, ops, ops_ptr = st.sg_strings(a, b, np.asarray(mismatch, dtype=np.int8), match=match_score,
                                                             open_=opening_penalty, ext=gap_ext, tie_policy=TIE_POLICY,
                                                             return_ops=True, ed_upper=ed_upper)
    finally:
        if owned:
            st.close()
    from . import _lib
    H = _lib.pyhelp()
    if H is not None:
        # the 2 x 50 000 gapped strings are cut straight out of the (pinned) byte buffers
        ptr64 = np.ascontiguousarray(ptr, dtype=np.int64)
        ba, bb = np.frombuffer(aln_a, dtype=np.uint8), np.frombuffer(aln_b, dtype=np.uint8)
        sa = H.split_ascii(ba.ctypes.data if len(ba) else 0, ptr64.ctypes.data, len(pairs))
        sb = H.split_ascii(bb.ctypes.data if len(bb) else 0, ptr64.ctypes.data, len(pairs))
        if want_dict and hasattr(H, "alignment_dict") and isinstance(pairs, list):
            # the result tuples and the dict of dicts that files them under their pairs, in one pass (SWM:146-164)
            res_c = np.ascontiguousarray(res, dtype=np.int32)
            out, filed = H.alignment_dict(pairs, sa, sb, res_c.ctypes.data, len(res_c))
            _OPS_CACHE.set(out, ops, ops_ptr)
            return out, filed
        counts = list(zip(res[:, 3].tolist(), res[:, 4].tolist(), res[:, 5].tolist()))          # (matches, mismatches, indels) tuples
        out = list(zip(sa, sb, counts))
    else:
        counts = list(zip(res[:, 3].tolist(), res[:, 4].tolist(), res[:, 5].tolist()))
        aln_a = str(aln_a, "ascii")
        aln_b = str(aln_b, "ascii")
        ptr = ptr.tolist()
        out = [(aln_a[ptr[p]:ptr[p + 1]], aln_b[ptr[p]:ptr[p + 1]], counts[p]) for p in range(len(pairs))]
    _OPS_CACHE.set(out, ops, ops_ptr)          # (indexed and sliced by whoever asks: ops_of)
    return (out, None) if want_dict else out


def ops_of(result_tuple):
    """CIGAR ops (uint32 array) of an alignment tuple returned by the most recent batch call, or None"""
    c = _OPS_CACHE.get(id(result_tuple))
    if c is None or c[0] is not result_tuple:
        return None
    return c[1][int(c[2]):int(c[3])]


def parasail_alignment(s1, s2, i, j, x_acc="", y_acc="", match_score=2, mismatch_penalty=-3, opening_penalty=2, gap_ext=0):
    """SWM:64-86 (single pair)."""
    stats = _align_pairs([(s1, s2)], [mismatch_penalty], match_score, opening_penalty, gap_ext)[0]
    if x_acc == y_acc == "":
        return (s1, s2, stats)
    return (x_acc, y_acc, stats)


def parasail_alignment_helper(arguments):
    """SWM:59-61."""
    args, kwargs = arguments
    return parasail_alignment(*args, **kwargs)


def _penalty(ed, s1, s2):
    """SWM:102-109: mismatch penalty from the error-rate bucket of the INPUT edit distance."""
    error_rate = float(ed) / min(len(s1), len(s2))
    if error_rate <= 0.01:
        return -1
    elif 0.01 < error_rate <= 0.09:
        return -2
    return -4


def _ed_hint(ed):
    """The input edit distance as a band hint (-1 = none).  It is only a hint: the device certifies or redoes."""
    try:
        ed = int(ed)
    except (TypeError, ValueError):
        return -1
    return ed if 0 <= ed < 2 ** 30 else -1


def _int_distances(vals):
    """the callers' edit distances as an int64 array, or None if one of them is not an integer (then: the per-pair path)"""
    try:
        arr = np.asarray(vals)
        if arr.dtype.kind in "iu" and arr.ndim == 1:
            return arr.astype(np.int64)
    except (TypeError, ValueError, OverflowError):
        pass
    return None


def _batch(keys_pairs_eds):
    """[(key pair, (s1, s2), ed)] -> alignments in that order; penalties per pair as in SWM:102-109"""
    pairs = [t[1] for t in keys_pairs_eds]
    eds = _int_distances([t[2] for t in keys_pairs_eds])
    if eds is not None:
        return _align_pairs(pairs, None, ed_upper=eds)
    pens = [_penalty(t[2], t[1][0], t[1][1]) for t in keys_pairs_eds]
    return _align_pairs(pairs, pens, ed_upper=[_ed_hint(t[2]) for t in keys_pairs_eds])


def sw_align_sequences(matches, nr_cores=1, mismatch_penalty=-1):
    """SWM:89-164.  {s1: {s2: ed}} -> {s1: {s2: (s1_aln, s2_aln, (matches, mismatches, indels))}}."""
    from . import _lib
    H = _lib.pyhelp()
    if H is not None and hasattr(H, "flatten_pairs") and type(matches) is dict and all(type(v) is dict for v in matches.values()):
        # the pair list and the distances in one pass in C; the result dict is filed in C as well (every alignment tuple is truthy: SWM:158)
        pairs, eds = H.flatten_pairs(matches)
        eds = _int_distances(eds) if eds else None
        if eds is not None:
            out, filed = _align_pairs(pairs, None, ed_upper=eds, want_dict=True)
            if filed is not None:
                return filed
            exact_matches = {}
            for (s1, s2), stats in zip(pairs, out):
                exact_matches.setdefault(s1, {})[s2] = stats
            return exact_matches
    items = [((s1, s2), (s1, s2), ed) for s1, inner in matches.items() for s2, ed in inner.items()]
    exact_matches = {}
    for (key, _, _), stats in zip(items, _batch(items)):
        if stats:
            exact_matches.setdefault(key[0], {})[key[1]] = stats
    return exact_matches


def sw_align_sequences_keeping_accession(matches, nr_cores=1):
    """SWM:167-249.  {acc1: {acc2: (s1, s2, ed)}} -> {acc1: {acc2: (s1_aln, s2_aln, (matches, mismatches, indels))}}."""
    items = [((a1, a2), (v[0], v[1]), v[2]) for a1, inner in matches.items() for a2, v in inner.items()]
    exact_matches = {}
    for (key, _, _), stats in zip(items, _batch(items)):
        if stats:
            exact_matches.setdefault(key[0], {})[key[1]] = stats
    return exact_matches
