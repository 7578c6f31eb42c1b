"""Drop-in for the one function of /root/reference/modules/functions.py that runs on every SW output right after the hot
path: filter_exon_differences (+ get_mask_start_and_end).

Two evaluation routes with identical results:
  * alignments produced by isocon_amd.SW_alignment_module in this process still have their CIGAR ops cached -> the
    rule is evaluated on the run-length ops by isocon_exon_filter_from_ops (C, one batch call);
  * any other (s1_alignment, s2_alignment, counts) tuple -> evaluated on the gapped strings.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import SW_alignment_module, _lib


def get_mask_start_and_end(aln_t, aln_c):
    """functions.py:218-236: columns covered by an end gap of either string."""
    mask_start, mask_end = 0, len(aln_t)
    lead_t = len(aln_t) - len(aln_t.lstrip("-"))
    trail_t = len(aln_t) - len(aln_t.rstrip("-"))
    if lead_t:
        mask_start = lead_t
    if trail_t:
        mask_end = len(aln_t) - trail_t
    lead_c = len(aln_c) - len(aln_c.lstrip("-"))
    trail_c = len(aln_c) - len(aln_c.rstrip("-"))
    if lead_c:
        assert mask_start == 0
        mask_start = lead_c
    if trail_c:
        assert mask_end == len(aln_t)
        mask_end = len(aln_c) - trail_c
    return mask_start, mask_end


def _flag_from_strings(s1_alignment, s2_alignment, min_exon_diff, ignore_ends_len):
    start, end = get_mask_start_and_end(s1_alignment, s2_alignment)
    start = min(ignore_ends_len, start)
    end = max(len(s1_alignment) - ignore_ends_len, end)
    gap = "-" * min_exon_diff
    return gap in s1_alignment[start:end] or gap in s2_alignment[start:end]


def filter_exon_differences(pairwise_alignments, min_exon_diff, ignore_ends_len):
    """functions.py:23-50: deletes, in place, every pairwise_alignments[s1][s2] whose alignment shows a gap run of at
    least min_exon_diff columns away from the (masked) ends; returns the set of the deleted s2 keys."""
    keys = [(s1, s2) for s1 in list(pairwise_alignments.keys()) for s2 in list(pairwise_alignments[s1].keys())]
    cache = SW_alignment_module._OPS_CACHE
    flags = [None] * len(keys)
    with_ops = [i for i, (s1, s2) in enumerate(keys) if id(pairwise_alignments[s1][s2]) in cache]
    if with_ops:
        L = _lib.load()
        ops_list = [cache[id(pairwise_alignments[keys[i][0]][keys[i][1]])][1] for i in with_ops]
        ptr = np.zeros(len(ops_list) + 1, dtype=np.uint64)
        np.cumsum([len(o) for o in ops_list], out=ptr[1:])
        ops = np.ascontiguousarray(np.concatenate(ops_list) if ops_list else np.zeros(0, np.uint32), dtype=np.uint32)
        if ops.size == 0:
            ops = np.zeros(1, np.uint32)
        out = np.zeros(len(ops_list), dtype=np.uint8)
        _lib.check(L.isocon_exon_filter_from_ops(ops.ctypes.data_as(_lib.u32p), ptr.ctypes.data_as(_lib.u64p), len(ops_list),
                                                 int(min_exon_diff), int(ignore_ends_len), out.ctypes.data_as(_lib.u8p)),
                   "isocon_exon_filter_from_ops")
        for i, f in zip(with_ops, out.tolist()):
            flags[i] = bool(f)
    filtered = set()
    for i, (s1, s2) in enumerate(keys):
        f = flags[i]
        if f is None:
            s1_alignment, s2_alignment, _counts = pairwise_alignments[s1][s2]
            f = _flag_from_strings(s1_alignment, s2_alignment, min_exon_diff, ignore_ends_len)
        if f:
            del pairwise_alignments[s1][s2]
            filtered.add(s2)
    return filtered
