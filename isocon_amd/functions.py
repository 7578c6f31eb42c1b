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


# ---- multi-alignment of a partition around its centre -----------------------------------------
# Mirror of /root/reference/modules/functions.py:526-799 (create_position_frequency_matrix, create_multialignment_matrix,
# position_query_to_alignment, get_best_solution, create_multialignment_format_NEW, min_ed).  Same column layout as the
# reference: for every centre base one column, and in front of / behind every base one insertion slot that is 1 column
# wide unless some read inserts >= 2 bases there, in which case it is len(longest insertion) + 2 columns wide and every
# other insertion is placed inside it by get_best_solution.  The matrix is built as a numpy uint8 array (the reference
# builds dicts of character lists); `create_multialignment_matrix` converts it to the reference's shape.

def nw_path_cigar(query, target):
    """Extended CIGAR of one optimal global unit-cost alignment -- stands where the reference calls
    edlib.align(query, target, task="path", mode="NW") (functions.py:772).  Which optimum edlib reports is pinned by
    nothing in the reference ("parity unpinned"); backtracking from the end this prefers a query-only step ('I'), then
    a target-only step ('D'), then the diagonal -- the same rule as the stand-in the golden fixtures were made with."""
    n, m = len(query), len(target)
    D = [[0] * (m + 1) for _ in range(n + 1)]
    for i in range(1, n + 1):
        D[i][0] = i
    for j in range(1, m + 1):
        D[0][j] = j
    for i in range(1, n + 1):
        qi, Di, Dp = query[i - 1], D[i], D[i - 1]
        for j in range(1, m + 1):
            best = Dp[j - 1] + (qi != target[j - 1])
            if Dp[j] + 1 < best:
                best = Dp[j] + 1
            if Di[j - 1] + 1 < best:
                best = Di[j - 1] + 1
            Di[j] = best
    ops = []
    i, j = n, m
    while i > 0 or j > 0:
        if i > 0 and D[i - 1][j] + 1 == D[i][j]:
            ops.append("I"); i -= 1
        elif j > 0 and D[i][j - 1] + 1 == D[i][j]:
            ops.append("D"); j -= 1
        else:
            ops.append("=" if query[i - 1] == target[j - 1] else "X"); i -= 1; j -= 1
    ops.reverse()
    return ops


def min_ed(max_insertion, q_ins):
    """functions.py:771-799: thread q_ins into max_insertion along an optimal alignment that deletes nothing from
    max_insertion; "" if the reported optimum needs such a deletion."""
    ops = nw_path_cigar(max_insertion, q_ins)
    if "D" in ops:
        return ""
    out, k = [], 0
    for op in ops:
        if op == "I":
            out.append("-")
        else:
            out.append(q_ins[k]); k += 1
    return "".join(out)


def get_best_solution(max_insertion, q_ins):
    """functions.py:635-676: the columns of q_ins inside the (padded) longest insertion of its slot."""
    L = len(max_insertion)
    if q_ins == "-":
        return ["-"] * L
    pos = max_insertion.find(q_ins)
    if pos >= 0:
        return list("-" * pos + max_insertion[pos:pos + len(q_ins)] + "-" * (L - pos - len(q_ins)))
    threaded = min_ed(max_insertion, q_ins)
    if threaded:
        return list(threaded)
    max_p, max_matches = 0, 0
    for p in range(0, L - len(q_ins) + 1):
        nr = sum(1 for c1, c2 in zip(q_ins, max_insertion[p:p + len(q_ins)]) if c1 == c2)
        if nr > max_matches:
            max_p, max_matches = p, nr
    if max_p > 0:
        return list("-" * max_p + q_ins + "-" * (L - max_p - len(q_ins)))
    return [q_ins[p] if p < len(q_ins) else "-" for p in range(L)]


def position_query_to_alignment(query_aligned, target_aligned, target_alignment_start_position):
    """functions.py:598-631: (list of 2*len(target)+1 entries: insertion string or '-' on even, aligned character on odd
    positions; first vector position; last vector position)."""
    out, ins, t = [], [], target_alignment_start_position
    for qc, tc in zip(query_aligned, target_aligned):
        if tc == "-":
            ins.append(qc)
        else:
            out.append("".join(ins) if ins else "-")
            ins = []
            out.append(qc)
            t += 1
    out.append("".join(ins) if ins else "-")
    return out, 2 * target_alignment_start_position, 2 * (t - 1) + 2


def msa_matrix(m, partition):
    """(keys, M): keys = list(partition) and M = uint8 [len(keys), columns] multi-alignment matrix (ASCII, '-' = 45)
    with the reference's column layout (see above).  All rows are processed at once: the gapped strings are joined
    into two byte arrays, the centre's non-gap columns give the aligned characters of every read by one boolean
    index, and only the inserted characters (a percent of the cells) are looked at individually."""
    keys = list(partition)
    nr, Lm = len(keys), len(m)
    m_all = np.frombuffer("".join(partition[s][1] for s in keys).encode(), dtype=np.uint8)
    s_all = np.frombuffer("".join(partition[s][2] for s in keys).encode(), dtype=np.uint8)
    aln_len = np.fromiter((len(partition[s][1]) for s in keys), dtype=np.int64, count=nr)
    if len(m_all) != len(s_all) or int(aln_len.sum()) != len(m_all):
        raise ValueError("gapped strings of a pair differ in length")
    tmask = m_all != 45
    A = s_all[tmask]
    if len(A) != nr * Lm:
        raise ValueError("alignment does not spell the centre")
    A = A.reshape(nr, Lm)                                  # character aligned to every centre base
    # inserted characters: position in the joined array, row, slot (= centre bases of the row in front of it)
    ins_pos = np.flatnonzero(~tmask)
    row_start = np.zeros(nr + 1, dtype=np.int64)
    np.cumsum(aln_len, out=row_start[1:])
    ins_row = np.searchsorted(row_start, ins_pos, side="right") - 1
    ins_slot = np.cumsum(tmask)[ins_pos] - ins_row * Lm
    # runs of inserted characters = one insertion string each (same row, same slot, consecutive positions)
    if len(ins_pos):
        new_run = np.ones(len(ins_pos), dtype=bool)
        new_run[1:] = (np.diff(ins_pos) != 1) | (np.diff(ins_row) != 0) | (np.diff(ins_slot) != 0)
        first = np.flatnonzero(new_run)
        run_len = np.diff(np.append(first, len(ins_pos)))
        run_row, run_slot, run_pos = ins_row[first], ins_slot[first], ins_pos[first]
    else:
        run_len = run_row = run_slot = run_pos = np.zeros(0, dtype=np.int64)
    # slot widths: 1, or longest + 2 where some read inserts >= 2 characters (functions.py:722-731)
    width = np.ones(Lm + 1, dtype=np.int64)
    longest = np.zeros(Lm + 1, dtype=np.int64)
    np.maximum.at(longest, run_slot, run_len)
    wide_slots = np.flatnonzero(longest > 1)
    width[wide_slots] = longest[wide_slots] + 2
    col_slot = np.zeros(Lm + 1, dtype=np.int64)           # first column of slot t; the base column of t follows the slot
    col_slot[1:] = np.cumsum(width[:-1] + 1)
    ncols = int(width.sum()) + Lm
    M = np.full((nr, ncols), 45, dtype=np.uint8)
    M[:, col_slot[:Lm] + width[:Lm]] = A
    if len(run_len):
        in_wide = longest[run_slot] > 1
        # single characters in 1-column slots: one scatter
        one = ~in_wide
        M[run_row[one], col_slot[run_slot[one]]] = s_all[run_pos[one]]
        # slots with a multi-character insertion: place every insertion inside the padded longest one
        if in_wide.any():
            s_bytes = s_all.tobytes()
            by_slot = {}
            for r, t, p0, ln in zip(run_row[in_wide].tolist(), run_slot[in_wide].tolist(), run_pos[in_wide].tolist(), run_len[in_wide].tolist()):
                by_slot.setdefault(t, []).append((r, s_bytes[p0:p0 + ln].decode()))
            cache = {}
            for t, items in by_slot.items():
                lg = int(longest[t])
                mx = "-" + sorted(x for _, x in items if len(x) == lg)[0] + "-"
                c0 = int(col_slot[t])
                for r, ins in items:
                    sol = cache.get((mx, ins))
                    if sol is None:
                        sol = cache[(mx, ins)] = np.frombuffer("".join(get_best_solution(mx, ins)).encode(), dtype=np.uint8)
                    M[r, c0:c0 + len(mx)] = sol
    return keys, M


def create_multialignment_matrix(m, partition):
    """functions.py:543-588: {sequence: [column characters]} for the partition {s: (ed, m_alignment, s_alignment, degree)}."""
    keys, M = msa_matrix(m, partition)
    return {s: list(M[r].tobytes().decode()) for r, s in enumerate(keys)}


def create_position_frequency_matrix(alignment_matrix, partition):
    """functions.py:526-536: per column {'A','C','G','T','-'} -> summed degrees."""
    nr_columns = len(alignment_matrix[next(iter(alignment_matrix))])
    PFM = [{"A": 0, "C": 0, "G": 0, "T": 0, "-": 0} for _ in range(nr_columns)]
    for s, row in alignment_matrix.items():
        deg = partition[s][3]
        for j, nucl in enumerate(row):
            PFM[j][nucl] += deg
    return PFM
