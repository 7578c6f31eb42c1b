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
        ops_list = [SW_alignment_module.ops_of(pairwise_alignments[keys[i][0]][keys[i][1]]) for i in with_ops]
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


# ---- helpers of the statistical-test phase (callers: hypothesis_test_module, isocon_statistical_test) ----
def transform(read):
    """functions.py:53-61: homopolymer compression (every run of equal characters becomes one character)."""
    return "".join(c for i, c in enumerate(read) if i == 0 or c != read[i - 1])


def get_homopolymer_invariants(candidate_transcripts):
    """functions.py:63-86: {acc: {other acc: 1}} over candidates that are equal after homopolymer compression."""
    clusters = {}
    for acc, seq in candidate_transcripts.items():
        clusters.setdefault(transform(seq), []).append(acc)
    edges = {}
    for members in clusters.values():
        if len(members) > 1:
            for acc in members:
                edges[acc] = {}
            for a in members:
                for b in members:
                    if a is not b:
                        edges[a][b] = 1
    return edges


def _run_length(s, ch):
    """length of the run of `ch` the string starts with"""
    n = 0
    while n < len(s) and s[n] == ch:
        n += 1
    return n


def get_variant_coordinates(t_seq, c_seq, aln_t, aln_c, variants):
    """functions.py:89-146.  For every differing alignment column (i, p_t, p_c): its coordinate on the reference t and
    on the candidate c, the variant type seen from c ('S', 'I' = c has an extra base, 'D' = c lacks one), the number u_v
    of equivalent placements inside a homopolymer of t, and the alignment snippets around it.  Returns
    (variant_coords_t, variant_coords_c, alignment_c_to_t, alignment_t_to_c); later variants overwrite earlier ones that
    land on the same coordinate, as in the reference."""
    variant_coords_t, variant_coords_c, alignment_c_to_t, alignment_t_to_c = {}, {}, {}, {}
    for (i, p_t, p_c) in variants:
        # last base of t / c at or before column i (sequence characters in aln[:i + 1], minus one)
        t_last, c_last = i - aln_t.count("-", 0, i + 1), i - aln_c.count("-", 0, i + 1)
        if p_c == "-":                                          # the candidate lacks a base of t
            v = t_seq[t_last]
            fwd = _run_length(t_seq[t_last + 1:], v)
            back = _run_length(t_seq[t_last::-1], v)            # includes t_last itself
            u_v = fwd + back if (fwd and back) else (fwd or back or 1)
            entry = ("D", "-", u_v)
            variant_coords_t[t_last] = entry
            variant_coords_c[c_last + 1] = entry                # the base right of the deletion carries it on c
            alignment_c_to_t[t_last] = aln_c[max(0, i - 1): i + u_v + 1]
            alignment_t_to_c[c_last + 1] = aln_t[max(0, i - 1): i + u_v + 1]
        elif p_t == "-":                                        # the candidate has an extra base
            v = c_seq[c_last]
            fwd = _run_length(t_seq[t_last + 1:], v)
            back = _run_length(t_seq[t_last::-1], v)            # t_last == -1 slices from the END of t, as in the reference
            # x + 1 ways to insert one more character into a homopolymer of x characters
            u_v = fwd + back + 1 if (fwd and back) else (fwd + 1 if fwd else (back + 1 if back else 1))
            entry = ("I", p_c, u_v)
            variant_coords_t[t_last + 1] = entry
            variant_coords_c[c_last] = entry
            alignment_c_to_t[t_last + 1] = aln_c[max(0, i - 1): i + u_v + 1]
            alignment_t_to_c[c_last] = aln_t[max(0, i - 1): i + u_v + 1]
        else:
            entry = ("S", p_c, 1)
            variant_coords_t[t_last] = entry
            variant_coords_c[c_last] = entry
            alignment_c_to_t[t_last] = aln_c[max(0, i - 1): i + 2]
            alignment_t_to_c[c_last] = aln_t[max(0, i - 1): i + 2]
    return variant_coords_t, variant_coords_c, alignment_c_to_t, alignment_t_to_c


def get_support(read_alignments_to_c, variant_coords_c, read_alignments_to_t, variant_coords_t, alignment_c_to_t):
    """functions.py:149-201: the reads that carry every variant of the candidate.  A read aligned to c must agree with
    c's alignment row over the variant and one column either side (u_v columns for a homopolymer); a read aligned to t
    must show c's snippet there (shifted by one for an insertion, whose coordinate on t is the base to its right).
    Returns the accessions: supporters among c's reads, then supporters among t's reads."""
    supporters = []
    for read_acc, (aln_c, aln_read, _) in read_alignments_to_c.items():
        col_of = [j for j, ch in enumerate(aln_c) if ch != "-"]
        for i, (_, _, u_v) in variant_coords_c.items():
            pos = col_of[i]
            lo, hi = max(0, pos - 1), pos + u_v + 1
            if aln_read[lo:hi] != aln_c[lo:hi]:
                break
        else:
            supporters.append(read_acc)
    from_t = []
    for read_acc, (aln_t, aln_read, _) in read_alignments_to_t.items():
        col_of = [j for j, ch in enumerate(aln_t) if ch != "-"]
        for i, (v_type, _, u_v) in variant_coords_t.items():
            pos = col_of[i]
            window = aln_read[max(0, pos - 2): pos + u_v] if v_type == "I" else aln_read[max(0, pos - 1): pos + u_v + 1]
            if window != alignment_c_to_t[i]:
                break
        else:
            from_t.append(read_acc)
    return supporters + from_t


def read_errors_from_alignment(ref_aln, read_aln):
    """functions.py:495-522: (insertions, deletions, substitutions) of the read, end gaps of either row excluded."""
    n = len(ref_aln)
    start = max(n - len(ref_aln.lstrip("-")), len(read_aln) - len(read_aln.lstrip("-")))
    stop = n - max(n - len(ref_aln.rstrip("-")), len(read_aln) - len(read_aln.rstrip("-")))
    ins = dele = sub = 0
    for a, b in zip(ref_aln[start:stop], read_aln[start:stop]):
        if a != b:
            if a == "-":
                ins += 1
            elif b == "-":
                dele += 1
            else:
                sub += 1
    return ins, dele, sub


def get_read_errors(read_alignments_to_c, read_alignments_to_t):
    """functions.py:204-216: {read: (insertions, deletions, substitutions)}, t's reads first."""
    errors = {}
    for read_acc, (aln_ref, aln_read, _) in read_alignments_to_t.items():
        errors[read_acc] = read_errors_from_alignment(aln_ref, aln_read)
    for read_acc, (aln_ref, aln_read, _) in read_alignments_to_c.items():
        errors[read_acc] = read_errors_from_alignment(aln_ref, aln_read)
    return errors


def get_empirical_error_probabilities(segment_length, errors, variant_coords_t):
    """functions.py:435-466: per read the probability of producing all variants by sequencing error, from its own error
    counts (never below one error per class: p = 0 is not allowed), uniform over positions and, for substitutions and
    insertions, over the 3 / 4 possible characters; homopolymer multiplicity u_v; indel factors capped at 0.5."""
    delta_size = float(len(variant_coords_t))
    assert delta_size > 0.0
    probability = {}
    for read_acc, (insertions, deletions, substitutions) in errors.items():
        p_S = (max(substitutions, delta_size) / float(segment_length)) / 3.0
        p_I = (max(insertions, delta_size) / float(segment_length)) / 4.0
        p_D = (max(deletions, delta_size) / float(segment_length))
        prob = 1.0
        for v_type, _, u_v in variant_coords_t.values():
            if v_type == "S":
                prob *= p_S * u_v
            elif v_type == "I":
                prob *= min(0.5, p_I * u_v)
            elif v_type == "D":
                prob *= min(0.5, p_D * u_v)
        if prob >= 1.0:
            prob = 0.99999
        probability[read_acc] = prob
    return probability


def choose(n, k):
    """functions.py:479-492: binomial coefficient, 0 outside 0 <= k <= n."""
    import math
    return math.comb(n, k) if 0 <= k <= n else 0


def _ccs_probabilities(read_alignments, variant_coords, other_snippets, ccs_dict, errors, max_phred_q_trusted, shifted_type, coord_when_other):
    """Shared body of get_read_ccs_probabilities_c / _t (functions.py:240-331, :334-433).  For every read aligned to one of
    the two sequences ("own"): at every variant the read must show either its own sequence's row or the other sequence's
    snippet exactly (else the read is not informative); the base quality at that place -- remapped from [3, 93] to
    [3, max_phred_q_trusted] -- gives the probability of the variant arising by a base-call error, split over the error
    classes by the partition's overall ratios (a homopolymer variant takes the whole base-call uncertainty).
    shifted_type: the variant type whose coordinate on "own" is the base to its right (snippet window shifted by one);
    coord_when_other: {type: offset} of the read coordinate relative to the bases seen up to the variant column when the
    read shows the OTHER sequence."""
    subs = float(max(1.0, sum([s for i, d, s in errors.values()])))
    ins = float(max(1.0, sum([i for i, d, s in errors.values()])))
    del_ = float(max(1.0, sum([d for i, d, s in errors.values()])))
    tot_errors = subs + ins + del_
    subs_ratio, ins_ratio, del_ratio = subs / tot_errors, ins / tot_errors, del_ / tot_errors
    assert len(variant_coords) > 0
    probabilities = {}
    not_supporting = set()
    for read_acc, (aln_own, aln_read, _) in read_alignments.items():
        col_of = [j for j, ch in enumerate(aln_own) if ch != "-"]
        prob = 1.0
        for i, (v_type, _, u_v) in variant_coords.items():
            pos = col_of[i]
            lo, hi = max(0, pos - 1), pos + u_v + 1
            shows_own = aln_read[lo:hi] == aln_own[lo:hi]
            if v_type == shifted_type:
                shows_other = aln_read[max(0, pos - 2): pos + u_v] == other_snippets[i]
            else:
                shows_other = aln_read[lo:hi] == other_snippets[i]
            assert not (shows_own and shows_other)
            seen = pos + 1 - aln_read.count("-", 0, pos + 1)           # read bases in aln_read[:pos + 1]
            if shows_own:
                read_coord = seen - 1
            elif shows_other:
                read_coord = seen + coord_when_other.get(v_type, -1)
            else:
                not_supporting.add(read_acc)
                prob = -1
                break
            record = ccs_dict[read_acc]
            q_qual = record.qual[record.read_aln_to_ccs_coord(aln_read, read_coord)]
            q_qual_mapped = (q_qual - 3) * (max_phred_q_trusted - 3.0) / (90.0) + 3
            if u_v > 1:
                p_error = (10 ** (-q_qual_mapped / 10.0))
            elif v_type == "S":
                p_error = ((10 ** (-q_qual_mapped / 10.0)) * subs_ratio) / 3.0
            elif v_type == "I":
                p_error = ((10 ** (-q_qual_mapped / 10.0)) * ins_ratio) / 4.0
            else:
                p_error = (10 ** (-q_qual_mapped / 10.0)) * del_ratio
            prob *= p_error
        if prob >= 0:
            assert 0.0 < prob < 1.0
            probabilities[read_acc] = prob
    return probabilities, not_supporting


def get_read_ccs_probabilities_c(read_alignments_to_c, variant_coords_c, alignment_t_to_c, ccs_dict, errors, max_phred_q_trusted):
    """functions.py:240-331: reads aligned to the candidate.  A deletion sits on the base to its right on c; a read that
    shows t at an insertion is judged on the base after the variant column."""
    return _ccs_probabilities(read_alignments_to_c, variant_coords_c, alignment_t_to_c, ccs_dict, errors, max_phred_q_trusted, "D", {"I": 0})


def get_read_ccs_probabilities_t(read_alignments_to_t, variant_coords_t, alignment_c_to_t, ccs_dict, errors, max_phred_q_trusted):
    """functions.py:334-433: reads aligned to the reference.  An insertion sits on the base to its right on t; a read that
    shows c is judged on the next base at a deletion and two bases back at an insertion."""
    return _ccs_probabilities(read_alignments_to_t, variant_coords_t, alignment_c_to_t, ccs_dict, errors, max_phred_q_trusted, "I", {"D": 0, "I": -2})
