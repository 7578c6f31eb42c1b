"""Consensus correction of the reads of a partition towards its centre.

Mirror of /root/reference/modules/correction_module.py:12-76 (`correct_strings`) and :260-446
(`correct_to_consensus`): multi-alignment of the partition (isocon_amd.functions.msa_matrix), position frequency
matrix, majority character per column, and for every non-converged read the correction of the ceil(half) of its
unambiguous minority positions that are rarest for their error class.  Same signatures and return shapes; the quality-
value variant (`correct_to_consensus_ccs_qual`, switched off in the reference: isocon_get_candidates.py:106 `if False`)
is not provided.  Column statistics are numpy reductions over the uint8 alignment matrix instead of Python loops over
dicts of lists; every tie rule of the reference is kept (first maximum in the order A, C, G, T, -; stable sort of the
candidate positions by frequency; ties with the last chosen frequency are corrected too)."""
from __future__ import annotations

import math

import numpy as np

from .functions import msa_matrix

_SYMS = np.frombuffer(b"ACGT-", dtype=np.uint8)


def correct_to_consensus(m, partition, seq_to_acc, step, verbose):
    """correction_module.py:260-446.  partition: {s: (edit_distance, m_alignment, s_alignment, degree)} incl. the centre m.
    Returns {accession: corrected sequence} for the reads that changed... (every non-converged read with at least one
    correctable position, as in the reference, even if the corrected string equals the old one)."""
    S_prime_partition = {}
    N_t = sum(t[3] for t in partition.values())
    if not (len(partition) > 1 and N_t > 2):
        return S_prime_partition
    keys, M = msa_matrix(m, partition)
    nr, ncols = M.shape
    if M[M != 45].tobytes() != "".join(keys).encode():             # correction_module.py:273-275, all rows at once
        raise AssertionError("multi-alignment rows do not spell their sequences")
    deg = np.array([partition[s][3] for s in keys], dtype=np.int64)
    sym_index0 = np.full(256, -1, dtype=np.int64)
    sym_index0[_SYMS] = np.arange(5)
    heavy = np.flatnonzero(deg != 1)                               # rows of multiplicity > 1 (the centre, usually) count extra
    counts = np.stack([np.count_nonzero(M == c, axis=0) for c in _SYMS]).astype(np.int64)     # [5, ncols], order A C G T -
    for r in heavy:
        counts[sym_index0[M[r]], np.arange(ncols)] += deg[r] - 1
    maj_idx = counts.argmax(axis=0)                                # first maximum in that order (max() over the dict)
    maj_cnt = counts.max(axis=0)
    unambiguous = (counts == maj_cnt[None, :]).sum(axis=0) == 1
    maj_chr = _SYMS[maj_idx]
    maj_is_gap = maj_idx == 4
    # error-type totals over the unambiguous columns (correction_module.py:296-307)
    tot = counts.sum(axis=0)
    c_ins = int((tot - maj_cnt)[unambiguous & maj_is_gap].sum())
    col_ok = unambiguous & ~maj_is_gap
    c_del = int(counts[4][col_ok].sum())
    c_subs = int((tot - maj_cnt - counts[4])[col_ok].sum())
    sym_index = np.full(256, -1, dtype=np.int64)
    sym_index[_SYMS] = np.arange(5)

    # Per read: the unambiguous columns where it differs from the majority are its correctable positions; ceil(half) of
    # them are corrected, rarest first (frequency of the read's character in the column relative to the partition's
    # total of that error class), plus every position tied with the last one chosen (correction_module.py:329-402).
    # Equivalent set form used here: positions whose frequency is <= the ceil(n/2)-th smallest of the read.
    single = deg == 1
    cand = (M != maj_chr[None, :]) & unambiguous[None, :] & single[:, None]
    rows, cols = np.nonzero(cand)                                   # row-major: ascending row, then ascending column
    new_M = M
    n_cand = np.bincount(rows, minlength=nr)
    if len(rows):
        v = M[rows, cols]
        own_cnt = counts[sym_index[v], cols].astype(np.float64)
        denom = np.where(maj_is_gap[cols], float(max(c_ins, 1)), np.where(v == 45, float(max(c_del, 1)), float(max(c_subs, 1))))
        freq = own_cnt / denom
        srt = np.lexsort((freq, rows))                              # by row, then frequency (stable)
        start = np.zeros(nr + 1, dtype=np.int64)
        np.cumsum(n_cand, out=start[1:])
        k = (n_cand + 1) // 2                                       # ceil(n / 2)
        thr = np.full(nr, -1.0)
        has = k > 0
        thr[has] = freq[srt[start[:-1][has] + k[has] - 1]]
        chosen = freq <= thr[rows]
        new_M = M.copy()
        new_M[rows[chosen], cols[chosen]] = maj_chr[cols[chosen]]
    keep = new_M != 45
    lens_new = keep.sum(axis=1)
    flat = new_M[keep].tobytes().decode()
    off = np.zeros(nr + 1, dtype=np.int64)
    np.cumsum(lens_new, out=off[1:])
    for r in sorted(range(nr), key=lambda r: keys[r]):
        if not single[r] or n_cand[r] == 0:
            continue
        s_modified = flat[off[r]:off[r + 1]]
        for acc in seq_to_acc[keys[r]]:
            S_prime_partition[acc] = s_modified
    return S_prime_partition


def correct_strings(partition_alignments, seq_to_acc, ccs_dict, step, nr_cores=1, verbose=False):
    """correction_module.py:12-76.  partition_alignments: {centre: {s: (ed, aln_centre, aln_s, degree)}};
    seq_to_acc: {sequence: [accessions]}.  Returns (S_prime, S_prime_quality) -- the second is always {} here."""
    if ccs_dict:
        raise NotImplementedError("correction with CCS quality values (disabled in the reference, isocon_get_candidates.py:106)")
    S_prime = {}
    for m, partition in sorted(partition_alignments.items()):
        acc_of = {m: seq_to_acc[m]}
        for s in partition:
            if s in seq_to_acc:
                acc_of[s] = seq_to_acc[s]
        for acc, s in correct_to_consensus(m, partition, acc_of, step, verbose).items():
            assert acc not in S_prime
            S_prime[acc] = s
    return S_prime, {}
