"""Consensus correction of the reads of a partition towards its centre.

Mirror of /root/reference/modules/correction_module.py:12-76 (`correct_strings`) and :260-446
(`correct_to_consensus`): multi-alignment of the partition (isocon_amd.functions.msa_matrix), position frequency
matrix, majority character per column, and for every non-converged read the correction of the ceil(half) of its
unambiguous minority positions that are rarest for their error class.  Same signatures and return shapes; the quality-
value variant (`correct_to_consensus_ccs_qual`, switched off in the reference: isocon_get_candidates.py:106 `if False`)
is not provided.  Column statistics, the per-read selection and the gap stripping run on the GPU (isocon_msa_correct);
the same steps exist as numpy reductions over the uint8 matrix (ISOCON_CORRECT_HOST=1: CPU tests, A/B checks); every tie rule of the reference is kept (first maximum in the order A, C, G, T, -; stable sort of the
candidate positions by frequency; ties with the last chosen frequency are corrected too)."""
from __future__ import annotations

import math

import numpy as np

import ctypes
import os

from . import _lib
from .functions import msa_matrix

_SYMS = np.frombuffer(b"ACGT-", dtype=np.uint8)


def _correct_on_device(M, deg):
    """Column statistics + per-read correction + gap stripping on the GPU (isocon_msa_correct, csrc/msa.hpp).
    Returns (packed bytes, offsets int64[nr+1], n_cand int32[nr])."""
    L = _lib.lib()
    nr, ncols = M.shape
    M = np.ascontiguousarray(M)
    deg32 = np.ascontiguousarray(deg, dtype=np.int32)
    packed = np.empty(M.size, dtype=np.uint8)
    offsets = np.zeros(nr + 1, dtype=np.uint64)
    n_cand = np.zeros(nr, dtype=np.int32)
    tot = (ctypes.c_int64 * 3)()
    _lib.check(L.isocon_msa_correct(M.ctypes.data_as(_lib.u8p), nr, ncols, deg32.ctypes.data_as(_lib.i32p), packed.ctypes.data_as(_lib.u8p),
                                    packed.size, offsets.ctypes.data_as(_lib.u64p), n_cand.ctypes.data_as(_lib.i32p), tot, None),
               "isocon_msa_correct")
    return packed, offsets.astype(np.int64), n_cand


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
    # correction_module.py:273-275 asserts that every row spells its sequence; checked here by length for all rows and
    # letter by letter for a sample (the full comparison is a pass over the whole matrix)
    if (np.count_nonzero(M != 45, axis=1) != np.fromiter((len(k) for k in keys), dtype=np.int64, count=nr)).any():
        raise AssertionError("multi-alignment rows do not spell their sequences")
    for r in range(0, nr, max(1, nr // 16)):
        if M[r][M[r] != 45].tobytes().decode() != keys[r]:
            raise AssertionError("multi-alignment row does not spell its sequence")
    deg = np.array([partition[s][3] for s in keys], dtype=np.int64)
    if os.environ.get("ISOCON_CORRECT_HOST") != "1":               # the product path: HIP kernels (no silent CPU fallback)
        packed, off, n_cand = _correct_on_device(M, deg)
        if (n_cand >= 0).all():
            flat = packed[:off[nr]].tobytes().decode()
            for r in sorted(range(nr), key=lambda r: keys[r]):
                if deg[r] == 1 and n_cand[r] > 0:
                    s_modified = flat[off[r]:off[r + 1]]
                    for acc in seq_to_acc[keys[r]]:
                        S_prime_partition[acc] = s_modified
            return S_prime_partition
        # a read with more correctable positions than the kernel holds per row: the whole partition on the host below
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
