"""TEST INFRASTRUCTURE ONLY -- numpy checker for the device part of the consensus correction (isocon_msa_correct).

Restates /root/reference/modules/correction_module.py:283-402 on the finished multi-alignment matrix: position frequency
matrix with multiplicities (:283, functions.py:526-536), majority per column = first maximum in the order A, C, G, T, -
(:288-294), error-class totals over the unambiguous columns (:296-307), per read the correctable positions, their
frequencies, the ceil(half) rarest of them plus ties (:329-402), and the corrected row without gaps (:404-410).
Pinned through tests/golden/g11_correction.json (outputs of the reference's own correct_strings).  Imported by tests/
only (tests substitute isocon_amd.correction_module._correct_on_device with `correct_rows` where no GPU exists, and
compare the HIP kernels against it where one does); the product never imports it."""
from __future__ import annotations

import numpy as np

_SYMS = np.frombuffer(b"ACGT-", dtype=np.uint8)


def correct_rows(M, deg):
    """M: uint8 [nr, ncols] of 'A','C','G','T','-'; deg: multiplicity per row.
    Returns (packed bytes uint8, offsets int64[nr+1], n_cand int32[nr]) like isocon_msa_correct."""
    M = np.ascontiguousarray(M)
    deg = np.asarray(deg, dtype=np.int64)
    nr, ncols = M.shape
    sym_index = np.full(256, -1, dtype=np.int64)
    sym_index[_SYMS] = np.arange(5)
    heavy = np.flatnonzero(deg != 1)                               # rows of multiplicity > 1 (the centre, usually) count extra
    counts = np.stack([np.count_nonzero(M == c, axis=0) for c in _SYMS]).astype(np.int64)     # [5, ncols], order A C G T -
    for r in heavy:
        counts[sym_index[M[r]], np.arange(ncols)] += deg[r] - 1
    maj_idx = counts.argmax(axis=0)                                # first maximum in that order (max() over the dict)
    maj_cnt = counts.max(axis=0)
    unambiguous = (counts == maj_cnt[None, :]).sum(axis=0) == 1
    maj_chr = _SYMS[maj_idx]
    maj_is_gap = maj_idx == 4
    tot = counts.sum(axis=0)
    c_ins = int((tot - maj_cnt)[unambiguous & maj_is_gap].sum())
    col_ok = unambiguous & ~maj_is_gap
    c_del = int(counts[4][col_ok].sum())
    c_subs = int((tot - maj_cnt - counts[4])[col_ok].sum())

    # Per read: the unambiguous columns where it differs from the majority are its correctable positions; ceil(half) of
    # them are corrected, rarest first (frequency of the read's character in the column relative to the partition's
    # total of that error class), plus every position tied with the last one chosen.  Set form: positions whose
    # frequency is <= the ceil(n/2)-th smallest of the read.
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
    off = np.zeros(nr + 1, dtype=np.int64)
    np.cumsum(keep.sum(axis=1), out=off[1:])
    return new_M[keep], off, n_cand.astype(np.int32)
