"""Consensus correction of the reads of a partition towards its centre.

Mirror of /root/reference/modules/correction_module.py:12-76 (`correct_strings`) and :260-446
(`correct_to_consensus`): multi-alignment of the partition (isocon_amd.functions.msa_matrix), position frequency
matrix, majority character per column, and for every non-converged read the correction of the ceil(half) of its
unambiguous minority positions that are rarest for their error class.  Same signatures and return shapes; the quality-
value variant (`correct_to_consensus_ccs_qual`, switched off in the reference: isocon_get_candidates.py:106 `if False`)
is not provided.  Column statistics, the per-read selection and the gap stripping run on the GPU (isocon_msa_correct,
csrc/msa.hpp) -- there is no host path: without the library or a GPU this module raises.  Every tie rule of the
reference is kept (first maximum in the order A, C, G, T, -; stable sort of the candidate positions by frequency; ties
with the last chosen frequency are corrected too).  The CPU tests substitute `_correct_on_device` with the numpy
checker of oracle/correction.py."""
from __future__ import annotations

import numpy as np

import ctypes

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
    # the product path: HIP kernels only (no host restatement here; the numpy checker lives in oracle/correction.py)
    packed, off, n_cand = _correct_on_device(M, deg)
    if (n_cand < 0).any():
        raise RuntimeError("isocon_msa_correct left rows unprocessed")
    flat = packed[:off[nr]].tobytes().decode()
    for r in sorted(range(nr), key=lambda r: keys[r]):
        if deg[r] == 1 and n_cand[r] > 0:
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
