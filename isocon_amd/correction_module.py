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


def _partition_rows(batch, m):
    """(row ids, ops, ops_ptr) of the partition of centre m for isocon_msa_build_ops: row 0 the centre, then its surviving members"""
    rows = batch.rows_of.get(m, [])
    nr = 1 + len(rows)
    idx = np.asarray(rows, dtype=np.int64)
    row_ids = np.concatenate([batch.a[idx[:1]], batch.b[idx]]).astype(np.uint32)
    cnt = (batch.ops_ptr[idx + 1] - batch.ops_ptr[idx]).astype(np.int64)
    ops_ptr = np.zeros(nr + 1, dtype=np.uint64)
    np.cumsum(cnt, out=ops_ptr[2:])
    # the rows' ops, gathered: pairs of one centre are consecutive in the batch unless the exon filter removed some
    starts = batch.ops_ptr[idx]
    if len(idx) and int(starts[-1] + cnt[-1] - starts[0]) == int(cnt.sum()):
        ops = batch.ops[int(starts[0]):int(starts[0]) + int(cnt.sum())]
    else:
        ops = np.concatenate([batch.ops[int(b0):int(b0 + c)] for b0, c in zip(starts.tolist(), cnt.tolist())]) if len(idx) else np.zeros(0, np.uint32)
    return row_ids, ops, ops_ptr


_CODE = np.frombuffer(b"ACGT", dtype=np.uint8)
_PLACEMENT_CACHE = {}          # (padded longest insertion, insertion) -> bytes of get_best_solution: the same few short strings recur in every
#                                partition and iteration (a function of its two arguments only); bounded below


def _wide_slot_patches(members, wide, col_slot, longest):
    """Where the insertions of the wide slots sit (functions.py:722-767): every insertion of a slot whose longest insertion has 2 or
    more characters is placed inside "-" + the (alphabetically first) longest one + "-" by get_best_solution.  wide: the records of
    isocon_msa_build_ops (row, slot, position in the member, length, 2-bit codes of the first 32 bases).  Tens of reads insert the
    same base or two at a wide slot: the placement is computed once per DISTINCT (slot, insertion) -- a numpy unique over the records --
    and scattered to the rows.  Returns (patch_row, patch_col, patch_ptr, patch_bytes) or four None."""
    from .functions import get_best_solution
    if not len(wide):
        return None, None, None, None
    wide = np.asarray(wide, dtype=np.uint32)
    slot, ln = wide[:, 1].astype(np.int64), wide[:, 3].astype(np.int64)
    codes = wide[:, 4].astype(np.uint64) | (wide[:, 5].astype(np.uint64) << np.uint64(32))
    # distinct (slot, length, codes[, row for the rare insertions longer than the 32 coded bases])
    long_ins = ln > 32
    tie = np.where(long_ins, np.arange(len(wide), dtype=np.int64) + 1, 0)
    k1, k2 = (slot << 32) | ln, codes.astype(np.int64)
    if int(ln.max()) <= 20 and int(slot.max()) < (1 << 17):
        # the usual case: slot, length and the bases in ONE 63-bit key (17 + 6 + 40 bits): one sort instead of a three-key lexsort
        order = np.argsort((slot << 46) | (ln << 40) | (k2 & ((1 << 40) - 1)), kind="stable")
    else:
        order = np.lexsort((tie, k2, k1))
    new = np.ones(len(order), dtype=bool)
    new[1:] = (np.diff(k1[order]) != 0) | (np.diff(k2[order]) != 0) | (np.diff(tie[order]) != 0)
    first = order[new]
    inv = np.empty(len(order), dtype=np.int64)
    inv[order] = np.cumsum(new) - 1
    u_slot, u_len = slot[first], ln[first]
    # the distinct insertion strings: the 2-bit codes of all of them decoded in one numpy pass, cut out of one str
    n_u = len(first)
    u_codes = codes[first]
    chars = _CODE[((u_codes[:, None] >> (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]) & np.uint64(3)).astype(np.intp)]
    flat = chars.tobytes().decode()
    first_l, u_len_l = first.tolist(), u_len.tolist()
    strings = [flat[32 * i:32 * i + n_b] if n_b <= 32 else None for i, n_b in enumerate(u_len_l)]
    for i, n_b in enumerate(u_len_l):
        if n_b > 32:          # (rare: an insertion longer than the coded bases is read from the member itself)
            f = first_l[i]
            r, sp = int(wide[f, 0]), int(wide[f, 2])
            strings[i] = members[r - 1][sp:sp + n_b]
    # per slot: the padded longest insertion (functions.py:722-731), then every distinct insertion inside it.  `first` is ordered by slot, so
    # a slot's distinct insertions are a run of it
    width = longest.astype(np.int64) + 2
    sol_len = width[u_slot]
    sol_off = np.zeros(n_u + 1, dtype=np.int64)
    np.cumsum(sol_len, out=sol_off[1:])
    sol_bytes = np.empty(int(sol_off[-1]), dtype=np.uint8)
    cuts = [0] + (np.flatnonzero(np.diff(u_slot)) + 1).tolist() + [n_u]
    sol_off_l = sol_off.tolist()
    longest_l = longest.tolist()
    u_slot_l = u_slot.tolist()
    cache = _PLACEMENT_CACHE
    H = _lib.pyhelp()
    place = getattr(H, "best_solution", None) if H is not None else None          # (functions.get_best_solution in the CPython helper)
    for g in range(len(cuts) - 1):
        lo, hi = cuts[g], cuts[g + 1]
        lg = longest_l[u_slot_l[lo]]
        mx = "-" + min(strings[i] for i in range(lo, hi) if u_len_l[i] == lg) + "-"
        for i in range(lo, hi):
            key = (mx, strings[i])
            sol = cache.get(key)
            if sol is None:
                if len(cache) > 200000:
                    cache.clear()
                placed = None
                if place is not None and len(mx) <= 255 and len(strings[i]) <= 255:
                    try:
                        placed = place(mx, strings[i])
                    except TypeError:
                        placed = None
                if placed is None:
                    placed = "".join(get_best_solution(mx, strings[i])).encode()
                sol = cache[key] = np.frombuffer(placed, dtype=np.uint8)
            sol_bytes[sol_off_l[i]:sol_off_l[i + 1]] = sol
    # one patch per record: the bytes of its distinct insertion's placement, at its slot's first column
    p_len = sol_len[inv]
    p_ptr = np.zeros(len(wide) + 1, dtype=np.int64)
    np.cumsum(p_len, out=p_ptr[1:])
    src = np.repeat(sol_off[inv] - p_ptr[:-1], p_len) + np.arange(int(p_ptr[-1]), dtype=np.int64)
    return wide[:, 0], col_slot[slot], p_ptr.astype(np.uint32), sol_bytes[src]


def _correct_partition_from_ops(batch, m, partition, seq_to_acc):
    """correct_to_consensus for a partition whose alignments are CIGAR ops on the device's store (isocon_get_candidates.AlignmentBatch):
    the matrix is built there (isocon_msa_build_ops), the insertions of the wide slots are placed by get_best_solution here and sent
    as patches, the correction runs on the built matrix (isocon_msa_correct_built).  Returns {accession: corrected sequence}."""
    rows = batch.rows_of.get(m, [])
    st = batch.store
    members = [batch.pairs[p][1] for p in rows]
    nr = 1 + len(rows)
    N_t = partition[m][3] + len(rows)
    out = {}
    if not (nr > 1 and N_t > 2):
        return out
    row_ids, ops, ops_ptr = _partition_rows(batch, m)
    n_cols, col_slot, longest, wide = st.msa_build_ops(row_ids, ops, ops_ptr)
    p_row, p_col, p_ptr, p_bytes = _wide_slot_patches(members, wide, col_slot, longest)
    deg = np.ones(nr, dtype=np.int32)
    deg[0] = partition[m][3]
    packed, off, n_cand = st.msa_correct_built(nr, n_cols, deg, p_row, p_col, p_ptr, p_bytes)
    if (n_cand < 0).any():
        raise RuntimeError("isocon_msa_correct left rows unprocessed")
    todo = np.flatnonzero((deg == 1) & (n_cand > 0))
    if len(todo):
        H = _lib.pyhelp()
        keys = [m] + members
        if H is not None and hasattr(H, "split_ascii"):
            # the corrected rows as str objects, cut out of the packed buffer in one call
            strs = H.split_ascii(packed.ctypes.data, np.ascontiguousarray(off, dtype=np.int64).ctypes.data, nr)
            rows_todo = todo.tolist()
            acc_lists = list(map(seq_to_acc.__getitem__, map(keys.__getitem__, rows_todo)))
            if all(map((1).__eq__, map(len, acc_lists))):          # a read that is still being corrected has multiplicity 1: one accession each
                out.update(zip(map(next, map(iter, acc_lists)), map(strs.__getitem__, rows_todo)))
            else:
                for r, accs_of_r in zip(rows_todo, acc_lists):
                    for acc in accs_of_r:
                        out[acc] = strs[r]
        else:
            flat = packed[:off[nr]].tobytes().decode()
            for r in todo.tolist():
                for acc in seq_to_acc[keys[r]]:
                    out[acc] = flat[off[r]:off[r + 1]]
    return out


def _rows_intact(partition, m, batch):
    from .isocon_get_candidates import LazyAlignment
    H = _lib.pyhelp()
    if H is not None and hasattr(H, "lazy_rows_intact"):
        return H.lazy_rows_intact(partition, m, LazyAlignment, batch, batch.pairs)
    pairs = batch.pairs
    for s, v in partition.items():
        if s is m:
            continue
        if type(v) is not LazyAlignment or v._batch is not batch or not (0 <= v._p < len(pairs)) or pairs[v._p][0] is not m or pairs[v._p][1] is not s:
            return False
    return True


def _correct_all_from_ops(batch, partition_alignments, centres, seq_to_acc):
    """correct_to_consensus for ALL the given partitions (centres: sorted list) in one batched build + correct on the device
    (isocon_msa_build_ops_batch / isocon_msa_correct_built_batch).  Returns ({accession: corrected sequence}, centres whose partition has a
    row with more correctable positions than the batched kernel keeps: those go through the single-partition path)."""
    st = batch.store
    first_row, idx_parts, keys, deg_parts = [0], [], [], []
    members_all = getattr(batch, "_members", None)
    if members_all is None:          # the member (second) sequence of every pair, once per batch
        members_all = batch._members = [pr[1] for pr in batch.pairs]
    member_of = members_all.__getitem__
    for m in centres:
        rows = batch.rows_of[m]
        idx_parts.append(np.asarray(rows, dtype=np.int64))
        first_row.append(first_row[-1] + 1 + len(rows))
        keys.append(m)
        keys.extend(map(member_of, rows))
        d = np.ones(1 + len(rows), dtype=np.int32)
        d[0] = partition_alignments[m][m][3]
        deg_parts.append(d)
    n_parts, n_rows = len(centres), first_row[-1]
    first_row = np.asarray(first_row, dtype=np.int64)
    all_idx = np.concatenate(idx_parts)
    member_row = np.ones(n_rows, dtype=bool)
    member_row[first_row[:-1]] = False                          # the centres' rows
    row_ids = np.empty(n_rows, dtype=np.uint32)
    row_ids[member_row] = batch.b[all_idx]
    row_ids[first_row[:-1]] = batch.a[np.fromiter((ix[0] for ix in idx_parts), dtype=np.int64, count=n_parts)]
    cnt = np.zeros(n_rows, dtype=np.int64)
    cnt[member_row] = batch.ops_ptr[all_idx + 1] - batch.ops_ptr[all_idx]
    ops_ptr = np.zeros(n_rows + 1, dtype=np.uint64)
    np.cumsum(cnt, out=ops_ptr[1:])
    starts = np.zeros(n_rows, dtype=np.int64)
    starts[member_row] = batch.ops_ptr[all_idx]
    total = int(ops_ptr[-1])
    src = np.repeat(starts - ops_ptr[:-1].astype(np.int64), cnt) + np.arange(total, dtype=np.int64)
    ops = batch.ops[src] if total else np.zeros(0, dtype=np.uint32)
    n_cols, slot_base, col_slot, longest, wide = st.msa_build_ops_batch(first_row, row_ids, ops, ops_ptr)
    p_row = p_col = p_ptr = p_bytes = None
    if len(wide):
        order = np.argsort(wide[:, 6], kind="stable")
        wide = wide[order]
        cut = np.flatnonzero(np.diff(wide[:, 6].astype(np.int64))) + 1
        rows_l, cols_l, lens_l, bytes_l = [], [], [], []
        for part in np.split(wide, cut):
            p = int(part[0, 6])
            r0 = int(first_row[p])
            local = part.copy()
            local[:, 0] -= r0
            members = keys[r0 + 1:int(first_row[p + 1])]
            sb = int(slot_base[p])
            pr, pc, pp, pb = _wide_slot_patches(members, local, col_slot[sb:int(slot_base[p + 1])], longest[sb:int(slot_base[p + 1])])
            rows_l.append(np.asarray(pr, dtype=np.int64) + r0); cols_l.append(np.asarray(pc, dtype=np.int64)); lens_l.append(np.diff(pp.astype(np.int64))); bytes_l.append(pb)
        p_row, p_col = np.concatenate(rows_l), np.concatenate(cols_l)
        p_ptr = np.zeros(len(p_row) + 1, dtype=np.int64)
        np.cumsum(np.concatenate(lens_l), out=p_ptr[1:])
        p_bytes = np.concatenate(bytes_l)
    deg = np.concatenate(deg_parts)
    lens_rows = st.lens[row_ids].astype(np.int64)
    packed, off, n_cand = st.msa_correct_built_batch(n_parts, n_rows, deg, int(lens_rows.sum()) + 16 * n_rows + 1024, p_row, p_col, p_ptr, p_bytes)
    part_of_row = np.repeat(np.arange(n_parts), np.diff(first_row))
    redo = sorted(set(part_of_row[n_cand < 0].tolist()))
    ok_row = np.ones(n_rows, dtype=bool)
    for p in redo:
        ok_row[int(first_row[p]):int(first_row[p + 1])] = False
    out = {}
    todo = np.flatnonzero((deg == 1) & (n_cand > 0) & ok_row)
    if len(todo):
        H = _lib.pyhelp()
        rows_todo = todo.tolist()
        if H is not None and hasattr(H, "split_ascii_rows"):
            # the corrected rows only, equal ones as ONE str object (they converge on their consensus: every dict built over them
            # afterwards hashes an object once and compares by identity first)
            off64, sel = np.ascontiguousarray(off, dtype=np.int64), np.ascontiguousarray(todo, dtype=np.int64)
            strs_todo = H.split_ascii_rows(packed.ctypes.data, off64.ctypes.data, len(off64) - 1, sel.ctypes.data, len(sel))
        else:
            flat = packed[:off[n_rows]].tobytes().decode()
            strs_todo = [flat[off[r]:off[r + 1]] for r in rows_todo]
        acc_lists = list(map(seq_to_acc.__getitem__, map(keys.__getitem__, rows_todo)))
        if all(map((1).__eq__, map(len, acc_lists))):          # a read that is still being corrected has multiplicity 1: one accession each
            out.update(zip(map(next, map(iter, acc_lists)), strs_todo))
        else:
            for s_r, accs_of_r in zip(strs_todo, acc_lists):
                for acc in accs_of_r:
                    out[acc] = s_r
    return out, [centres[p] for p in redo]


def correct_strings(partition_alignments, seq_to_acc, ccs_dict, step, nr_cores=1, verbose=False):
    """correction_module.py:12-76.  partition_alignments: {centre: {s: (ed, aln_centre, aln_s, degree)}};
    seq_to_acc: {sequence: [accessions]}.  Returns (S_prime, S_prime_quality) -- the second is always {} here.
    A partition_alignments that still carries its alignments as CIGAR ops (isocon_get_candidates.get_partition_alignments on the
    store the NN search remembered) is corrected from those on the device, ALL its partitions in one batched build + correct; gapped
    strings are never expanded."""
    if ccs_dict:
        raise NotImplementedError("correction with CCS quality values (disabled in the reference, isocon_get_candidates.py:106)")
    S_prime = {}
    batch = getattr(partition_alignments, "batch", None)
    from_ops = batch is not None and batch.alive() and _correct_on_device is _CORRECT_ON_DEVICE
    single = []          # partitions for the one-at-a-time paths
    batched = []
    for m, partition in sorted(partition_alignments.items()):
        n_members = len(batch.rows_of.get(m, [])) if from_ops else 0
        # unchanged since it was built: as many entries as the batch filed there, and every one of them the batch's own value under its
        # own key (a caller that replaced or re-keyed an entry gets the string path, which reads the dict)
        if from_ops and len(partition) == 1 + n_members and _rows_intact(partition, m, batch):
            if n_members >= 1 and partition[m][3] + n_members > 2:          # correction_module.py:263: len(partition) > 1 and N_t > 2
                batched.append(m)
        else:
            single.append(m)
    if batched:
        part, redo = _correct_all_from_ops(batch, partition_alignments, batched, seq_to_acc)
        S_prime.update(part)
        for m in redo:          # a row with more correctable positions than the batched kernel's list: the single-partition kernels
            for acc, s in _correct_partition_from_ops(batch, m, partition_alignments[m], seq_to_acc).items():
                assert acc not in S_prime
                S_prime[acc] = s
    for m in single:
        partition = partition_alignments[m]
        acc_of = {m: seq_to_acc[m]}
        for s in partition:
            if s in seq_to_acc:
                acc_of[s] = seq_to_acc[s]
        for acc, s in correct_to_consensus(m, partition, acc_of, step, verbose).items():
            assert acc not in S_prime
            S_prime[acc] = s
    return S_prime, {}


_CORRECT_ON_DEVICE = _correct_on_device          # (the CPU tests substitute the numpy checker: then the string path runs)
