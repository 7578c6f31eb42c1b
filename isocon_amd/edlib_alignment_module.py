"""GPU drop-in for /root/reference/modules/edlib_alignment_module.py (same names, arguments, return shapes).

All distances come from isocon_ed_pairs (include/isocon_hip.h); `nr_cores` is accepted and ignored (the reference
forks a Pool per call, EAM:28-41 -- here one batched launch replaces it).
"""
from __future__ import annotations

import numpy as np

from . import _lib, perf_log
from .store import SeqStore, store_for_pairs


def _intern(pairs):
    """[(x, y), ...] -> (unique sequence list, a ids, b ids)."""
    index, seqs = {}, []
    a = np.empty(len(pairs), dtype=np.uint32)
    b = np.empty(len(pairs), dtype=np.uint32)
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
    return seqs, a, b


def _distances(pairs):
    if not pairs:
        return np.zeros(0, dtype=np.int32)
    with perf_log.call("edlib_alignment_module.distances", pairs=len(pairs)) as rec:
        st, a, b, owned = store_for_pairs(pairs)
        try:
            ed, ms = st.ed_pairs(a, b, None, return_ms=True)
            rec.add(kernel_ms=ms)
        finally:
            if owned:
                st.close()
    assert (ed >= 0).all()  # EAM:113
    return ed


def edlib_align_sequences(matches, nr_cores=1):
    """EAM:10-49.  {s1: iterable(s2)} -> {s1: {s2: ed}} keyed by the sequences; keys without members are absent."""
    H = _lib.pyhelp()
    if H is not None and hasattr(H, "pairs_of") and type(matches) is dict and matches:
        # on the store the NN graph remembered (the pipeline's case, isocon_get_candidates.py:129-130,38): pairs, ids and the dict of dicts in C
        # without a tuple per pair (cpy/_pyhelp.c): 19 -> 10 ms for the 49 990 partition pairs of C3
        from . import store as _store
        st, index = _store._RECENT["store"], _store._RECENT["index"]
        if st is not None and index is not None and getattr(st, "_h", None):
            n = sum(len(v) for v in matches.values())
            a = np.empty(max(n, 1), dtype=np.uint32)
            b = np.empty(max(n, 1), dtype=np.uint32)
            rows = H.pairs_of(matches, index.mapping(), a.ctypes.data, b.ctypes.data, n)
            if rows is not None:
                outer, counts, inner = rows
                if not inner:
                    return {}
                with perf_log.call("edlib_alignment_module.distances", pairs=len(inner)) as rec:
                    ed, ms = st.ed_pairs(a[:len(inner)], b[:len(inner)], None, return_ms=True)
                    rec.add(kernel_ms=ms)
                assert (ed >= 0).all()  # EAM:113
                ed = np.ascontiguousarray(ed, dtype=np.int32)
                return H.distance_rows(outer, counts, inner, ed.ctypes.data)
    if H is not None and hasattr(H, "distance_dict") and type(matches) is dict and matches and \
            all(type(v) in (set, frozenset, list, tuple, dict) for v in matches.values()) and len(set(type(v) is dict for v in matches.values())) == 1:
        # the pair list, the ids and the dict of dicts in C (cpy/_pyhelp.c): 19 -> 6 ms for the 49 990 partition pairs of C3
        pairs = H.flatten_pairs(matches)[0]
        ed = np.ascontiguousarray(_distances(pairs), dtype=np.int32)
        return H.distance_dict(pairs, ed.ctypes.data)
    pairs = [(s1, s2) for s1 in matches for s2 in matches[s1]]
    ed = _distances(pairs)
    exact_edit_distances = {}
    for (s1, s2), d in zip(pairs, ed):
        exact_edit_distances.setdefault(s1, {})[s2] = int(d)
    return exact_edit_distances


def edlib_align_sequences_keeping_accession(matches, nr_cores=1):
    """EAM:51-99.  {acc1: {acc2: (s1, s2)}} -> {acc1: {acc2: (s1, s2, ed)}}."""
    keys = [(a1, a2) for a1 in matches for a2 in matches[a1]]
    pairs = [(matches[a1][a2][0], matches[a1][a2][1]) for a1, a2 in keys]
    ed = _distances(pairs)
    exact_matches = {}
    for (a1, a2), (s1, s2), d in zip(keys, pairs, ed):
        exact_matches.setdefault(a1, {})[a2] = (s1, s2, int(d))
    return exact_matches


def edlib_alignment(x, y, i, j, x_acc="", y_acc=""):
    """EAM:107-128 (single pair)."""
    ed = int(_distances([(x, y)])[0])
    if x_acc == y_acc == "":
        return (x, y, ed)
    return (x_acc, y_acc, (x, y, ed))


def edlib_alignment_helper(arguments):
    """EAM:103-105."""
    args, kwargs = arguments
    return edlib_alignment(*args, **kwargs)


def edlib_traceback(x, y, mode="NW", task="path", k=1):
    """EAM:130-135: (editDistance, locations, cigar) of edlib.align(x, y, mode="NW", task="path", k=k).  Its only caller
    (isocon_statistical_test.get_nearest_neighbor_graph, :63-104) is never called in v0.3.3.  The
    k-bounded distance comes from the GPU; above k edlib reports (-1, [], None).  The path of a hit is traced on the host
    (functions.nw_path_cigar, O(len x * len y): this is not a hot path) with the tie rule used everywhere else (from the
    end: I, then D, then the diagonal; "parity unpinned").  HW mode lives in end_invariant_functions.edlib_traceback."""
    if mode != "NW" or task != "path":
        raise NotImplementedError("edlib_alignment_module.edlib_traceback: only mode='NW', task='path' (EAM:130-135)")
    from .store import SeqStore
    st = SeqStore([x, y])
    try:
        ed = int(st.ed_pairs([0], [1], None if k is None or k < 0 else [int(k)])[0])
    finally:
        st.close()
    if ed < 0:
        return -1, [], None
    from .functions import nw_path_cigar
    ops = nw_path_cigar(x, y)
    cigar, i = [], 0
    while i < len(ops):
        j = i
        while j < len(ops) and ops[j] == ops[i]:
            j += 1
        cigar.append("%d%s" % (j - i, ops[i]))
        i = j
    return ed, [(0, len(y) - 1)], "".join(cigar)
