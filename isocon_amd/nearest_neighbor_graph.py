"""GPU drop-in for /root/reference/modules/nearest_neighbor_graph.py (NNG).

compute_nearest_neighbor_graph / compute_2set_nearest_neighbor_graph keep the reference's signatures and return the
same dict-of-dict with the same key order.  The adaptive loops NNG:110-198 / :341-424 are
replaced by isocon_nn_graph (include/isocon_hip.h); `params.nr_cores` is ignored (the result of the reference is
independent of it).  One process per GPU: when the calling program has initialised torch.distributed with more than one
rank (every rank running the same pipeline on the same input), the search is shared between the ranks
(isocon_amd.dist.sharded_nn_graph: pairs split by ownership, min-reductions and one all-gather) and every rank gets the
full graph; nothing else in the callers changes.
"""
from __future__ import annotations

import numpy as np

from . import perf_log
from .store import SeqStore, remember

LAST_STATS = {}  # statistics block of the most recent device call (bench / tests)


def _process_group():
    """torch.distributed if the caller runs one process per GPU (initialised, world size > 1), else None.  torch is never
    imported from here: a program that shards has imported it itself."""
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist
    return None


def _graph(st, seqs, is_converged=None, is_target=None, depth=2 ** 32):
    """(best, row_ptr, cols, stats) of the store: one device call, or this rank's share of it + the exchange steps."""
    group = _process_group()
    if group is None:
        return st.nn_graph(is_converged=is_converged, is_target=is_target, depth=depth)
    from . import dist as D

    def shared(store, conv, targ):
        best, row_ptr, cols, stats_all = D.sharded_nn_graph(store, is_converged=conv, is_target=targ, depth=depth, dist=group, return_stats=True)
        stats = {}
        for part in stats_all:
            for k, v in part.items():
                stats[k] = stats.get(k, 0) + v
        return best, row_ptr, cols, stats

    if D.same_everywhere(st.fingerprint, group):
        return shared(st, is_converged, is_target)
    # The ranks sorted equal-length sequences differently (the callers' orders go back to Python sets).  The arg-min
    # sets do not depend on that order as long as the depth limit does not bind: search in a canonical
    # order -- length, sequence, role -- and bring rows and neighbour order back to this rank's.
    n = len(seqs)
    if depth < n:
        raise RuntimeError("nearest_neighbor_graph: the ranks hold the sequences in different orders and neighbor_search_depth binds")
    conv = np.zeros(n, dtype=np.uint8) if is_converged is None else np.asarray(is_converged, dtype=np.uint8)
    targ = np.zeros(n, dtype=np.uint8) if is_target is None else np.asarray(is_target, dtype=np.uint8)
    order = np.asarray(sorted(range(n), key=lambda i: (len(seqs[i]), seqs[i], int(targ[i]), int(conv[i]))), dtype=np.int64)
    canon = SeqStore([seqs[i] for i in order.tolist()])
    try:
        best_c, row_ptr_c, cols_c, stats = shared(canon, None if is_converged is None else conv[order], None if is_target is None else targ[order])
    finally:
        canon.close()
    best = np.empty_like(best_c)
    best[order] = best_c[:n]
    rows = order[np.repeat(np.arange(n, dtype=np.int64), np.diff(np.asarray(row_ptr_c, dtype=np.int64)))]
    cols = order[np.asarray(cols_c, dtype=np.int64)]
    # this rank's insertion order inside a row: ascending offset, the lower index first (NNG:145-172)
    srt = np.lexsort((cols, np.abs(cols - rows), rows))
    row_ptr = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(np.bincount(rows, minlength=n), out=row_ptr[1:])
    return best, row_ptr, cols[srt].astype(np.uint32), stats


def _rows_to_dict(accs, is_query, best, row_ptr, cols):
    from . import _lib
    H = _lib.pyhelp()
    if H is not None and hasattr(H, "csr_to_dict") and isinstance(accs, list):
        # 50 000 rows, 80 000 edges: the dict of dicts built in C (isocon_amd/cpy/_pyhelp.c) -- same insertion order as the loop below
        b = np.ascontiguousarray(best, dtype=np.int32)
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        c = np.ascontiguousarray(cols, dtype=np.uint32)
        isq = np.ascontiguousarray(is_query, dtype=np.uint8)
        allq = bool(isq.all())
        return H.csr_to_dict(accs, 0 if allq else isq.ctypes.data, b.ctypes.data, rp.ctypes.data, c.ctypes.data if len(c) else 0, len(accs))
    out = {}
    nbr = [accs[c] for c in cols.tolist()]            # neighbour accessions in CSR order: rows are slices of it
    row_ptr = row_ptr.tolist()
    best = best.tolist()
    fromkeys = dict.fromkeys
    if bool(np.all(is_query)):
        for i, acc in enumerate(accs):
            out[acc] = fromkeys(nbr[row_ptr[i]:row_ptr[i + 1]], best[i])
        return out
    for i, acc in enumerate(accs):
        if not is_query[i]:
            continue
        out[acc] = fromkeys(nbr[row_ptr[i]:row_ptr[i + 1]], best[i])
    return out


def nn_1set_arrays(seqs, conv, depth):
    """The exact 1-set graph of the length-sorted unique sequences `seqs` as arrays: (best int32[n], row_ptr int64[n + 1], cols uint32[]),
    conv[i] != 0 = converged (no row of its own, NNG:120-123).  The store is remembered for the pair-list wrappers that follow
    (EAM / SWM are called next on pairs of these very sequences).  Used by _nn_1set (dict shapes of NNG) and by
    isocon_amd.partitions.partition_strings, which works on the ids directly."""
    with perf_log.call("nearest_neighbor_graph.1set", sequences=len(seqs)) as rec:
        st = SeqStore(seqs)
        try:
            best, row_ptr, cols, stats = _graph(st, seqs, is_converged=conv, depth=depth)
        except Exception:
            st.close()
            raise
        rec.add(edges=int(len(cols)), **{k: v for k, v in stats.items()})
    remember(st, seqs)
    LAST_STATS.clear()
    LAST_STATS.update(stats)
    return best, row_ptr, cols


def _nn_1set_lists(seqs, accs, has_converged, depth):
    conv = np.fromiter((1 if s in has_converged else 0 for s in seqs), dtype=np.uint8, count=len(seqs)) if has_converged else np.zeros(len(seqs), dtype=np.uint8)
    best, row_ptr, cols = nn_1set_arrays(seqs, conv, depth)
    # every entry gets a key; converged ones an empty dict (NNG:120-123)
    return _rows_to_dict(accs, np.ones(len(accs), dtype=bool), best, row_ptr, cols)


def _nn_1set(seq_to_acc_list_sorted, has_converged, depth):
    return _nn_1set_lists([s for s, _ in seq_to_acc_list_sorted], [a for _, a in seq_to_acc_list_sorted], has_converged, depth)


_SLICED = {"key": None, "graph": None}      # the whole graph of the most recent sliced call


def _sliced(kind, seq_to_acc_list_sorted, role, depth, compute):
    """The sliced entry points are what the reference's Pool calls once per chunk (NNG:33-65: 10 x nr_cores chunks) on the
    SAME list: the whole graph is computed on the first chunk and the following chunks of that list read their rows from
    it.  Key = the strings' (cached) hashes, the accessions, the role set and the depth."""
    key = (kind, len(seq_to_acc_list_sorted), hash(tuple(map(tuple, seq_to_acc_list_sorted))), hash(frozenset(role)), depth)
    if _SLICED["key"] != key:
        _SLICED["graph"] = compute()
        _SLICED["key"] = key
    return _SLICED["graph"]


def get_nearest_neighbors(batch_of_queries, global_index_in_matrix, start_index, seq_to_acc_list_sorted, has_converged,
                          neighbor_search_depth):
    """NNG:110-198.  The rows of the queries [start_index, start_index+len(batch)) of the exact graph."""
    full = _sliced(1, seq_to_acc_list_sorted, has_converged, neighbor_search_depth,
                   lambda: _nn_1set(seq_to_acc_list_sorted, has_converged, neighbor_search_depth))
    keep = [seq_to_acc_list_sorted[i][1] for i in range(start_index, start_index + len(batch_of_queries))]
    return {acc: full[acc] for acc in keep}


def get_nearest_neighbors_helper(arguments):
    args, kwargs = arguments
    return get_nearest_neighbors(*args, **kwargs)


def get_exact_nearest_neighbor_graph(seq_to_acc_list_sorted, has_converged, params):
    """NNG:19-82 (serial and Pool branches give the same dict; one device call here)."""
    return _nn_1set(seq_to_acc_list_sorted, has_converged, params.neighbor_search_depth)


def compute_nearest_neighbor_graph(S, has_converged, params):
    """NNG:237-296 -> (nearest_neighbor_graph, isolated)."""
    from . import _lib
    H = _lib.pyhelp()
    if H is not None and hasattr(H, "unique_values_by_length") and type(S) is dict:
        try:
            # unique sequences (first position, last accession: NNG:243) stably sorted by length (NNG:246), in one pass in C
            seqs, accs = H.unique_values_by_length(S)
        except TypeError:
            seqs = None
        if seqs is not None:
            nearest_neighbor_graph = _nn_1set_lists(seqs, accs, has_converged, params.neighbor_search_depth)
            if len(nearest_neighbor_graph) == len(seqs):       # every unique sequence has a row (NNG:120-123): nothing is isolated
                return nearest_neighbor_graph, set()
            seen = set(S[acc1] for acc1 in nearest_neighbor_graph)
            return nearest_neighbor_graph, set(seqs).difference(seen)
    seq_to_acc = {seq: acc for (acc, seq) in S.items()}
    items = list(seq_to_acc.items())
    # stable sort by length (NNG:246) through numpy: the same order as sorted(..., key=len) at a fraction of the calls
    order = np.argsort(np.fromiter(map(len, seq_to_acc), dtype=np.int64, count=len(items)), kind="stable").tolist()
    seq_to_acc_list_sorted = [items[i] for i in order]
    nearest_neighbor_graph = get_exact_nearest_neighbor_graph(seq_to_acc_list_sorted, has_converged, params)
    if len(nearest_neighbor_graph) == len(seq_to_acc):       # every unique sequence has a row (NNG:120-123): nothing is isolated
        return nearest_neighbor_graph, set()
    seen = set(S[acc1] for acc1 in nearest_neighbor_graph)
    isolated = set(seq_to_acc).difference(seen)
    return nearest_neighbor_graph, isolated


def _replay_2set_depth(seqs, accs, is_t, depth, st):
    """neighbor_search_depth smaller than the number of candidates (never the case with the reference's default
    2**32): NNG:416 stops after `depth` candidate alignments, an order-dependent rule.  Distances to the candidates
    come from the GPU (bounded by len(read), NNG:356); the stop/depth bookkeeping of NNG:362-419 is replayed here."""
    n = len(seqs)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=n)
    t_idx = np.nonzero(is_t)[0]
    q_idx = np.nonzero(~is_t)[0]
    a = np.repeat(q_idx, len(t_idx)).astype(np.uint32)
    b = np.tile(t_idx, len(q_idx)).astype(np.uint32)
    k = lens[a].astype(np.int32)
    ed = st.ed_pairs(a, b, k).reshape(len(q_idx), len(t_idx)) if len(a) else np.zeros((len(q_idx), 0), np.int32)
    col_of = {int(t): c for c, t in enumerate(t_idx)}
    out = {}
    for r, i in enumerate(q_idx.tolist()):
        best_ed = int(lens[i])
        cur = {}
        stop_up = stop_down = False
        processed = 0
        j = 1
        while True:
            if i - j < 0:
                stop_down = True
            if i + j >= n:
                stop_up = True
            if not stop_down and abs(lens[i] - lens[i - j]) > best_ed:
                stop_down = True
            if not stop_up and abs(lens[i] - lens[i + j]) > best_ed:
                stop_up = True
            for side_stopped, p in ((stop_down, i - j), (stop_up, i + j)):
                if side_stopped or not is_t[p]:
                    continue
                processed += 1
                d = int(ed[r, col_of[p]])
                d = d if 0 <= d <= best_ed else -1
                if 0 <= d < best_ed:
                    best_ed = d
                    cur = {accs[p]: d}
                elif d == best_ed:
                    cur[accs[p]] = d
            if stop_down and stop_up:
                break
            if processed >= depth:
                break
            j += 1
        out[accs[i]] = cur
    return out


def _nn_2set(seq_to_acc_list_sorted_all, target_accessions, depth):
    seqs = [s for s, _ in seq_to_acc_list_sorted_all]
    accs = [a for _, a in seq_to_acc_list_sorted_all]
    is_t = np.fromiter((acc in target_accessions for acc in accs), dtype=bool, count=len(accs))
    return _nn_2set_arrays(seqs, accs, is_t, depth)


def _nn_2set_arrays(seqs, accs, is_t, depth):
    """the length-sorted merged list as three parallel sequences: strings, accessions, target flags"""
    with perf_log.call("nearest_neighbor_graph.2set", sequences=len(seqs), targets=int(is_t.sum())) as rec:
        st = SeqStore(seqs)
        try:
            if depth < int(is_t.sum()):
                return _replay_2set_depth(seqs, accs, is_t, depth, st)
            best, row_ptr, cols, stats = _graph(st, seqs, is_target=is_t.astype(np.uint8), depth=depth)
        finally:
            st.close()
        rec.add(edges=int(len(cols)), **{k: v for k, v in stats.items()})
    LAST_STATS.clear()
    LAST_STATS.update(stats)
    return _rows_to_dict(accs, ~is_t, best, row_ptr, cols)


def get_nearest_neighbors_2set(batch, start_index, seq_to_acc_list_sorted, target_accessions, neighbor_search_depth):
    """NNG:341-424 for the entries [start_index, start_index+len(batch))."""
    full = _sliced(2, seq_to_acc_list_sorted, target_accessions, neighbor_search_depth,
                   lambda: _nn_2set(seq_to_acc_list_sorted, target_accessions, neighbor_search_depth))
    keep = [seq_to_acc_list_sorted[i][1] for i in range(start_index, start_index + len(batch))]
    return {acc: full[acc] for acc in keep if acc in full}


def get_nearest_neighbors_2set_helper(arguments):
    args, kwargs = arguments
    return get_nearest_neighbors_2set(*args, **kwargs)


def get_exact_nearest_neighbor_graph_2set(seq_to_acc_list_sorted_all, target_accessions, params):
    """NNG:300-334."""
    return _nn_2set(seq_to_acc_list_sorted_all, target_accessions, params.neighbor_search_depth)


def compute_2set_nearest_neighbor_graph(X, C, params):
    """NNG:201-234: reads X against candidates C -> {read_acc: {cand_acc: ed}}."""
    if type(X) is dict and type(C) is dict and (X or C) and len(set(C).intersection(X)) == 0:
        # NNG:202-208 without 51 000 tuples and a Python key function: reads first, then candidates, STABLE sort by length (the order
        # sorted(queries + targets, key=len) gives); a position at or behind len(X) of the unsorted list is a candidate.  38.7 -> 29 ms at C3.
        from operator import itemgetter
        all_seqs = list(X.values()) + list(C.values())
        all_accs = list(X.keys()) + list(C.keys())
        lens = np.fromiter(map(len, all_seqs), dtype=np.int64, count=len(all_seqs))
        order = np.argsort(lens, kind="stable")
        pick = itemgetter(*order.tolist()) if len(order) > 1 else (lambda lst: (lst[0],))
        return _nn_2set_arrays(list(pick(all_seqs)), list(pick(all_accs)), order >= len(X), params.neighbor_search_depth)
    seq_to_acc_queries = [(seq, acc) for (acc, seq) in X.items()]
    seq_to_acc_targets = [(seq, acc) for (acc, seq) in C.items()]
    seq_to_acc_list_sorted_all = sorted(seq_to_acc_queries + seq_to_acc_targets, key=lambda x: len(x[0]))
    return get_exact_nearest_neighbor_graph_2set(seq_to_acc_list_sorted_all, set(C.keys()), params)


def edlib_ed(x, y, mode="NW", task="distance", k=1):
    """NNG:104-107 (single pair)."""
    if mode != "NW" or task != "distance":
        raise NotImplementedError("only the hot path's mode='NW', task='distance' is implemented")
    st = SeqStore([x, y])
    try:
        return int(st.ed_pairs([0], [1], [k])[0])
    finally:
        st.close()
