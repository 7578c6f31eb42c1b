"""Collapse of candidates that are identical up to their ends (the step between the correction loop and
the read-to-candidate alignment of find_candidate_transcripts).

Mirror of /root/reference/modules/end_invariant_functions.py:884-918 (`is_overlap`), :920-951
(`get_invariants_under_ignored_edge_ends_speed`), :405-533 (`partition_highest_reachable_with_edge_degrees`) and
:975-1065 (`collapse_candidates_under_ends_invariant`).  Pure string work on the (few thousand) candidates -- exact
substring / suffix-prefix tests, no alignment -- so it stays on the host, like in the reference.

Second part: the candidate-vs-candidate graph of the statistical-test phase --
`edlib_traceback` (:593-620, edlib HW mode + path), `get_all_NN` (:622-681), `get_all_NN_under_ignored_edge_ends`
(:687-754) and `get_NN_graph_ignored_ends_edlib` (:757-788): every candidate against all candidates whose length differs
by at most 10 + 2 * ignore_ends_len, infix edit distance with threshold 10 + ignore_ends_len, target overhangs beyond the
ignored ends charged, terminal insertion runs forgiven.  All pairs of a call go through one launch of isocon_hw_pairs
(csrc/hw.hpp); the arithmetic on the five integers it returns per pair is done here as in the reference.  The other
alignment-based variants of that module (`get_nearest_neighbors*`, the parasail graph) have no live caller in v0.3.3."""
from __future__ import annotations

import numpy as np

from . import partitions
from .store import SeqStore


def is_overlap(text1, text2, ignore_ends_threshold):
    """end_invariant_functions.py:884-918: does a suffix of text1 equal a prefix of text2 such that at most
    ignore_ends_threshold characters of either ORIGINAL string are left uncovered?  Like the reference: 0 for an empty
    string, the common length (truthy) if the truncated strings are identical, else True / False."""
    len1, len2 = len(text1), len(text2)
    if len1 == 0 or len2 == 0:
        return 0
    if len1 > len2:
        text1 = text1[-len2:]
    elif len1 < len2:
        text2 = text2[:len1]
    if text1 == text2:
        return min(len1, len2)
    # The reference searches the LONGEST suffix-prefix overlap `best` and accepts iff len1 - best and len2 - best are
    # within the threshold: equivalent to "some overlap of at least max(len1, len2) - threshold characters exists".
    n = len(text1)
    need = max(len1, len2) - ignore_ends_threshold
    if need <= 0:
        return True          # (best >= 0 always satisfies both offsets)
    if n < need:
        return False         # (lengths more than the threshold apart: no overlap of the truncated strings is long enough;
                             #  without this the probe's end index below goes negative and str.find reads it from the end)
    if need >= 16:
        # an overlap of L >= need characters starts at position n - L <= n - need of text1 and begins with text2[:16]:
        # let str.find look for that probe in the admissible start range (one C call instead of a Python loop)
        probe, last_start = text2[:16], n - need
        pos = text1.find(probe, 0, last_start + 16)
        while pos != -1:
            if text1[pos:] == text2[:n - pos]:
                return True
            pos = text1.find(probe, pos + 1, last_start + 16)
        return False
    for L in range(n, need - 1, -1):
        if text1[n - L:] == text2[:L]:
            return True
    return False


def _pair_is_invariant(seq1, seq2, thr):
    """The reference's test for one ordered pair (len(seq1) - 2 thr <= len(seq2) <= len(seq1)), :933-946."""
    if seq2 in seq1:
        start_offset = seq1.find(seq2)
        end_offset = len(seq1) - (start_offset + len(seq2))
        return start_offset <= thr and end_offset <= thr
    return bool(is_overlap(seq1, seq2, thr) or is_overlap(seq2, seq1, thr))


def _shift_consistent(A, B, thr, K=24):
    """Necessary condition for "B[d + t] == A[t] on the whole overlap, for some 0 <= d <= thr" (every relation of
    _pair_is_invariant is one of these, in one of the two directions): A's k-mer at offset thr fixes d, and the k-mers
    in the middle and at the end of the overlap must agree with it.  No copies: str.find / str.startswith with offsets."""
    probe = A[thr:thr + K]
    pos = B.find(probe, thr, 2 * thr + K)
    while pos != -1:
        d = pos - thr
        end = min(len(A), len(B) - d)               # overlap = A[0:end] vs B[d:d+end]
        if end >= thr + K:
            mid = end // 2
            if B.startswith(A[mid:mid + K], d + mid) and B.startswith(A[end - K:end], d + end - K):
                return True
        pos = B.find(probe, pos + 1, 2 * thr + K)
    return False


def get_invariants_under_ignored_edge_ends_speed(candidate_transcripts, candidate_support, params):
    """:920-951.  DiGraph on candidate accessions (node attribute `degree` = support) with edges in both directions
    between candidates of which one is contained in the other, or which overlap suffix-to-prefix, leaving at most
    params.ignore_ends_len characters at either end.

    The reference tests every pair inside the length window (quadratic in the candidates).  Here pairs are proposed by an
    anchor index first: whichever of the three relations holds, the K-mer at offset thr of one string occurs in the
    other at an offset in [thr, 2 thr] (the strings coincide up to a shift of at most thr; K = shortest length - 3 thr),
    so only pairs sharing such a K-mer are tested -- with the reference's own test.  Sets with a very short candidate take the quadratic route."""
    import bisect

    import networkx as nx
    from . import _lib
    H = _lib.pyhelp()
    if H is not None and not hasattr(H, "invariant_partners"):
        H = None
    thr = params.ignore_ends_len
    G = nx.DiGraph()
    for acc in candidate_transcripts:
        G.add_node(acc, degree=candidate_support[acc])
    by_len = sorted(candidate_transcripts.items(), key=lambda x: len(x[1]))
    lens = [len(s) for _, s in by_len]
    pos_of = {acc: i for i, (acc, _) in enumerate(by_len)}
    # anchor length: as long as the shortest candidate allows (related strings are IDENTICAL on their whole overlap, so a
    # long anchor separates candidates of one isoform that differ by a base somewhere); keyed by hash, verified exactly
    K = max(24, lens[0] - 3 * thr - 1) if by_len else 24
    indexed = bool(by_len) and lens[0] >= 3 * thr + 24 + 1
    if indexed:
        anchor_owner = {}                  # hash of the K-mer at offset thr -> candidates having it there
        window_owner = {}                  # hash of a K-mer at an offset in [thr, 2 thr] -> candidates
        for acc, seq in by_len:
            anchor_owner.setdefault(hash(seq[thr:thr + K]), []).append(acc)
            for o in range(thr, 2 * thr + 1):
                window_owner.setdefault(hash(seq[o:o + K]), set()).add(acc)
    for i1, (acc1, seq1) in enumerate(by_len):
        lo = bisect.bisect_left(lens, len(seq1) - 2 * thr)          # shorter candidates cannot be merged
        hi = bisect.bisect_right(lens, len(seq1))                   # the reference stops at the first longer one
        if indexed:
            partners = set(window_owner.get(hash(seq1[thr:thr + K]), ()))           # my anchor inside their window
            for o in range(thr, 2 * thr + 1):
                partners.update(anchor_owner.get(hash(seq1[o:o + K]), ()))          # their anchor inside my window
            todo = sorted((pos_of[a] for a in partners if lo <= pos_of[a] < hi and a != acc1))
            todo = [by_len[k] for k in todo]
        else:
            todo = [x for x in by_len[lo:hi] if x[0] != acc1]
        if H is not None and todo:
            # the reference's test for all partners of seq1 in one call of the CPython helper (cpy/_pyhelp.c: invariant_partners)
            try:
                hits = H.invariant_partners(seq1, [x[1] for x in todo], thr)
            except TypeError:               # (not plain ASCII strings)
                H, hits = None, None
            if hits is not None:
                for k in hits:
                    acc2 = todo[k][0]
                    G.add_edge(acc2, acc1)
                    G.add_edge(acc1, acc2)
                continue
        for acc2, seq2 in todo:
            if indexed and not (_shift_consistent(seq2, seq1, thr) or _shift_consistent(seq1, seq2, thr)):
                continue                    # (candidates of one isoform share the anchor but differ inside)
            if _pair_is_invariant(seq1, seq2, thr):
                G.add_edge(acc2, acc1)
                G.add_edge(acc1, acc2)
    return G


def partition_highest_reachable_with_edge_degrees(G_star, params):
    """:405-533.  Same greedy extraction as partitions.get_partitions_no_copy, ties between equally heavy reachable sets
    decided by the accession only.  Returns (G_star, partition, M)."""
    M, partition = partitions._partition_graph(G_star, False, nbr_tiebreak=False)
    members = set(partition)
    for m in partition:
        members.update(partition[m])
    assert members == set(G_star.nodes())
    return G_star, partition, M


def collapse_candidates_under_ends_invariant(candidate_transcripts, candidate_support, params):
    """:975-1065.  {kept candidate accession: set(accessions merged into it)}."""
    G = get_invariants_under_ignored_edge_ends_speed(candidate_transcripts, candidate_support, params)
    _, partition, _ = partition_highest_reachable_with_edge_degrees(G, params)
    return partition


# ---- candidate-vs-candidate graph with ignored ends (statistical-test phase) ------------------------------------------
def _ends_adjusted(res, len_t, end_threshold):
    """end_invariant_functions.py:597-619 on the kernel's (distance, start, end, leading I run, trailing I run) rows:
    target bases left of `start` / right of `end` beyond the threshold are added, a terminal insertion run is forgiven up
    to the threshold.  Rows with distance -1 (above k: edlib returns no cigar) stay -1."""
    res = np.asarray(res, dtype=np.int64).reshape(-1, 5)
    ed = res[:, 0].copy()
    ok = ed >= 0
    T = int(end_threshold)
    adj = (np.maximum(0, res[:, 1] - T) + np.maximum(0, np.asarray(len_t, dtype=np.int64) - (res[:, 2] + 1) - T)
           - np.minimum(res[:, 4], T) - np.minimum(res[:, 3], T))
    ed[ok] += adj[ok]
    return ed


def edlib_traceback(x, y, mode="HW", task="path", k=1, end_threshold=0):
    """end_invariant_functions.py:593-620 for one pair (x = query, y = target)."""
    if mode != "HW" or task != "path":
        raise NotImplementedError("end_invariant_functions.edlib_traceback: only mode='HW', task='path' (the reference's only use, :661,:668)")
    res = SeqStore([x, y]).hw_pairs([0], [1], [int(k)])
    return int(_ends_adjusted(res, [len(y)], end_threshold)[0])


def _window_pairs(lens, q_lo, q_hi, window, depth):
    """Neighbour indices the loop of get_all_NN visits for every query q_lo <= i < q_hi, in its order (offset j = 1, 2, ...:
    i - j, then i + j; either side stops for good at the first length difference above `window` or at the list's end,
    :636-655; offsets stop at max(1, depth), :677).  Returns (query index, neighbour index) arrays."""
    n = len(lens)
    jmax = max(1, int(min(depth, n)))
    qs, ts = [], []
    sorted_ok = bool((np.diff(lens) >= 0).all())
    idx = np.arange(q_lo, q_hi, dtype=np.int64)
    if sorted_ok and len(idx):
        lo = np.maximum(np.searchsorted(lens, lens[idx] - window, side="left"), idx - jmax)
        hi = np.minimum(np.searchsorted(lens, lens[idx] + window, side="right") - 1, idx + jmax)
        cd, cu = idx - lo, hi - idx
        tot = cd + cu
        q = np.repeat(idx, tot)
        first = np.zeros(len(idx) + 1, dtype=np.int64)
        np.cumsum(tot, out=first[1:])
        r = np.arange(int(first[-1]), dtype=np.int64) - np.repeat(first[:-1], tot)      # rank within the query's list
        cdr, cur = np.repeat(cd, tot), np.repeat(cu, tot)
        both = np.minimum(cdr, cur)
        # the first 2 * both entries alternate down / up; then only the longer side continues
        j_alt = r // 2 + 1
        down_alt = (r % 2) == 0
        rest = r - 2 * both
        in_alt = r < 2 * both
        j = np.where(in_alt, j_alt, both + rest + 1)
        down = np.where(in_alt, down_alt, cdr > cur)
        t = np.where(down, q - j, q + j)
        return q, t
    for i in idx.tolist():                                     # general list (not length-sorted): the loop itself
        stop_down = stop_up = False
        j = 1
        while True:
            if i - j < 0:
                stop_down = True
            if i + j >= n:
                stop_up = True
            if not stop_down and abs(int(lens[i]) - int(lens[i - j])) > window:
                stop_down = True
            if not stop_up and abs(int(lens[i]) - int(lens[i + j])) > window:
                stop_up = True
            if not stop_down:
                qs.append(i); ts.append(i - j)
            if not stop_up:
                qs.append(i); ts.append(i + j)
            if (stop_down and stop_up) or j >= depth:
                break
            j += 1
    return np.asarray(qs, dtype=np.int64), np.asarray(ts, dtype=np.int64)


def get_all_NN(batch_of_queries, global_index_in_matrix, start_index, seq_to_acc_list_sorted, neighbor_search_depth, ignore_ends_threshold):
    """end_invariant_functions.py:622-681: {acc1: {acc2: ed}} for the queries start_index .. start_index + len(batch) of
    the (length-sorted) list against their length window; an edge is kept iff 0 <= ed <= 10 after the end adjustments."""
    max_variants = 10
    max_ed_allowed = max_variants + ignore_ends_threshold
    window = max_variants + 2 * ignore_ends_threshold
    seqs = [s for s, _ in seq_to_acc_list_sorted]
    accs = [a for _, a in seq_to_acc_list_sorted]
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    q_hi = start_index + len(batch_of_queries)
    all_neighbors_graph = {accs[i]: {} for i in range(start_index, q_hi)}
    q, t = _window_pairs(lens, start_index, q_hi, window, neighbor_search_depth)
    if len(q) == 0:
        return all_neighbors_graph
    from .nearest_neighbor_graph import _process_group
    group = _process_group()            # one process per GPU: every rank aligns its share of the pairs, all get all results
    if group is None:
        res = SeqStore(seqs).hw_pairs(q, t, np.full(len(q), max_ed_allowed, dtype=np.int32), reuse_buffer=True)       # (digested right below)
    else:
        from .dist import sharded_hw_pairs
        res = sharded_hw_pairs(SeqStore(seqs), q, t, max_ed_allowed, dist=group)
    ed = _ends_adjusted(res, lens[t], ignore_ends_threshold)
    for p in np.flatnonzero((ed >= 0) & (ed <= max_variants)).tolist():
        all_neighbors_graph[accs[q[p]]][accs[t[p]]] = int(ed[p])
    return all_neighbors_graph


def get_all_NN_under_ignored_edge_ends(seq_to_acc_list_sorted, params):
    """end_invariant_functions.py:687-754.  The reference's Pool chunks (halo = neighbor_search_depth + 1 on either side)
    see exactly the neighbours the serial loop sees and are merged in query order: one call, nr_cores ignored."""
    return get_all_NN(seq_to_acc_list_sorted, 0, 0, seq_to_acc_list_sorted, params.neighbor_search_depth, params.ignore_ends_len)


def get_NN_graph_ignored_ends_edlib(candidate_transcripts, args):
    """end_invariant_functions.py:757-788: the graph above on the unique candidate sequences (last accession wins, first
    position kept, stable sort by length), made symmetric with the smaller of the two directed values."""
    seq_to_acc = {seq: acc for (acc, seq) in candidate_transcripts.items()}
    seq_to_acc_list_sorted = sorted(seq_to_acc.items(), key=lambda x: len(x[0]))
    all_neighbors_graph = get_all_NN_under_ignored_edge_ends(seq_to_acc_list_sorted, args)
    for c1 in all_neighbors_graph:                             # the reference inserts while it iterates the inner dict of
        for c2 in list(all_neighbors_graph[c1]):               # c1 only when c2 == c1, which cannot happen
            ed = all_neighbors_graph[c1][c2]
            if c1 not in all_neighbors_graph[c2]:
                all_neighbors_graph[c2][c1] = ed
            else:
                all_neighbors_graph[c2][c1] = min(all_neighbors_graph[c1][c2], all_neighbors_graph[c2][c1])
    assert len(candidate_transcripts) == len(all_neighbors_graph)
    return all_neighbors_graph
