"""Partition of the nearest-neighbour graph into consensus-centre neighbourhoods.

Mirror of /root/reference/modules/partitions.py:301-413 (`get_partitions_no_copy`) and :416-593 (`partition_strings`).
The reference walks Python sets of strings, so its visiting order depends on PYTHONHASHSEED; its RESULT is
order-independent except when two different reachable sets tie in weight.  This restatement works on integer ids and
is deterministic: candidates are ranked by (weight of the reachable set, number of direct in-neighbours, sequence) and
all members of a strongly connected top set compete as its representative.  tests/golden/g7_partitions.json holds
outputs of the reference itself (eight hash seeds, all agreeing) that this module must reproduce.

The partition itself runs in native code on integer ids (isocon_partition_ids, csrc/partition_host.hpp: 50 000 nodes in a few
milliseconds instead of 0.35 s of Python sets); `partition_ids_py` below is the same algorithm in Python, kept as the statement the
native routine is tested against (tests/test_partition_native.py).  partition_strings never builds the reference's networkx
graph: it hands the arrays of isocon_amd.graphs.LazyDiGraph straight to the routine.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib, graphs


def _string_ranks(names):
    """rank[i] = position of names[i] in sorted(names) (the reference breaks ties with `m < centre` on the strings)"""
    n = len(names)
    H = _lib.pyhelp()
    if H is not None and hasattr(H, "rank_strings") and isinstance(names, list):
        out = np.zeros(max(n, 1), dtype=np.uint32)
        try:
            H.rank_strings(names, out.ctypes.data)
            return out[:n]
        except TypeError:          # (not str / not ASCII: accessions of the end-invariant graph are str too, but be general)
            pass
    rank = np.zeros(max(n, 1), dtype=np.uint32)
    rank[sorted(range(n), key=names.__getitem__)] = np.arange(n, dtype=np.uint32)
    return rank[:n]


def partition_ids(n, degree, edges, names, nbr_tiebreak=True):
    """get_partitions_no_copy on integer ids through the C ABI (isocon_partition_ids; host code of libisocon_hip.so).

    n nodes, degree[i] = multiplicity, edges = (a[], b[]) arrays or [(a, b)]: a's nearest neighbour is b (an edge of G*),
    names[i] = the sequence (only used to break ties the way the reference does, `m < centre`).
    nbr_tiebreak: between start nodes of equal reachable weight prefer the one with more direct in-neighbours
    (partitions.py:346-361); False = only the name decides (end_invariant_functions.py:461-470).
    Returns [(centre, weight, members ndarray)] in the reference's extraction order (components by size, largest first)."""
    L = _lib.load()
    if isinstance(edges, tuple):
        ea, eb = (np.ascontiguousarray(x, dtype=np.uint32) for x in edges)
    else:
        arr = np.asarray(edges, dtype=np.uint32).reshape(-1, 2)
        ea, eb = np.ascontiguousarray(arr[:, 0]), np.ascontiguousarray(arr[:, 1])
    deg = np.ascontiguousarray(degree, dtype=np.int32)
    rank = np.ascontiguousarray(_string_ranks(names), dtype=np.uint32)
    centre = np.zeros(max(n, 1), dtype=np.uint32)
    weight = np.zeros(max(n, 1), dtype=np.int64)
    ptr = np.zeros(n + 1, dtype=np.uint64)
    members = np.zeros(max(n, 1), dtype=np.uint32)
    n_parts = ctypes.c_uint32(0)
    P = ctypes.POINTER
    _lib.check(L.isocon_partition_ids(n, deg.ctypes.data_as(_lib.i32p), len(ea), ea.ctypes.data_as(_lib.u32p), eb.ctypes.data_as(_lib.u32p),
                                      rank.ctypes.data_as(_lib.u32p), 1 if nbr_tiebreak else 0, centre.ctypes.data_as(_lib.u32p),
                                      weight.ctypes.data_as(P(ctypes.c_int64)), ptr.ctypes.data_as(_lib.u64p), members.ctypes.data_as(_lib.u32p),
                                      ctypes.byref(n_parts)), "isocon_partition_ids")
    k = int(n_parts.value)
    ptr = ptr.astype(np.int64)
    return [(int(centre[p]), int(weight[p]), members[ptr[p]:ptr[p + 1]]) for p in range(k)]


def _reach(start, succ, alive):
    """Nodes reachable from `start` along `succ` inside `alive` (iterative DFS), start included."""
    seen = {start}
    stack = [start]
    while stack:
        v = stack.pop()
        for w in succ[v]:
            if w in alive and w not in seen:
                seen.add(w)
                stack.append(w)
    return seen


def partition_ids_py(n, degree, edges, names, nbr_tiebreak=True):
    """The same partition in Python (the statement isocon_partition_ids is tested against; not used by the product path).

    n nodes, degree[i] = multiplicity, edges = [(a, b)]: a's nearest neighbour is b (an edge of G*; the search runs on
    the transpose, b -> a), names[i] = the sequence (only used to break ties the way the reference does, `m < centre`).
    nbr_tiebreak: between start nodes of equal reachable weight prefer the one with more direct in-neighbours
    (partitions.py:346-361); False = only the name decides (end_invariant_functions.py:461-470).
    Returns [(centre, weight, members)] in the reference's extraction order (components by size, largest first)."""
    rank = [0] * n                       # position of the sequence in sorted order: compares like the strings, in O(1)
    for r, v in enumerate(sorted(range(n), key=lambda v: names[v])):
        rank[v] = r
    succ_t = [[] for _ in range(n)]      # transpose: b -> a   (who points at me)
    succ_g = [[] for _ in range(n)]      # G*: a -> b
    for a, b in edges:
        succ_t[b].append(a)
        succ_g[a].append(b)

    # weakly connected components in first-node order, then by size (stable), partitions.py:306-307
    comp_of = [-1] * n
    comps = []
    for s in range(n):
        if comp_of[s] >= 0:
            continue
        cid = len(comps)
        comp_of[s] = cid
        members = [s]
        stack = [s]
        while stack:
            v = stack.pop()
            for w in succ_t[v] + succ_g[v]:
                if comp_of[w] < 0:
                    comp_of[w] = cid
                    members.append(w)
                    stack.append(w)
        comps.append(members)
    comps.sort(key=len, reverse=True)

    out = []
    live_in = [len(x) for x in succ_t]                    # direct in-neighbours of G* still in the graph (transpose successors)
    for members in comps:
        alive = set(members)
        while alive:
            # One sweep: reachable set and weight of every start node that is not inside an earlier start's set (its own
            # set would be a subset of that one).  Keys: (-weight, -direct in-neighbours of the representative, its rank);
            # every node of the strongly connected top of a set may represent it.
            # (hubs first: most nodes are then inside an earlier start's set and never start a search of their own)
            order = sorted(alive, key=lambda v: (-live_in[v], rank[v]))
            processed = set()
            cands = []
            best_w = -1
            for m in order:
                if m in processed:
                    continue
                if live_in[m] == 0:
                    reach = {m}
                else:
                    reach = _reach(m, succ_t, alive)
                    processed |= reach
                weight = degree[m] if len(reach) == 1 else sum(degree[v] for v in reach)
                cands.append((weight, m, reach))
                if weight > best_w:
                    best_w = weight

            def full_key(weight, m, reach):
                if len(reach) == 1:
                    return (-weight, 0, rank[m])
                top = reach & _reach(m, succ_g, alive)
                if nbr_tiebreak:
                    rep = min(top, key=lambda v: (-live_in[v], rank[v]))
                    return (-weight, -live_in[rep], rank[rep])
                return (-weight, 0, rank[min(top, key=rank.__getitem__)])

            # Extract in key order for as long as the next set is untouched by what was removed in this sweep: removing
            # nodes can only shrink other sets, so an untouched set with the best key is what a fresh sweep would pick.
            cands.sort(key=lambda c: -c[0])
            removed = set()
            k = 0
            while k < len(cands):
                w = cands[k][0]
                e = k
                while e < len(cands) and cands[e][0] == w:
                    e += 1
                group = sorted(cands[k:e], key=lambda c: full_key(*c)) if e - k > 1 else cands[k:e]
                stop = False
                for weight, m, reach in group:
                    if removed and not removed.isdisjoint(reach):
                        stop = True
                        break
                    # the centre: largest direct weight (own multiplicity + direct in-neighbours), then smallest sequence
                    centre = m if len(reach) == 1 else min(reach, key=lambda v: (-(degree[v] + live_in[v]), rank[v]))
                    out.append((centre, weight, reach - {centre}))
                    removed |= reach
                    alive -= reach
                    for v in reach:                       # their nearest neighbours lose an in-neighbour
                        for u in succ_g[v]:
                            live_in[u] -= 1
                if stop:
                    break
                k = e
    return out


def _partition_graph(G, transposed, nbr_tiebreak=True):
    if isinstance(G, graphs.LazyDiGraph) and not transposed:
        names, degree, edges = G.names, G.degree, (G.ea, G.eb)
    else:
        names = list(G.nodes())
        idx = {s: i for i, s in enumerate(names)}
        degree = [G.nodes[s]["degree"] for s in names]
        if transposed:
            edges = [(idx[b], idx[a]) for a, b in G.edges()]      # transpose edge a -> b  <=>  G* edge b -> a
        else:
            edges = [(idx[a], idx[b]) for a, b in G.edges()]
    M, partition = {}, {}
    name_of = names.__getitem__
    for centre, weight, members in partition_ids(len(names), degree, edges, names, nbr_tiebreak):
        M[names[centre]] = weight
        partition[names[centre]] = set(map(name_of, members.tolist()))
    return M, partition


def get_partitions_no_copy(G_transpose):
    """partitions.py:301-413.  G_transpose: networkx.DiGraph (edge centre -> follower), node attribute `degree`.
    Returns (M, partition): M[centre] = total weight, partition[centre] = set of the other sequences.  Unlike the
    reference the input graph is left untouched."""
    return _partition_graph(G_transpose, True)


def partition_strings(S, params):
    """partitions.py:416-593.  Returns (G_star, partition, M, converged).  G_star: isocon_amd.graphs.LazyDiGraph (the reference's
    networkx graph on demand)."""
    G_star, converged = graphs.construct_exact_nearest_neighbor_graph(S, params)
    M, partition = _partition_graph(G_star, False)      # same result as on nx.reverse(G_star), without the deep copy
    # every unique sequence is in exactly one partition (partitions.py:590-591)
    assert sum(len(partition[p]) + 1 for p in partition) == len(G_star.nodes)
    assert all(m not in partition[m] for m in partition)
    return G_star, partition, M, converged


def partition_strings_2set(X, C, X_file, C_file, params):
    """partitions.py:595-647: greedy assignment of reads to candidates on the bipartite NN graph -- the candidate with
    the most supporting reads takes all of them (ties: the smallest accession, `max(sorted(...))` in the reference),
    they leave the graph, repeat until no candidate is left.  Reads with no candidate in reach end up in no partition.
    Returns (G_star, partition) with partition[cand_acc] = set(read_acc); G_star is left untouched."""
    G_star = graphs.construct_exact_2set_nearest_neighbor_bipartite_graph(X, C, X_file, C_file, params)
    support = {}                       # candidate -> set of reads still pointing at it
    of_read = {}
    for x, c in G_star.edges():
        support.setdefault(c, set()).add(x)
        of_read.setdefault(x, set()).add(c)
    partition = {}
    while support:
        m = min(support, key=lambda c: (-len(support[c]), c))
        reads = support.pop(m)
        partition[m] = set(reads)
        for x in reads:                # the reads leave the graph: they no longer support their other candidates
            for c in of_read[x]:
                if c != m and c in support:
                    support[c].discard(x)
    return G_star, partition
