"""Nearest-neighbour graph construction on top of the MI355X NN search.

Mirror of /root/reference/modules/graphs.py:29-82 (`construct_exact_nearest_neighbor_graph`): same signature, same
node/edge attributes (`degree`, `edit_distance`), same "already converged" handling.  The all-pairs alignment behind it
runs on the GPU through isocon_amd.nearest_neighbor_graph.  The reference builds a networkx.DiGraph keyed by the 2.5 kb
sequences (requirements.txt:3) -- 0.5 s of dict work at 50 000 reads, of which its callers on the hot path read exactly one
thing, G.nodes[seq]["degree"] (isocon_get_candidates.py:64).  Here the graph is kept as integer arrays (what the partition
routine isocon_partition_ids consumes) behind an object that answers `.nodes` itself and turns into the very networkx graph
the reference would have built -- same node order, same edge order, same attributes -- the first time anything else is asked.
"""
from __future__ import annotations

from collections import Counter

import numpy as np

from . import nearest_neighbor_graph


def _nx():
    try:
        import networkx as nx
    except ImportError as e:      # the reference depends on it as well
        raise ImportError("isocon_amd.graphs needs networkx (the reference's graph container)") from e
    return nx


class _Nodes(object):
    """G.nodes of a LazyDiGraph: G.nodes[seq] -> {"degree": multiplicity}, G.nodes() / iteration -> the sequences in insertion order"""

    def __init__(self, names, degree, counts=None):
        self._names = names
        self._degree = degree
        self._index = None
        self._counts = counts          # {seq: multiplicity} if the builder has it anyway: degree lookups without an index of the names

    def _idx(self):
        if self._index is None:
            self._index = {s: i for i, s in enumerate(self._names)}
        return self._index

    def __getitem__(self, seq):
        if self._counts is not None:
            return {"degree": self._counts[seq]}
        return {"degree": self._degree[self._idx()[seq]]}

    def __call__(self, data=False):
        if data:
            return [(s, {"degree": d}) for s, d in zip(self._names, self._degree)]
        return self

    def __iter__(self):
        return iter(self._names)

    def __len__(self):
        return len(self._names)

    def __contains__(self, seq):
        return seq in (self._counts if self._counts is not None else self._idx())


class LazyDiGraph(object):
    """G_star as arrays: names[i] (unique sequences in first-appearance order), degree[i], edges ea[e] -> eb[e] with distance ed[e]
    (ids into names; in the order the reference inserts them: rows in length-sorted order, neighbours in the NN order).  `.nodes` is
    answered from the arrays; any other attribute materialises the networkx.DiGraph of graphs.py:37-69 and is forwarded to it."""

    def __init__(self, names, degree, ea, eb, ed, counts=None):
        self.names, self.degree, self.ea, self.eb, self.ed = names, degree, ea, eb, ed
        self.nodes = _Nodes(names, degree, counts)
        self._g = None

    def to_networkx(self):
        if self._g is None:
            nx = _nx()
            G = nx.DiGraph()
            G.add_nodes_from((seq, {"degree": d}) for seq, d in zip(self.names, self.degree))
            names = self.names
            G.add_edges_from((names[a], names[b], {"edit_distance": d}) for a, b, d in zip(self.ea.tolist(), self.eb.tolist(), self.ed.tolist()))
            self._g = G
        return self._g

    def __getattr__(self, name):          # (only called for attributes this object does not have)
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.to_networkx(), name)

    def __getitem__(self, key):
        return self.to_networkx()[key]

    def __iter__(self):
        return iter(self.names)

    def __len__(self):
        return len(self.names)

    def __contains__(self, seq):
        return seq in self.nodes


def construct_exact_nearest_neighbor_graph(S, params):
    """graphs.py:29-82.  S: {acc: seq} (not necessarily unique).  Returns (G, converged): a node per unique sequence
    with weight `degree` = multiplicity; an edge s1 -> s2 (attribute `edit_distance`) for every nearest neighbour s2 of
    a sequence s1 of multiplicity 1.  G is a LazyDiGraph (see above)."""
    counts = dict(Counter(S.values()))                             # multiplicities (graphs.py:37-51); a plain dict: unknown sequences raise KeyError
    names = list({seq: None for seq in S.values()})                 # unique sequences, first appearance first (node order of the reference)
    n = len(names)
    degree = list(map(counts.__getitem__, names))
    deg = np.asarray(degree, dtype=np.int64)
    converged = bool((deg > 1).all())                               # no sequence of multiplicity 1 left
    empty = np.zeros(0, dtype=np.int64)
    if converged:
        return LazyDiGraph(names, degree, empty, empty, empty, counts), converged
    # NNG.compute_nearest_neighbor_graph: unique sequences, stable sort by length (NNG:243-246)
    lens = np.fromiter(map(len, names), dtype=np.int64, count=n)
    order = np.argsort(lens, kind="stable")
    seqs_sorted = [names[i] for i in order.tolist()]
    conv = (deg[order] > 1).astype(np.uint8)
    if not hasattr(nearest_neighbor_graph, "nn_1set_arrays"):
        # a stand-in with the reference's interface only (the tests run these callers on the CPU oracle): through the dict of dicts
        unique_strings = {seq: acc for acc, seq in S.items()}          # last accession of a sequence wins (graphs.py:56)
        S_prime = {acc: seq for seq, acc in unique_strings.items()}
        has_converged = set(s for s, d in zip(names, degree) if d > 1)
        edges, _isolated = nearest_neighbor_graph.compute_nearest_neighbor_graph(S_prime, has_converged, params)
        index = {s: i for i, s in enumerate(names)}
        trip = [(index[S[a1]], index[S[a2]], ed) for a1, nbrs in edges.items() if S[a1] not in has_converged for a2, ed in nbrs.items()]
        arr = np.asarray(trip, dtype=np.int64).reshape(-1, 3)
        return LazyDiGraph(names, degree, arr[:, 0].copy(), arr[:, 1].copy(), arr[:, 2].copy(), counts), converged
    best, row_ptr, cols = nearest_neighbor_graph.nn_1set_arrays(seqs_sorted, conv, params.neighbor_search_depth)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(np.asarray(row_ptr, dtype=np.int64)))
    keep = conv[rows] == 0                                          # (graphs.py:61-69 skips converged s1; they have no rows anyway)
    rows, c = rows[keep], np.asarray(cols, dtype=np.int64)[keep]
    return LazyDiGraph(names, degree, order[rows], order[c], np.asarray(best, dtype=np.int64)[rows], counts), converged


def construct_exact_2set_nearest_neighbor_bipartite_graph(X, C, X_file, C_file, params):
    """graphs.py:150-160.  X: {read_acc: seq}, C: {cand_acc: seq} (X_file / C_file are unused by the reference as well).
    Returns the bipartite DiGraph read -> nearest candidate(s): node attribute `bipartite` = 0 (reads, every read, also
    those without a candidate in reach) / 1 (candidates that are somebody's nearest neighbour)."""
    nx = _nx()
    best_exact_matches = nearest_neighbor_graph.compute_2set_nearest_neighbor_graph(X, C, params)
    G = nx.DiGraph()
    G.add_nodes_from(best_exact_matches.keys(), bipartite=0)
    G.add_nodes_from(set(c for x in best_exact_matches for c in best_exact_matches[x]), bipartite=1)
    G.add_edges_from((x, c) for x in best_exact_matches for c in best_exact_matches[x])
    return G
