"""Nearest-neighbour graph construction on top of the MI355X NN search.

Mirror of /root/reference/modules/graphs.py:29-82 (`construct_exact_nearest_neighbor_graph`): same signature, same
node/edge attributes (`degree`, `edit_distance`), same "already converged" handling.  The all-pairs alignment behind it
runs on the GPU through isocon_amd.nearest_neighbor_graph; everything else is bookkeeping on the host, as in the
reference.  The graph container is networkx.DiGraph, which the reference's callers expect (requirements.txt:3).
"""
from __future__ import annotations

from collections import defaultdict

from . import nearest_neighbor_graph


def _nx():
    try:
        import networkx as nx
    except ImportError as e:      # the reference depends on it as well
        raise ImportError("isocon_amd.graphs needs networkx (the reference's graph container)") from e
    return nx


def construct_exact_nearest_neighbor_graph(S, params):
    """graphs.py:29-82.  S: {acc: seq} (not necessarily unique).  Returns (G, converged): a node per unique sequence
    with weight `degree` = multiplicity; an edge s1 -> s2 (attribute `edit_distance`) for every nearest neighbour s2 of
    a sequence s1 of multiplicity 1."""
    nx = _nx()
    predicted_seq_to_acc = defaultdict(list)
    for acc, seq in S.items():
        predicted_seq_to_acc[seq].append(acc)

    G = nx.DiGraph()
    G.add_nodes_from((seq, {"degree": len(list_acc)}) for seq, list_acc in predicted_seq_to_acc.items())
    has_converged = set(seq for seq, list_acc in predicted_seq_to_acc.items() if len(list_acc) > 1)
    converged = len(has_converged) == len(predicted_seq_to_acc)          # no sequence of multiplicity 1 left
    if converged:
        return G, converged

    unique_strings = {seq: acc for acc, seq in S.items()}          # last accession of a sequence wins (graphs.py:56)
    S_prime = {acc: seq for seq, acc in unique_strings.items()}
    edges, _isolated = nearest_neighbor_graph.compute_nearest_neighbor_graph(S_prime, has_converged, params)
    G.add_edges_from((S[s1_acc], S[s2_acc], {"edit_distance": ed})
                     for s1_acc, nbrs in edges.items() if S[s1_acc] not in has_converged
                     for s2_acc, ed in nbrs.items())
    return G, converged


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
