"""Partition step against outputs of the reference's own partitions.py (tests/golden/g7_partitions.json,
eight PYTHONHASHSEEDs agreeing).  CPU: the host algorithm on the fixture's graph and with the oracle as the NN search;
GPU: the whole partition_strings path with the HIP NN search."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G7 = json.load(open(os.path.join(HERE, "golden", "g7_partitions.json")))


class Params(object):
    nr_cores = 1
    neighbor_search_depth = 2 ** 32
    verbose = False
    develop_logfile = None


def unique_ids(S):
    uid = {}
    for seq in S.values():
        uid.setdefault(seq, len(uid))
    return uid


def canon(uid, partition, M, converged, G):
    return {"partition": sorted([uid[c], sorted(uid[x] for x in m)] for c, m in partition.items()),
            "M": sorted([uid[c], w] for c, w in M.items()), "converged": bool(converged),
            "nodes": sorted([uid[x], int(G.nodes[x]["degree"])] for x in G.nodes()),
            "edges": sorted([uid[a], uid[b], int(G[a][b]["edit_distance"])] for a, b in G.edges())}


@pytest.mark.parametrize("case", G7["cases"], ids=[c["name"] for c in G7["cases"]])
def test_partition_of_the_reference_graph(case):
    """get_partitions_no_copy on exactly the graph the reference built."""
    import networkx as nx
    from isocon_amd import partitions
    S = dict(case["S"])
    uid = unique_ids(S)
    seq_of = {i: s for s, i in uid.items()}
    G = nx.DiGraph()
    for i, deg in case["expect"]["nodes"]:
        G.add_node(seq_of[i], degree=deg)
    for a, b, d in case["expect"]["edges"]:
        G.add_edge(seq_of[a], seq_of[b], edit_distance=d)
    M, partition = partitions.get_partitions_no_copy(nx.reverse(G))
    got = canon(uid, partition, M, case["expect"]["converged"], G)
    assert got["partition"] == case["expect"]["partition"]
    assert got["M"] == case["expect"]["M"]


def test_partition_is_independent_of_node_order():
    import random
    from isocon_amd import partitions
    case = G7["cases"][0]
    n = len(case["expect"]["nodes"])
    degree = [d for _, d in sorted(case["expect"]["nodes"])]
    edges = [(a, b) for a, b, _ in case["expect"]["edges"]]
    uid = unique_ids(dict(case["S"]))
    names = [None] * n
    for s, i in uid.items():
        names[i] = s
    ref = sorted((names[c], w, sorted(names[v] for v in mem)) for c, w, mem in partitions.partition_ids(n, degree, edges, names))
    rng = random.Random(5)
    for _ in range(3):
        perm = list(range(n)); rng.shuffle(perm)
        inv = [0] * n
        for new, old in enumerate(perm):
            inv[old] = new
        e2 = [(inv[a], inv[b]) for a, b in edges]; rng.shuffle(e2)
        got = partitions.partition_ids(n, [degree[o] for o in perm], e2, [names[o] for o in perm])
        n2 = [names[o] for o in perm]
        assert sorted((n2[c], w, sorted(n2[v] for v in mem)) for c, w, mem in got) == ref


@pytest.mark.parametrize("case", G7["cases"], ids=[c["name"] for c in G7["cases"]])
def test_partition_strings_with_the_oracle_search(case, monkeypatch):
    """Whole host path (graphs + partitions) with the CPU oracle standing in for the GPU NN search."""
    from isocon_amd import graphs, partitions
    from oracle import oracle as O
    monkeypatch.setattr(graphs, "nearest_neighbor_graph", O)
    S = dict(case["S"])
    G, partition, M, converged = partitions.partition_strings(S, Params())
    assert canon(unique_ids(S), partition, M, converged, G) == case["expect"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", G7["cases"], ids=[c["name"] for c in G7["cases"]])
def test_gpu_partition_strings(case):
    from isocon_amd import partitions
    S = dict(case["S"])
    G, partition, M, converged = partitions.partition_strings(S, Params())
    assert canon(unique_ids(S), partition, M, converged, G) == case["expect"]


G9 = json.load(open(os.path.join(HERE, "golden", "g9_partitions_2set.json")))


def canon2(G, partition):
    return {"partition": sorted([c, sorted(m)] for c, m in partition.items()), "edges": sorted([a, b] for a, b in G.edges()),
            "nodes": sorted(G.nodes())}


@pytest.mark.parametrize("case", G9["cases"], ids=[c["name"] for c in G9["cases"]])
def test_partition_strings_2set_with_the_oracle_search(case, monkeypatch):
    from isocon_amd import graphs, partitions
    from oracle import oracle as O
    monkeypatch.setattr(graphs, "nearest_neighbor_graph", O)
    G, partition = partitions.partition_strings_2set(dict(case["X"]), dict(case["C"]), None, None, Params())
    assert canon2(G, partition) == case["expect"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", G9["cases"], ids=[c["name"] for c in G9["cases"]])
def test_gpu_partition_strings_2set(case):
    from isocon_amd import partitions
    G, partition = partitions.partition_strings_2set(dict(case["X"]), dict(case["C"]), None, None, Params())
    assert canon2(G, partition) == case["expect"]
