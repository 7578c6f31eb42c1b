"""get_partition_alignments (distances -> alignments -> exon filter -> correction input) against outputs of
the reference's own isocon_get_candidates.py (tests/golden/g8_partition_alignments.json).  GPU test: the whole chain
partition_strings -> get_partition_alignments through the HIP kernels."""
import hashlib
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G7 = {c["name"]: c for c in json.load(open(os.path.join(HERE, "golden", "g7_partitions.json")))["cases"]}
G8 = json.load(open(os.path.join(HERE, "golden", "g8_partition_alignments.json")))["cases"]


class Params(object):
    nr_cores = 1
    neighbor_search_depth = 2 ** 32
    verbose = False
    develop_logfile = None
    min_exon_diff = 20
    ignore_ends_len = 15


def sha(s):
    return hashlib.sha1(s.encode()).hexdigest()[:16]


def canon(S, pa, exon_filtered):
    uid = {}
    for seq in S.values():
        uid.setdefault(seq, len(uid))
    rows = sorted([uid[m], uid[s], int(t[0]), sha(t[1]), sha(t[2]), int(t[3])] for m in pa for s, t in pa[m].items())
    return {"rows": rows, "exon_filtered": sorted(uid[s] for s in exon_filtered)}


def test_unique_seq_accessions():
    from isocon_amd import isocon_get_candidates as IGC
    assert IGC.get_unique_seq_accessions({"a": "AC", "b": "GG", "c": "AC"}) == {"AC": ["a", "c"], "GG": ["b"]}


@pytest.mark.gpu
@pytest.mark.parametrize("case", G8, ids=[c["name"] for c in G8])
def test_gpu_partition_alignments(case):
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import partitions
    S = dict(G7[case["name"]]["S"])
    G, partition, M, converged = partitions.partition_strings(S, Params())
    exon_filtered = set()
    pa = IGC.get_partition_alignments(partition, M, G, exon_filtered, Params())
    got = canon(S, pa, exon_filtered)
    assert got["exon_filtered"] == case["expect"]["exon_filtered"]
    assert [r[:3] + r[5:] for r in got["rows"]] == [r[:3] + r[5:] for r in case["expect"]["rows"]]     # pairs, edit distances, weights
    assert got["rows"] == case["expect"]["rows"]                                                       # gapped strings (tie policy 0)


@pytest.mark.parametrize("case", [c for c in G8 if c["name"] != "synth_300x600_4iso_dups"], ids=[c["name"] for c in G8 if c["name"] != "synth_300x600_4iso_dups"])
def test_partition_alignments_with_the_oracle_kernels(case, monkeypatch):
    """Host logic of the chain on CPU: the oracle stands in for the three device entry points."""
    from isocon_amd import graphs
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import partitions
    from oracle import oracle as O
    monkeypatch.setattr(graphs, "nearest_neighbor_graph", O)
    monkeypatch.setattr(IGC, "edlib_align_sequences", O.edlib_align_sequences)
    monkeypatch.setattr(IGC, "sw_align_sequences", O.sw_align_sequences)
    S = dict(G7[case["name"]]["S"])
    G, partition, M, converged = partitions.partition_strings(S, Params())
    exon_filtered = set()
    pa = IGC.get_partition_alignments(partition, M, G, exon_filtered, Params())
    assert canon(S, pa, exon_filtered) == case["expect"]
