#!/usr/bin/env python3
"""Golden fixture for the partition step: tests/golden/g7_partitions.json.

The REFERENCE's own modules/partitions.py::partition_strings (with modules/graphs.py and modules/
nearest_neighbor_graph.py) is imported from /root/reference and executed on each input under PYTHONHASHSEED = 0..7
(the reference iterates over sets of strings).  A case is kept only if all eight runs agree; the file
records how many did not.  edlib is absent here: tests/golden/shims/edlib.py forwards to the CPU oracle (distances
are pinned by definition, see make_golden.py).  Only inputs (synthetic reads, and sequences of the reference's public
test FASTA) and outputs are stored.

Usage:  python tests/golden/make_golden_partitions.py          (build container only)
"""
import contextlib
import io
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def cases():
    sys.path.insert(0, ROOT)
    from isocon_amd import synth
    from tests.golden.make_golden_inputs import read_fasta
    out = []
    fa = read_fasta(os.path.join(REF, "test", "data", "simulated_pacbio_reads_n_200.fa"))
    out.append(("test_data_n200", fa))
    for (name, n, L, iso, seed, dup) in (("synth_120x400_3iso", 120, 400, 3, 71, 0), ("synth_300x600_4iso_dups", 300, 600, 4, 72, 40),
                                         ("synth_60x300_2iso", 60, 300, 2, 73, 10)):
        accs, seqs, _ = synth.make_reads(n, L, iso, seed=seed)
        S = dict(zip(accs, seqs))
        # duplicates (weight > 1 nodes, "converged" strings): copy some sequences under new accessions
        for i in range(dup):
            S["dup_%d" % i] = seqs[(i * 7) % len(seqs)]
        out.append((name, S))
    # hand-made corner cases: mutual nearest neighbours, a chain, an isolated string, everything converged
    out.append(("mutual_and_chain", {"a": "ACGTACGTACGTAAAA", "b": "ACGTACGTACGTAAAT", "c": "ACGTACGTACGTAATT", "d": "ACGTACGTACGAATTT",
                                      "iso": "GGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGG", "e": "TTTTACGTTTTTGGGG", "f": "TTTTACGTTTTTGGGC"}))
    out.append(("all_converged", {"a1": "ACGTACGTAC", "a2": "ACGTACGTAC", "b1": "GGGTACGTAC", "b2": "GGGTACGTAC"}))
    return out


def child(case_index):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "shims"))
    sys.path.insert(0, REF)
    import networkx
    if not hasattr(networkx.Graph, "node"):  # the reference targets networkx <= 2.3
        networkx.Graph.node = property(lambda g: g.nodes)
    with contextlib.redirect_stdout(io.StringIO()):
        from modules import partitions as R_PART

    class Params(object):
        nr_cores = 1
        neighbor_search_depth = 2 ** 32
        verbose = False
        develop_logfile = None

    name, S = cases()[case_index]
    with contextlib.redirect_stdout(io.StringIO()):
        G, partition, M, converged = R_PART.partition_strings(S, Params())
    # sequences are written as indices into the list of unique strings (first appearance in S)
    uid = {}
    for seq in S.values():
        uid.setdefault(seq, len(uid))
    res = {"partition": sorted([uid[c], sorted(uid[x] for x in m)] for c, m in partition.items()), "M": sorted([uid[c], w] for c, w in M.items()),
           "converged": bool(converged), "nodes": sorted([uid[x], int(G.node[x]["degree"])] for x in G.nodes()),
           "edges": sorted([uid[a], uid[b], int(G[a][b]["edit_distance"])] for a, b in G.edges())}
    sys.stdout.write(json.dumps(res))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    all_cases = cases()
    kept, dropped = [], []
    for ci, (name, S) in enumerate(all_cases):
        outs = []
        for seed in range(8):
            env = dict(os.environ, PYTHONHASHSEED=str(seed))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(ci)], env=env, capture_output=True, text=True, check=True)
            outs.append(r.stdout)
        if all(o == outs[0] for o in outs):
            kept.append({"name": name, "S": [[a, s] for a, s in S.items()], "expect": json.loads(outs[0])})
        else:
            dropped.append(name)
        print(name, "agree" if all(o == outs[0] for o in outs) else "HASH-ORDER DEPENDENT", len(S), "strings")
    json.dump({"generator": "tests/golden/make_golden_partitions.py", "hash_seeds": list(range(8)), "hash_order_dependent_cases_dropped": dropped,
               "cases": kept}, open(os.path.join(HERE, "g7_partitions.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
