#!/usr/bin/env python3
"""Golden fixture tests/golden/g9_partitions_2set.json: the reference's own modules/partitions.py::partition_strings_2set
(greedy read -> candidate assignment on the bipartite nearest-neighbour graph, partitions.py:595-647 with
graphs.py:150-160) on synthetic read / candidate sets, under PYTHONHASHSEED 0..3 (kept only if all agree).
edlib is absent: tests/golden/shims/edlib.py forwards to the CPU oracle.

Usage:  python tests/golden/make_golden_partitions_2set.py          (build container only)
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
    import numpy as np
    from isocon_amd import synth
    out = []
    for (name, n, L, iso, seed, extra) in (("reads150_cands", 150, 400, 3, 91, 4), ("reads300_cands_ties", 300, 500, 4, 92, 8), ("reads40_two_cands", 40, 200, 2, 93, 0)):
        accs, seqs, isoforms = synth.make_reads(n, L, iso, seed=seed)
        rng = np.random.default_rng(seed)
        cands = list(isoforms)
        prof = dict(synth.CCS_PROFILE, rate=0.01)
        for i in range(extra):            # near-duplicates of the true isoforms: ties between candidates
            base = np.frombuffer(isoforms[i % len(isoforms)].encode(), dtype=np.uint8)
            cands.append(synth.mutate(rng, base, prof).tobytes().decode())
        X = dict(zip(accs, seqs))
        C = {"cand_%d" % i: c for i, c in enumerate(dict.fromkeys(cands))}
        out.append((name, X, C))
    # hand-made: reads at equal distance from two candidates, a candidate nobody prefers, equal degrees (name decides)
    c1, c2, c3 = "AAAAACCCCCGGGGGTTTTTACGT", "AAAAACCCCCGGGGGTTTTAACGT", "TTTTTGGGGGCCCCCAAAAATGCA"
    X = {"r_tie1": "AAAAACCCCCGGGGGTTTTACGT", "r_tie2": "AAAAACCCCCGGGGGTTTTCACGT", "r_c1": "AAAAACCCCCGGGGGTTTTTACG", "r_c2a": "AAAAACCCCCGGGGGTTTTAACG",
         "r_c2b": "AAAACCCCCGGGGGTTTTAACGT", "r_c3": "TTTTTGGGGGCCCCCAAAAATGC"}
    out.append(("ties_between_candidates", X, {"cand_b": c1, "cand_a": c2, "cand_c": c3}))
    return out


def child(ci):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "shims"))
    sys.path.insert(0, REF)
    import networkx
    if not hasattr(networkx.Graph, "node"):
        networkx.Graph.node = property(lambda g: g.nodes)
    with contextlib.redirect_stdout(io.StringIO()):
        from modules import partitions as R_PART

    class Params(object):
        nr_cores = 1
        neighbor_search_depth = 2 ** 32
        verbose = False
        develop_logfile = None

    name, X, C = cases()[ci]
    with contextlib.redirect_stdout(io.StringIO()):
        G, partition = R_PART.partition_strings_2set(X, C, None, None, Params())
    sys.stdout.write(json.dumps({"partition": sorted([c, sorted(m)] for c, m in partition.items()),
                                 "edges": sorted([a, b] for a, b in G.edges()), "nodes": sorted(G.nodes())}))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    kept, dropped = [], []
    for ci, (name, X, C) in enumerate(cases()):
        outs = []
        for seed in range(4):
            env = dict(os.environ, PYTHONHASHSEED=str(seed))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(ci)], env=env, capture_output=True, text=True, check=True)
            outs.append(r.stdout)
        agree = all(o == outs[0] for o in outs)
        if agree:
            kept.append({"name": name, "X": [[a, s] for a, s in X.items()], "C": [[a, s] for a, s in C.items()], "expect": json.loads(outs[0])})
        else:
            dropped.append(name)
        e = json.loads(outs[0])
        print(name, "agree" if agree else "HASH-ORDER DEPENDENT", len(X), "reads", len(C), "candidates ->", len(e["partition"]), "partitions", len(e["edges"]), "edges")
    json.dump({"generator": "tests/golden/make_golden_partitions_2set.py", "hash_order_dependent_cases_dropped": dropped, "cases": kept},
              open(os.path.join(HERE, "g9_partitions_2set.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
