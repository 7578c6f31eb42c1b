#!/usr/bin/env python3
"""Golden fixture tests/golden/g14_all_nn.json: the reference's own modules/end_invariant_functions.py::
get_NN_graph_ignored_ends_edlib (-> get_all_NN_under_ignored_edge_ends -> get_all_NN -> edlib_traceback, HW mode) on
crafted candidate sets -- variants of a few base sequences with substitutions, indels, truncated and extended ends --
for ignore_ends_len in {0, 5, 15}, serial and through its Pool, plus single edlib_traceback calls.  What is pinned:
the window rule, the sticky stops, the end arithmetic, the 0..10 filter, insertion order and the symmetrisation.
edlib itself is absent: its HW mode is the stand-in tests/golden/shims/edlib.py -> oracle ("parity unpinned" for the
choice among equally good locations / paths).  Build container only."""
import contextlib
import io
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def mutate(rng, b):
    v = list(b)
    for _ in range(rng.choice([0, 0, 1, 2, 3, 6, 12])):
        p = rng.randrange(len(v))
        r = rng.random()
        if r < 0.5:
            v[p] = rng.choice("ACGT")
        elif r < 0.75:
            del v[p]
        else:
            v.insert(p, rng.choice("ACGT"))
    v = "".join(v)
    cut_l, cut_r = rng.choice([0, 0, 3, 8, 16, 22]), rng.choice([0, 0, 3, 8, 16, 22])
    v = v[cut_l:len(v) - cut_r]
    if rng.random() < 0.3:
        v = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 20))) + v
    if rng.random() < 0.3:
        v = v + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 20)))
    return v


def cases():
    rng = random.Random(14)
    out = []
    for ci, (thr, cores, depth) in enumerate([(15, 1, 2 ** 32), (0, 1, 2 ** 32), (5, 1, 2 ** 32), (15, 2, 2 ** 32), (15, 1, 3), (5, 1, 1)]):
        base = ["".join(rng.choice("ACGT") for _ in range(rng.randint(150, 330))) for _ in range(3)]
        base.append(base[0][:60] + base[0][95:])            # an exon-sized difference: never an edge
        C = {}
        for b in base:
            for v in [b] + [mutate(rng, b) for _ in range(rng.randint(5, 9))]:
                if v not in C.values():
                    C["transcript_%d_support_%d" % (len(C), rng.randint(1, 9))] = v
        out.append({"name": "crafted_%d_ends%d_cores%d_depth%s" % (ci, thr, cores, "inf" if depth > 10 ** 6 else depth), "C": C,
                    "ignore_ends_len": thr, "nr_cores": cores, "neighbor_search_depth": depth})
    return out


def pairs():
    rng = random.Random(15)
    out = []
    for _ in range(40):
        b = "".join(rng.choice("ACGT") for _ in range(rng.randint(60, 200)))
        out.append((mutate(rng, b), mutate(rng, b), rng.choice([10, 15, 25]), rng.choice([0, 5, 15])))
    out += [("ACGTACGTACGTACGTACGT", "TTACGTACGTACGTACGTACGTGG", 5, 1), ("GGGACGTACGTACGTAAA", "ACGTACGTACGT", 10, 2),
            ("ACGTACGTACGT", "ACGTACGTACGT", 0, 0), ("AAAAAAAAAAAAAAAA", "CCCCCCCCCCCCCCCCCC", 5, 5)]
    return out


def main():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "shims"))
    sys.path.insert(0, REF)
    import networkx
    if not hasattr(networkx.Graph, "node"):
        networkx.Graph.node = property(lambda g: g.nodes)
    with contextlib.redirect_stdout(io.StringIO()):
        from modules import end_invariant_functions as R_END
    kept = []
    for c in cases():
        class Params(object):
            ignore_ends_len = c["ignore_ends_len"]
            nr_cores = c["nr_cores"]
            neighbor_search_depth = c["neighbor_search_depth"]
            verbose = False
        with contextlib.redirect_stdout(io.StringIO()):
            g = R_END.get_NN_graph_ignored_ends_edlib(dict(c["C"]), Params())
        c = dict(c, C=[[a, s] for a, s in c["C"].items()], expect=[[a, list(nb.items())] for a, nb in g.items()])
        kept.append(c)
        print(c["name"], len(c["C"]), "candidates,", sum(len(nb) for _, nb in c["expect"]), "directed edges, sum ed", sum(e for _, nb in c["expect"] for _, e in nb))
    tb = [[x, y, k, t, R_END.edlib_traceback(x, y, mode="HW", task="path", k=k, end_threshold=t)] for x, y, k, t in pairs()]
    print("edlib_traceback:", [r[4] for r in tb])
    json.dump({"generator": "tests/golden/make_golden_all_nn.py", "cases": kept, "edlib_traceback": tb},
              open(os.path.join(HERE, "g14_all_nn.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
