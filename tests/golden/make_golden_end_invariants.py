#!/usr/bin/env python3
"""Golden fixture tests/golden/g13_end_invariants.json: the reference's own
modules/end_invariant_functions.py::collapse_candidates_under_ends_invariant and ::is_overlap on crafted candidate sets
(containment and suffix-prefix overlaps at, below and above the end threshold; supports that decide the kept candidate;
ties decided by the accession), under PYTHONHASHSEED 0..3.  Build container only."""
import contextlib
import io
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def cases():
    rng = random.Random(5)
    out = []
    for ci in range(6):
        base = ["".join(rng.choice("ACGT") for _ in range(rng.randint(120, 260))) for _ in range(4)]
        C, sup = {}, {}
        k = 0
        for b in base:
            variants = [b]
            for _ in range(rng.randint(2, 6)):
                cut_l, cut_r = rng.randint(0, 22), rng.randint(0, 22)
                v = b[cut_l:len(b) - cut_r]
                r = rng.random()
                if r < 0.3:
                    v = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 18))) + v     # different start, shared rest
                elif r < 0.5:
                    v = v + "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 18)))
                elif r < 0.6:
                    p = rng.randrange(len(v)); v = v[:p] + rng.choice("ACGT") + v[p + 1:]      # an internal difference
                variants.append(v)
            for v in variants:
                if v and v not in C.values():
                    acc = "transcript_%d_support_%d" % (k, rng.choice([1, 2, 2, 3, 7])); k += 1
                    C[acc] = v; sup[acc] = int(acc.rsplit("_", 1)[1])
        out.append(("crafted_%d" % ci, C, sup, rng.choice([0, 5, 15, 15, 20])))
    return out


OVERLAPS = [("ACGTACGTAC", "GTACGTACGG", 3), ("ACGTACGTAC", "GTACGTACGG", 1), ("AAAACCCC", "CCCCGGGG", 4), ("AAAACCCC", "CCCCGGGG", 3),
            ("ACGT", "ACGT", 0), ("", "ACGT", 5), ("ACGTTT", "ACG", 2), ("TTACGT", "ACGTAA", 2), ("TTACGT", "ACGTAA", 1), ("ACGTACGT", "TTTTTTTT", 15)]


def wide_overlaps():
    """thresholds above 16 with the two lengths more than thr + 16 apart (the region where an end index of the restatement's probe search
    went negative), and ordinary near misses at those thresholds"""
    rng = random.Random(77)
    out = [("".join(random.Random(3).choice("ACGT") for _ in range(400)),) * 2]
    s = out.pop()[0]
    out.append((s, s[60:] + "A", 40))                     # (the advisor's case: prefix offset 60 > 40)
    for thr in (17, 25, 40, 64):
        for _ in range(12):
            L = rng.randint(3 * thr + 20, 420)
            a = "".join(rng.choice("ACGT") for _ in range(L))
            gap = rng.choice([rng.randint(thr + 17, 2 * thr), rng.randint(0, thr), thr, thr + 1, thr + 16, thr + 17])
            gap = min(gap, L - 2)
            tail = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, thr + 2)))
            b = a[gap:] + tail
            if rng.random() < 0.25:
                k = rng.randrange(len(b)); b = b[:k] + rng.choice("ACGT") + b[k + 1:]
            out.append((a, b, thr) if rng.random() < 0.7 else (b, a, thr))
    return out


OVERLAPS = OVERLAPS + wide_overlaps()


def child(ci):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "shims"))
    sys.path.insert(0, REF)
    import networkx
    if not hasattr(networkx.Graph, "node"):
        networkx.Graph.node = property(lambda g: g.nodes)
    with contextlib.redirect_stdout(io.StringIO()):
        from modules import end_invariant_functions as R_END
    if ci < 0:
        res = []
        for a, b, t in OVERLAPS:
            r = R_END.is_overlap(a, b, t)
            res.append(bool(r))
        sys.stdout.write(json.dumps(res))
        return
    name, C, sup, thr = cases()[ci]

    class Params(object):
        ignore_ends_len = thr
        verbose = False

    with contextlib.redirect_stdout(io.StringIO()):
        part = R_END.collapse_candidates_under_ends_invariant(dict(C), dict(sup), Params())
    sys.stdout.write(json.dumps(sorted([c, sorted(m)] for c, m in part.items())))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    kept, dropped = [], []
    for ci, (name, C, sup, thr) in enumerate(cases()):
        outs = []
        for seed in range(4):
            env = dict(os.environ, PYTHONHASHSEED=str(seed))
            outs.append(subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(ci)], env=env, capture_output=True, text=True, check=True).stdout)
        agree = all(o == outs[0] for o in outs)
        e = json.loads(outs[0])
        (kept if agree else dropped).append({"name": name, "C": [[a, s] for a, s in C.items()], "support": sup, "ignore_ends_len": thr, "expect": e} if agree else name)
        print(name, "agree" if agree else "HASH-ORDER DEPENDENT", len(C), "candidates ->", len(e), "kept, threshold", thr)
    ov = json.loads(subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "-1"], capture_output=True, text=True, check=True).stdout)
    json.dump({"generator": "tests/golden/make_golden_end_invariants.py", "hash_order_dependent_cases_dropped": dropped, "cases": kept,
               "is_overlap": [[a, b, t, r] for (a, b, t), r in zip(OVERLAPS, ov)]}, open(os.path.join(HERE, "g13_end_invariants.json"), "w"), indent=0)
    print("is_overlap", ov)


if __name__ == "__main__":
    main()
