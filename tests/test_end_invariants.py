"""End-invariant collapse of candidates against outputs of the reference's own end_invariant_functions
(tests/golden/g13_end_invariants.json)."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G13 = json.load(open(os.path.join(HERE, "golden", "g13_end_invariants.json")))


@pytest.mark.parametrize("row", G13["is_overlap"], ids=[str(i) for i in range(len(G13["is_overlap"]))])
def test_is_overlap(row):
    from isocon_amd import end_invariant_functions as END
    a, b, t, expect = row
    assert bool(END.is_overlap(a, b, t)) == expect


@pytest.mark.parametrize("case", G13["cases"], ids=[c["name"] for c in G13["cases"]])
def test_collapse_candidates(case):
    from isocon_amd import end_invariant_functions as END

    class Params(object):
        ignore_ends_len = case["ignore_ends_len"]
        verbose = False

    part = END.collapse_candidates_under_ends_invariant(dict(case["C"]), dict(case["support"]), Params())
    assert sorted([c, sorted(m)] for c, m in part.items()) == case["expect"]


def test_anchor_index_finds_every_pair_the_quadratic_scan_finds():
    """The k-mer anchor index only proposes pairs; the graph must equal the one from testing every pair in the window."""
    import random
    from isocon_amd import end_invariant_functions as END
    rng = random.Random(1)
    base = ["".join(rng.choice("ACGT") for _ in range(400 + rng.randint(-40, 40))) for _ in range(4)]
    C, sup = {}, {}
    for i in range(400):
        b = list(base[i % 4])
        for _ in range(rng.randint(0, 3)):
            b[rng.randrange(len(b))] = rng.choice("ACGT")
        b = "".join(b)
        r = rng.random()
        if r < 0.4:
            b = b[rng.randint(0, 25):len(b) - rng.randint(0, 25)]
        elif r < 0.6:
            b = "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 20))) + b[rng.randint(0, 20):]
        elif r < 0.7:
            b = b[rng.randint(0, 20):] + "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 20)))
        if b not in C.values():
            C["t_%d" % i] = b
            sup["t_%d" % i] = 1 + i % 3

    class P(object):
        ignore_ends_len = 15
        verbose = False

    G = END.get_invariants_under_ignored_edge_ends_speed(C, sup, P())
    thr, E = 15, set()
    by_len = sorted(C.items(), key=lambda x: len(x[1]))
    for a1, s1 in by_len:
        for a2, s2 in by_len:
            if a2 == a1 or len(s2) < len(s1) - 2 * thr:
                continue
            if len(s2) > len(s1):
                break
            if END._pair_is_invariant(s1, s2, thr):
                E.add((a1, a2)); E.add((a2, a1))
    assert set(G.edges()) == E and len(E) > 100
