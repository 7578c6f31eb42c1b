"""The q-gram count bound the NN main pass relies on (isocon_amd/csrc/qgram_mm.hpp), checked on the CPU against the oracle's exact
distances: ceil((L1 + |sum difference|) / 2q) never exceeds the edit distance -- for random pairs, related pairs with every
kind of edit, homopolymers, repeats, saturating counts and sequences shorter than q."""
import random

import numpy as np

from oracle import oracle as O

Q = 6


def profile(s, cap=255, q=Q, bins=None):
    """counts of the q-grams; bins: hash the 4^q gram codes into that many bins (the kernel's choice: q = 8, 6144 bins)"""
    code = np.zeros(256, np.int64)
    code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
    c = code[np.frombuffer(s.encode(), np.uint8)]
    ng = len(c) - q + 1
    nb = bins or 4 ** q
    if ng <= 0:
        return np.zeros(nb, np.int64)
    idx = np.zeros(ng, np.int64)
    for i in range(q):
        idx |= (c[i:i + ng] & 1) << i
        idx |= (c[i:i + ng] >> 1) << (q + i)
    if bins:
        idx = (((idx * 0x9E3779B1) & 0xffffffff) >> 7) % bins
    return np.minimum(np.bincount(idx, minlength=nb), cap)


def bound(pa, pb, q=Q):
    return int((np.abs(pa - pb).sum() + abs(int(pa.sum()) - int(pb.sum())) + 2 * q - 1) // (2 * q))


def _edits(rng, s, e, kinds="sid"):
    s = list(s)
    for _ in range(e):
        k = rng.choice(kinds)
        p = rng.randrange(len(s) + 1)
        if k == "s" and s:
            s[min(p, len(s) - 1)] = rng.choice("ACGT")
        elif k == "i":
            s.insert(p, rng.choice("ACGT"))
        elif k == "d" and s:
            del s[min(p, len(s) - 1)]
    return "".join(s)


def test_bound_never_exceeds_the_edit_distance():
    rng = random.Random(23)
    seqs = ["", "A", "ACGTA", "ACGTAC", "A" * 400, "A" * 430, "AC" * 300, "ACG" * 200, "T" * 300 + "ACGT" * 50]
    for L in (40, 200, 800):
        base = "".join(rng.choice("ACGT") for _ in range(L))
        seqs.append(base)
        for kinds in ("s", "i", "d", "sid"):
            for e in (1, 2, 5, 20, 60):
                seqs.append(_edits(rng, base, e, kinds))
        seqs.append(base[:L // 2] + base[L // 2 + 30:])                # a 30-base deletion
        seqs.append(base[:L // 3] + "G" * 25 + base[L // 3:])          # a 25-base homopolymer insertion
        seqs.append(base[L // 2:] + base[:L // 2])                     # rotation: same grams, large distance
    n = len(seqs)
    a = np.array([rng.randrange(n) for _ in range(2500)], dtype=np.uint32)
    b = np.array([rng.randrange(n) for _ in range(2500)], dtype=np.uint32)
    d = O.ed_pairs(seqs, a, b, None)
    prof = [profile(s) for s in seqs]
    lb = np.array([bound(prof[i], prof[j]) for i, j in zip(a, b)])
    assert (lb <= d).all(), [(seqs[a[i]][:30], seqs[b[i]][:30], int(lb[i]), int(d[i])) for i in np.nonzero(lb > d)[0][:3]]
    assert (lb[a == b] == 0).all()
    # saturating counts (cap 3 here) and merged bins keep it a bound
    prof3 = [profile(s, cap=3) for s in seqs]
    lb3 = np.array([bound(prof3[i], prof3[j]) for i, j in zip(a, b)])
    assert (lb3 <= d).all()
    merged = [p.reshape(1024, 4).sum(axis=1) for p in prof]
    lbm = np.array([bound(merged[i], merged[j]) for i, j in zip(a, b)])
    assert (lbm <= d).all() and (lbm <= lb).all()
    # hashed bins: 8-grams into 6144 bins
    prof8 = [profile(s, q=8, bins=6144) for s in seqs]
    lb8 = np.array([bound(prof8[i], prof8[j], q=8) for i, j in zip(a, b)])
    assert (lb8 <= d).all(), [(seqs[a[i]][:30], seqs[b[i]][:30], int(lb8[i]), int(d[i])) for i in np.nonzero(lb8 > d)[0][:3]]
    assert (lb8[a == b] == 0).all()
    # the kernel's stored vector (qgram_mm.hpp): presence bits of hashed 9-grams + capped excess counts in coarser bins -- with the
    # kernel's parameters and with tiny bin counts that force every merge / cap case
    import qgram_ref as R
    for kw in ({}, {"q": 4, "b0": 64, "b1": 16, "cap": 2}, {"q": 3, "b0": 64, "b1": 64, "cap": 1}, {"q": 5, "b0": 256, "b1": 32, "cap": 3}):
        pk = [R.profile(s, **kw) for s in seqs]
        qq = kw.get("q", R.Q)
        lbk = np.array([R.bound(pk[i], pk[j], qq) for i, j in zip(a, b)])
        assert (lbk <= d).all(), (kw, [(seqs[a[i]][:30], seqs[b[i]][:30], int(lbk[i]), int(d[i])) for i in np.nonzero(lbk > d)[0][:3]])
        assert (lbk[a == b] == 0).all()
    # and it is not vacuous (short sequences with dense edits included; 0.76 on 2.5 kb reads at 1 % errors)
    rel = (d > 0) & (d <= 60)
    assert np.median(lb[rel] / d[rel]) > 0.4


def test_greedy_block_count_is_a_lower_bound():
    """The second bound of the main pass (isocon_amd/csrc/nn_filter.hpp; restated in tests/qgram_ref.block_count): the greedy number of pairwise
    disjoint b-grams of one sequence that occur nowhere in the other never exceeds the edit distance -- every probe stride, random pairs, related
    pairs with every kind of edit (also clustered ones), homopolymers, repeats, sequences shorter than a gram or a word."""
    from tests import qgram_ref
    rng = random.Random(77)
    cases = []
    for _ in range(300):
        L = rng.choice([5, 19, 20, 21, 36, 64, 100, 257, 700])
        alpha = rng.choice(["ACGT", "ACGT", "AC", "A"])
        a = "".join(rng.choice(alpha) for _ in range(L))
        kind = rng.random()
        if kind < 0.6:
            b = _edits(rng, a, rng.randint(0, 12), rng.choice(["sid", "i", "d", "s", "id"]))
        elif kind < 0.8:          # a burst of edits inside a few bases, and another one close behind
            s = list(a)
            p = rng.randrange(max(1, len(s) - 12))
            for j in range(rng.randint(1, 5)):
                s.insert(p + j, rng.choice("ACGT"))
            if len(s) > p + 9:
                del s[p + 8]
            b = "".join(s)
        else:
            b = "".join(rng.choice(alpha) for _ in range(max(1, L + rng.randint(-6, 6))))
        cases.append((a, b or "A"))
    cases += [("ACGT" * 40, "ACGT" * 39 + "ACG"), ("A" * 100, "A" * 99 + "C"), ("ACGTTGCA" * 12, "TGCAACGT" * 12), ("", "ACGT" * 8), ("ACGT" * 8, "")]
    tight = 0
    for a, b in cases:
        d = O.ed_bounded(a, b) if a and b else max(len(a), len(b))
        for gram, stride in ((8, 4), (8, 2), (8, 1), (4, 4), (4, 2), (12, 4)):
            for x, y in ((a, b), (b, a)):
                c = qgram_ref.block_count(x, y, b=gram, s=stride)
                assert c <= d, (x, y, gram, stride, c, d)
        tight += qgram_ref.block_count(a, b) == d and d > 0
    assert tight > 10          # (and it is not a trivial bound: it meets the distance on some pairs)
