"""GPU parity: batched edit distance (isocon_ed_pairs) against the golden vectors and the CPU oracle.  Bit-exact."""
import random

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def _rs(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def _mut(rng, s, rate):
    out = []
    for c in s:
        r = rng.random()
        if r < rate * 0.3:
            continue
        if r < rate * 0.6:
            out.append(rng.choice("ACGT")); out.append(c); continue
        if r < rate:
            out.append(rng.choice("ACGT")); continue
        out.append(c)
    return "".join(out) or "A"


def test_golden_g1():
    from isocon_amd.store import SeqStore
    cases = golden("g1_edit_distance.json")["cases"]
    index, seqs, a, b, k, exp = {}, [], [], [], [], []
    for q, t, kk, e in cases:
        for s in (q, t):
            if s not in index:
                index[s] = len(seqs); seqs.append(s)
        a.append(index[q]); b.append(index[t]); k.append(kk); exp.append(e)
    st = SeqStore(seqs)
    got = st.ed_pairs(a, b, k)
    bad = [(i, cases[i][2], exp[i], int(got[i])) for i in range(len(exp)) if exp[i] != got[i]]
    assert not bad, bad[:10]


def test_random_pairs_vs_oracle():
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = random.Random(99)
    seqs, a, b, k = [], [], [], []
    for it in range(300):
        m = rng.choice([1, 2, 17, 63, 64, 65, 130, 400, 1000, 2500])
        base = _rs(rng, m)
        ia = len(seqs); seqs.append(base)
        for _ in range(rng.choice([1, 5, 70])):
            r = rng.random()
            if r < 0.6:
                t = _mut(rng, base, rng.choice([0.004, 0.02, 0.06, 0.2]))
            elif r < 0.8:
                t = _rs(rng, max(1, m + rng.randint(-30, 30)))
            else:
                cut = rng.randint(0, m - 1); ln = rng.randint(0, min(300, m - cut)); t = (base[:cut] + base[cut + ln:]) or "C"
            seqs.append(t)
            a.append(ia); b.append(len(seqs) - 1)
            k.append(rng.choice([-1, -1, 0, 5, 31, 63, 64, 100, 127, 128, 300, 511, 512, 3000]))
    st = SeqStore(seqs)
    got = st.ed_pairs(a, b, k)
    exp = O.ed_pairs(seqs, a, b, k)
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, [(int(i), len(seqs[a[i]]), len(seqs[b[i]]), k[i], int(exp[i]), int(got[i])) for i in bad[:10]]
    # swapped roles and explicit k=None (unbounded)
    got2 = st.ed_pairs(b, a, None)
    exp2 = O.ed_pairs(seqs, a, b, None)
    assert (got2 == exp2).all()


def test_long_sequences_unbanded():
    """rows > 4096 exercise the multi-pass path of the un-banded kernel."""
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = random.Random(5)
    s1, s2, s3 = _rs(rng, 9000), _rs(rng, 8700), _rs(rng, 4200)
    s4 = _mut(rng, s1, 0.1)
    seqs = [s1, s2, s3, s4]
    a, b = [0, 0, 1, 0, 2], [1, 2, 2, 3, 3]
    st = SeqStore(seqs)
    got = st.ed_pairs(a, b, None)
    exp = O.ed_pairs(seqs, a, b, None)
    assert got.tolist() == exp.tolist()


def test_other_symbols_are_symbols_of_their_own():
    """edlib's semantics (EAM:111): 'N' is just another character (tests/test_gpu_alphabet.py has the parity cases); alignments refuse."""
    from isocon_amd import _lib
    from isocon_amd.store import SeqStore
    st = SeqStore(["ACGT", "ACNT", "ACNT" * 30, "ACGT" * 30, "acgtN"])
    try:
        assert st.ed_pairs([0, 1, 2, 2, 0], [1, 1, 3, 2, 4], None).tolist() == [1, 0, 30, 0, 5]
        with pytest.raises(_lib.IsoconError):
            st.sg_trace([0], [1], -2)
    finally:
        st.close()


def test_empty_and_tiny():
    from isocon_amd.store import SeqStore
    st = SeqStore(["A", "C", "AC", "ACGT" * 20])
    assert st.ed_pairs([], [], None).tolist() == []
    assert st.ed_pairs([0, 0, 0, 2, 3], [0, 1, 2, 3, 3], None).tolist() == [0, 1, 1, 78, 0]
    assert st.ed_pairs([0, 2], [1, 3], [0, 10]).tolist() == [-1, -1]


def test_one_pair_per_lane_kernel_equals_the_oracle(monkeypatch):
    """k_ed_lanes (csrc/ed_lanes.hpp): scattered pairs, every threshold 0..63 and unbounded, lengths from 0 to 3 kb, both
    orientations of the length difference, pairs far beyond the threshold."""
    import random
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = random.Random(17)

    def mutate(s, e):
        s = list(s)
        for _ in range(e):
            r = rng.random()
            p = rng.randrange(len(s) + 1)
            if r < 0.4 and s:
                s[min(p, len(s) - 1)] = rng.choice("ACGT")
            elif r < 0.7:
                s.insert(p, rng.choice("ACGT"))
            elif s:
                del s[min(p, len(s) - 1)]
        return "".join(s)

    seqs = ["", "A", "C", "ACGT", "ACGTACGTAC"]
    for L in (30, 63, 64, 65, 130, 700, 1500, 3000):
        base = "".join(rng.choice("ACGT") for _ in range(L))
        seqs.append(base)
        for e in (0, 1, 2, 5, 17, 40, 62, 63, 64, 90):
            seqs.append(mutate(base, e))
    n = len(seqs)
    a = np.array([rng.randrange(n) for _ in range(6000)], dtype=np.uint32)
    b = np.array([rng.randrange(n) for _ in range(6000)], dtype=np.uint32)
    # bias towards related pairs (same base block)
    for i in range(0, 6000, 2):
        blk = 5 + 11 * rng.randrange(8)
        a[i] = blk + rng.randrange(11); b[i] = blk + rng.randrange(11)
    k = np.array([rng.choice([-1, -1, 0, 1, 2, 3, 7, 20, 31, 32, 33, 50, 62, 63]) for _ in range(6000)], dtype=np.int32)
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "ed_lanes=1")
    st = SeqStore(seqs)
    try:
        got = st.ed_pairs(a, b, k)
        got_u = st.ed_pairs(a, b, None)
    finally:
        st.close()
    want = O.ed_pairs(seqs, a, b, k)
    want_u = O.ed_pairs(seqs, a, b, None)
    assert (got == want).all(), np.nonzero(got != want)[0][:10]
    assert (got_u == want_u).all()
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "ed_lanes=0")
    st = SeqStore(seqs)
    try:
        assert (st.ed_pairs(a, b, k) == want).all()
    finally:
        st.close()
