"""GPU: edge cases of the hot-path entry points (empty / singleton / all-converged inputs, duplicates, tiny strings)."""
import numpy as np
import pytest

from conftest import Params, ordered

pytestmark = pytest.mark.gpu


def test_nn_graph_degenerate_inputs():
    from isocon_amd import nearest_neighbor_graph as NNG
    from oracle import oracle as O
    p = Params(1)
    for S in ({}, {"a": "ACGTACGT"}, {"a": "ACGT", "b": "ACGT"}, {"a": "A", "b": "C", "c": "AC"}):
        if not S:
            continue   # the reference itself is never called with an empty dict (graphs.py:56-58 guards)
        g, iso = NNG.compute_nearest_neighbor_graph(dict(S), set(), p)
        ge, isoe = O.compute_nearest_neighbor_graph(dict(S), set(), p)
        assert ordered(g) == ordered(ge) and iso == isoe
    S = {"r%d" % i: "ACGTTGCA" * 5 + "A" * i for i in range(6)}
    allc = set(S.values())
    g, iso = NNG.compute_nearest_neighbor_graph(S, allc, p)          # every entry converged: all rows empty
    assert ordered(g) == ordered(O.compute_nearest_neighbor_graph(S, allc, p)[0])
    assert all(v == {} for v in g.values())


def test_nn_graph_distance_equal_to_length_is_admitted():
    """best_ed starts at len(seq1) and the equality branch admits d == len(seq1) (NNG:129,161)."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from oracle import oracle as O
    S = {"a": "AAAA", "b": "CCCC", "c": "GGGGGG"}
    g, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    assert ordered(g) == ordered(O.compute_nearest_neighbor_graph(S, set(), Params(1))[0])
    assert g["a"] == {"b": 4}


def test_2set_duplicate_sequences_and_no_candidates_in_reach():
    from isocon_amd import nearest_neighbor_graph as NNG
    from oracle import oracle as O
    X = {"r1": "ACGTACGTAC", "r2": "ACGTACGTAC", "r3": "TTTT"}
    C = {"c1": "ACGTACGTAC", "c2": "ACGTACGTAC", "c3": "ACGTACGAAC"}
    g = NNG.compute_2set_nearest_neighbor_graph(X, C, Params(1))
    assert ordered(g) == ordered(O.compute_2set_nearest_neighbor_graph(X, C, Params(1)))
    assert g["r1"] == {"c1": 0, "c2": 0} or set(g["r1"]) == {"c1", "c2"}
    assert g["r3"] == {}


def test_wrappers_with_empty_inputs():
    from isocon_amd import SW_alignment_module as SWM
    from isocon_amd import edlib_alignment_module as EAM
    assert EAM.edlib_align_sequences({"ACGT": []}) == {}
    assert EAM.edlib_align_sequences_keeping_accession({}) == {}
    assert SWM.sw_align_sequences({}) == {}
    assert SWM.sw_align_sequences({"ACGT": {}}) == {}
    assert SWM.sw_align_sequences_keeping_accession({"a": {}}) == {}


def test_sw_tiny_and_asymmetric_lengths():
    from isocon_amd import SW_alignment_module as SWM
    from oracle import oracle as O
    cases = [("A", "A"), ("A", "C"), ("A", "ACGTACGT"), ("ACGTACGT", "G"), ("ACGT" * 200, "ACGT"), ("AC", "ACGT" * 150)]
    for s1, s2 in cases:
        for mm in (-1, -4):
            got = SWM.parasail_alignment(s1, s2, 0, 0, mismatch_penalty=mm)
            assert got == O.parasail_alignment(s1, s2, 0, 0, mismatch_penalty=mm), (s1[:10], s2[:10], mm)


def test_hit_list_overflow_restart_is_transparent():
    """Many exact ties (identical distances everywhere) make the device hit list overflow its first capacity."""
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    base = "".join(rng.choice(list("ACGT"), 200))
    seqs = []
    for i in range(200):                     # every pair of these is at distance 2 (two distinct substitution sites)
        s = list(base); s[i] = "A" if s[i] != "A" else "C"; seqs.append("".join(s))
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    best, rp, cols, stats = st.nn_graph()
    row_ptr, c, e, _ = O.nn_1set(seqs, np.zeros(len(seqs), np.uint8), 0, len(seqs))
    assert rp.tolist() == row_ptr.tolist() and cols.tolist() == c.tolist()
    assert (best == 2).all()


@pytest.mark.gpu
def test_store_digest_is_order_sensitive():
    """isocon_store_digest (the identity the ranks of a sharded run compare): equal for equal stores, different as soon as
    two equal-length sequences swap places or one base changes -- also where the old sampled fingerprint could not see it."""
    import random
    from isocon_amd.store import SeqStore
    rng = random.Random(8)
    seqs = sorted(("".join(rng.choice("ACGT") for _ in range(rng.choice([90, 90, 91, 200, 200, 350]))) for _ in range(700)), key=len)
    a, b = SeqStore(seqs), SeqStore(list(seqs))
    assert a.fingerprint == b.fingerprint and 0 <= a.fingerprint < 2 ** 63
    swapped = list(seqs)
    i = next(i for i in range(1, len(seqs) - 1) if len(seqs[i]) == len(seqs[i + 1]) and i % max(1, len(seqs) // 64) != 0 and (i + 1) % max(1, len(seqs) // 64) != 0)
    swapped[i], swapped[i + 1] = swapped[i + 1], swapped[i]
    assert SeqStore(swapped).fingerprint != a.fingerprint
    changed = list(seqs)
    changed[5] = changed[5][:-1] + ("A" if changed[5][-1] != "A" else "C")
    assert SeqStore(changed).fingerprint != a.fingerprint


@pytest.mark.gpu
def test_symbols_outside_acgt_are_served_by_distances_and_refused_by_alignments():
    """isocon_store_create packs on the device (k_pack_planes).  A set with symbols outside ACGT is a valid store for distances (edlib
    compares any characters; tests/test_gpu_alphabet.py has the parity cases); the alignment entry points refuse it.  Empty sequences and
    an empty set are fine."""
    from isocon_amd import _lib
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    good = ["ACGT" * 40, "TTGCA" * 30, "G" * 70]
    SeqStore(good).close()
    for seqs in (good + ["ACGT" * 20 + "N" + "ACGT" * 20], ["ACGT" * 40, "ACGTacgt", "NNNN"], ["A" * 200 + "X" + "A" * 30 + "-", "ACGT"], ["ACGT", "AC GT"]):
        st = SeqStore(seqs)
        try:
            a = [i for i in range(len(seqs)) for _ in seqs]
            b = [j for _ in seqs for j in range(len(seqs))]
            assert st.ed_pairs(a, b, None).tolist() == [O.ed_dp(seqs[x], seqs[y]) for x, y in zip(a, b)]
            with pytest.raises(_lib.IsoconError) as e:
                st.sg_trace([0], [1], -2)
            assert "ACGT" in str(e.value)
        finally:
            st.close()
    st = SeqStore(["", "ACGT", ""])
    assert list(st.ed_pairs([0, 0, 1], [1, 2, 2])) == [4, 0, 4]
    st.close()


@pytest.mark.gpu
def test_sets_over_another_alphabet_of_at_most_four_symbols():
    """edlib takes any characters (EAM:111, NNG:105): a set in lower case, an RNA set, a two-symbol set and a set that mixes cases within
    four symbols are packed under their own symbol map; distances equal the textbook DP on the bytes (oracle orc_ed_dp) and the NN graph
    equals the oracle loop.  The alignment entry points refuse such a store (parasail's matrix is over "ACGT", SWM:65)."""
    import random
    from isocon_amd import _lib
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = random.Random(12)

    def family(alphabet, n, L):
        root = [rng.choice(alphabet) for _ in range(L)]
        out = []
        for _ in range(n):
            s = list(root)
            for _ in range(rng.randrange(0, 9)):
                i = rng.randrange(len(s))
                r = rng.random()
                if r < 0.4:
                    s[i] = rng.choice(alphabet)
                elif r < 0.7:
                    del s[i]
                else:
                    s.insert(i, rng.choice(alphabet))
            out.append("".join(s))
        return out

    for alphabet in ("acgt", "ACGU", "AB", "aCgT", "ACG"):
        seqs = sorted(set(family(alphabet, 40, 150) + family(alphabet, 30, 90)), key=len)
        st = SeqStore(seqs)
        try:
            a = [rng.randrange(len(seqs)) for _ in range(200)]
            b = [rng.randrange(len(seqs)) for _ in range(200)]
            got = st.ed_pairs(a, b, None)
            assert got.tolist() == [O.ed_dp(seqs[x], seqs[y]) for x, y in zip(a, b)], alphabet
            k = [rng.randrange(0, 12) for _ in range(200)]
            gk = st.ed_pairs(a, b, k)
            assert gk.tolist() == [d if d <= kk else -1 for d, kk in zip(got.tolist(), k)]
            if alphabet != "ACG":          # (a subset of ACGT is an ordinary store)
                with pytest.raises(_lib.IsoconError) as e:
                    st.sg_trace(a[:4], b[:4], -2)
                assert "ACGT" in str(e.value)
        finally:
            st.close()
        S = {"r%d" % i: s for i, s in enumerate(seqs)}
        g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
        g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
        assert ordered(g_gpu) == ordered(g_cpu), alphabet


@pytest.mark.gpu
def test_packing_in_several_launches(monkeypatch):
    """A launch holds fewer than 2^32 threads: k_pack_planes (one wavefront per 64 bases of a sequence) covers a large set in several launches.
    Forced here on a small set (ISOCON_DEBUG_VARIANT=pack_waves): same planes (fingerprint), same distances, also for a set with other symbols."""
    import random
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = random.Random(2)
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 700))) for _ in range(300)]
    mixed = [s if i % 7 else s[:len(s) // 2] + "Nn" + s[len(s) // 2:] for i, s in enumerate(seqs)]
    want = {}
    for name, ss in (("acgt", seqs), ("mixed", mixed)):
        st = SeqStore(ss)
        a = [rng.randrange(len(ss)) for _ in range(100)]
        b = [rng.randrange(len(ss)) for _ in range(100)]
        want[name] = (st.fingerprint, a, b, st.ed_pairs(a, b, None).tolist())
        assert want[name][3] == [O.ed_dp(ss[x], ss[y]) for x, y in zip(a, b)]
        st.close()
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "pack_waves=1000")
    for name, ss in (("acgt", seqs), ("mixed", mixed)):
        st = SeqStore(ss)
        fp, a, b, d = want[name]
        assert st.fingerprint == fp and st.ed_pairs(a, b, None).tolist() == d
        st.close()
