"""Candidate-vs-candidate graph with ignored ends (edlib HW mode + path) against outputs of the
reference's own end_invariant_functions.get_NN_graph_ignored_ends_edlib / edlib_traceback (tests/golden/g14_all_nn.json)."""
import json
import os
import random

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G14 = json.load(open(os.path.join(HERE, "golden", "g14_all_nn.json")))


def params_of(case):
    class Params(object):
        ignore_ends_len = case["ignore_ends_len"]
        nr_cores = case["nr_cores"]
        neighbor_search_depth = case["neighbor_search_depth"]
        verbose = False
    return Params()


def as_list(g):
    return [[a, [list(x) for x in nb.items()]] for a, nb in g.items()]


class OracleStore(object):
    """Stands in for SeqStore where no GPU exists: hw_pairs from the oracle's full-matrix restatement."""

    def __init__(self, seqs):
        self.seqs = list(seqs)

    def hw_pairs(self, q, t, k, **_unused):
        from oracle import oracle as O
        out = np.full((len(q), 5), -1, dtype=np.int32)
        out[:, 3:] = 0
        for p, (a, b, kk) in enumerate(zip(q, t, k)):
            out[p] = hw_row(O, self.seqs[a], self.seqs[b], int(kk))
        return out


def hw_row(O, x, y, k):
    ed, start, end = O.hw_locate(x, y, k)
    if ed < 0:
        return [-1, -1, -1, 0, 0]
    _, ops = O.nw_path(x, y[start:end + 1])
    return [ed, start, end, ops[0][0] if ops[0][1] == "I" else 0, ops[-1][0] if ops[-1][1] == "I" else 0]


# ---- oracle vs the reference's outputs -----------------------------------------------------------------------------------
@pytest.mark.parametrize("case", G14["cases"], ids=[c["name"] for c in G14["cases"]])
def test_oracle_graph_equals_reference(case):
    from oracle import oracle as O
    assert as_list(O.get_NN_graph_ignored_ends_edlib(dict(case["C"]), params_of(case))) == case["expect"]


def test_oracle_traceback_equals_reference():
    from oracle import oracle as O
    for x, y, k, t, expect in G14["edlib_traceback"]:
        assert O.edlib_traceback_hw(x, y, k=k, end_threshold=t) == expect


def test_oracle_hw_distance_is_the_minimum_over_substrings():
    """Definition check of the oracle itself: HW distance = min over substrings of the global distance; the reported
    location attains it and no earlier end / shorter start does."""
    from oracle import oracle as O
    rng = random.Random(3)
    for _ in range(40):
        q = "".join(rng.choice("ACGT") for _ in range(rng.randint(3, 14)))
        t = "".join(rng.choice("ACGT") for _ in range(rng.randint(3, 20)))
        ed, start, end = O.hw_locate(q, t, -1)
        best = {(s, e): O.ed_dp(q, t[s:e + 1]) for s in range(len(t)) for e in range(s, len(t))}
        m = min(best.values())
        if m >= len(q):                 # an empty substring is as good: edlib's -1 end location, not used by the reference
            continue
        assert ed == m and best[(start, end)] == m
        assert min(e for (s, e), v in best.items() if v == m) == end
        assert min(s for (s, e), v in best.items() if v == m and e == end) == start


# ---- host logic of the product with the oracle standing in for the kernel -----------------------------------------------
@pytest.mark.parametrize("case", G14["cases"], ids=[c["name"] for c in G14["cases"]])
def test_host_logic_with_the_oracle_kernel(case, monkeypatch):
    from isocon_amd import end_invariant_functions as END
    monkeypatch.setattr(END, "SeqStore", OracleStore)
    assert as_list(END.get_NN_graph_ignored_ends_edlib(dict(case["C"]), params_of(case))) == case["expect"]


def test_unsorted_list_takes_the_loop(monkeypatch):
    from isocon_amd import end_invariant_functions as END
    from oracle import oracle as O
    monkeypatch.setattr(END, "SeqStore", OracleStore)
    C = dict(G14["cases"][0]["C"])
    lst = [(s, a) for a, s in C.items()]            # fixture order, not sorted by length
    assert END.get_all_NN(lst, 0, 0, lst, 2 ** 32, 15) == O.get_all_NN(lst, 0, 0, lst, 2 ** 32, 15)
    assert END.get_all_NN(lst[3:9], 3, 3, lst, 2, 5) == O.get_all_NN(lst[3:9], 3, 3, lst, 2, 5)


# ---- the HIP kernel ------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", G14["cases"], ids=[c["name"] for c in G14["cases"]])
def test_gpu_graph_equals_reference(case):
    from isocon_amd import end_invariant_functions as END
    assert as_list(END.get_NN_graph_ignored_ends_edlib(dict(case["C"]), params_of(case))) == case["expect"]


@pytest.mark.gpu
def test_gpu_traceback_equals_reference():
    from isocon_amd import end_invariant_functions as END
    for x, y, k, t, expect in G14["edlib_traceback"]:
        assert END.edlib_traceback(x, y, mode="HW", task="path", k=k, end_threshold=t) == expect


def _mut(rng, b, nmut, ends):
    v = list(b)
    for _ in range(nmut):
        p = rng.randrange(len(v))
        r = rng.random()
        if r < 0.4:
            v[p] = rng.choice("ACGT")
        elif r < 0.7:
            del v[p]
        else:
            v.insert(p, rng.choice("ACGT"))
    v = "".join(v)
    a, b2 = rng.randint(0, ends), rng.randint(0, ends)
    v = v[a:len(v) - b2]
    if rng.random() < 0.4:
        v = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, ends + 1))) + v
    if rng.random() < 0.4:
        v += "".join(rng.choice("ACGT") for _ in range(rng.randint(1, ends + 1)))
    return v


@pytest.mark.gpu
@pytest.mark.parametrize("L,k,ends,npairs", [(120, 25, 20, 300), (700, 25, 20, 120), (2500, 25, 20, 40), (300, 40, 45, 100), (400, 70, 60, 60),
                                             (600, 120, 50, 40), (200, 0, 0, 60), (64, 5, 3, 200)],
                         ids=["short", "mid", "ccs", "two_blocks", "four_blocks", "eight_blocks", "k0", "tiny"])
def test_gpu_hw_pairs_equal_oracle(L, k, ends, npairs):
    """All five outputs (distance, start, end, leading / trailing insertion run) for random related and unrelated pairs,
    one, two and four 64-diagonal blocks."""
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = random.Random(L * 1000 + k)
    seqs, q, t = [], [], []
    for p in range(npairs):
        b = "".join(rng.choice("ACGT") for _ in range(rng.randint(L - L // 8, L + L // 8)))
        x = _mut(rng, b, rng.choice([0, 1, 3, 8, 20]), ends)
        y = _mut(rng, b, rng.choice([0, 1, 3, 8]), ends) if rng.random() < 0.85 else "".join(rng.choice("ACGT") for _ in range(len(x) + rng.randint(-5, 5)))
        if k == 0 and p % 2 == 0:
            x, y = b[rng.randint(0, 9):len(b) - rng.randint(0, 9)], b          # an exact infix
        seqs += [x, y]
        q.append(2 * p); t.append(2 * p + 1)
    got = SeqStore(seqs).hw_pairs(q, t, k)
    hits = 0
    for p in range(npairs):
        exp = hw_row(O, seqs[2 * p], seqs[2 * p + 1], k)
        assert list(got[p]) == exp, (p, seqs[2 * p], seqs[2 * p + 1])
        hits += exp[0] >= 0
    assert hits > npairs // 10


@pytest.mark.gpu
def test_gpu_hw_pairs_limits():
    from isocon_amd.store import SeqStore
    st = SeqStore(["ACGTACGTAA", "ACGTACGTAAGG" * 30])
    with pytest.raises(RuntimeError):
        st.hw_pairs([0], [1], [-1])                 # k is required
    with pytest.raises(RuntimeError):
        st.hw_pairs([0], [1], [200])                # 350 + 400 + 1 diagonals > 512
    assert list(st.hw_pairs([1], [0], [5])[0]) == [-1, -1, -1, 0, 0]        # query longer than target + k


@pytest.mark.gpu
def test_gpu_hw_pairs_tiles_built_on_the_device(monkeypatch):
    """Calls of >= 2048 pairs build their tiles on the device (csrc/hw_tiles.hpp): shuffled pair order, thresholds of every class in one
    call, pairs that need no kernel; against the host-built tiles for every pair and against the oracle for a sample; the same error
    behaviour; and the call that only the host's greedy cut can tile (one query, two targets whose common window exceeds 512 diagonals)."""
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = random.Random(4242)
    seqs = []
    for fam in range(6):
        b = "".join(rng.choice("ACGT") for _ in range(rng.randint(250, 700)))
        for _ in range(40):
            seqs.append(_mut(rng, b, rng.choice([0, 1, 3, 8, 20]), 20))
    seqs.append("")                                                 # an empty sequence: no kernel, no hit
    n = len(seqs)
    q = np.array([rng.randrange(n) for _ in range(9000)], dtype=np.uint32)
    t = np.array([(int(x) // 40 * 40 + rng.randrange(40)) % n if rng.random() < 0.9 else rng.randrange(n) for x in q], dtype=np.uint32)
    k = np.array([rng.choice([0, 5, 25, 25, 25, 40, 70, 120]) for _ in q], dtype=np.int32)
    lens = np.array([len(s) for s in seqs])
    k = np.where(np.maximum(lens[t] - lens[q], 0) + 2 * k + 1 > 512, 25, k).astype(np.int32)      # (every pair supported on its own)
    st = SeqStore(seqs)
    try:
        dev = st.hw_pairs(q, t, k)
        monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "hw_host_tiles=1")
        host = st.hw_pairs(q, t, k)
        monkeypatch.delenv("ISOCON_DEBUG_VARIANT")
        assert (dev == host).all()
        assert (dev[:, 0] >= 0).sum() > 1000 and (dev[:, 0] < 0).sum() > 1000
        for p in range(0, len(q), 45):
            assert list(dev[p]) == hw_row(O, seqs[q[p]], seqs[t[p]], int(k[p])) if seqs[q[p]] and seqs[t[p]] else list(dev[p]) == [-1, -1, -1, 0, 0]
        kbad = k.copy(); kbad[4000] = -1
        with pytest.raises(RuntimeError):
            st.hw_pairs(q, t, kbad)
        qbad = q.copy(); qbad[17] = n
        with pytest.raises(RuntimeError):
            st.hw_pairs(qbad, t, k)
    finally:
        st.close()
    # one query; targets that fit on their own (longer target with a small k; same length with a large k) but not in one window
    base = "".join(rng.choice("ACGT") for _ in range(300))
    seqs2 = [base, base + "".join(rng.choice("ACGT") for _ in range(380)), base[:140] + "T" + base[141:]]
    st = SeqStore(seqs2)
    try:
        q2 = np.zeros(2048, dtype=np.uint32)
        t2 = np.array([1, 2] * 1024, dtype=np.uint32)
        k2 = np.array([20, 200] * 1024, dtype=np.int32)            # 380 + 41 and 0 + 401 diagonals; together 380 + 401
        r = st.hw_pairs(q2, t2, k2)
        assert list(r[0]) == hw_row(O, seqs2[0], seqs2[1], 20) and list(r[1]) == hw_row(O, seqs2[0], seqs2[2], 200)
        assert (r[0::2] == r[0]).all() and (r[1::2] == r[1]).all()
    finally:
        st.close()
