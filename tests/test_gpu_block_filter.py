"""The block filter of the NN main pass (isocon_amd/csrc/nn_filter.hpp) against its numpy / Python restatement (tests/qgram_ref.py) and
against exact distances: the count is a lower bound of the edit distance (so a rejected pair is never a hit), and the graph with the
filter equals the graph without it and the oracle's statement of the reference loop (nearest_neighbor_graph.py:110-198)."""
import os

import numpy as np
import pytest

from tests import qgram_ref

pytestmark = pytest.mark.gpu


def _reads(n, length, iso, seed, profile=None):
    from isocon_amd import synth
    accs, seqs, _ = synth.make_reads(n, length, iso, seed, profile=profile)
    return sorted(dict.fromkeys(seqs), key=len)


def test_count_equals_the_restatement_and_bounds_the_distance():
    from isocon_amd.store import SeqStore
    seqs = _reads(400, 700, 3, 11)
    # edge shapes: shorter than one word, exactly at the word boundaries, a homopolymer, an exact copy, a prefix
    seqs += ["ACGTACGTAC", "A" * 19, "A" * 20, "C" * 36, "ACGT" * 9, seqs[5], seqs[7][:300], "G" * 700, seqs[9][::-1]]
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    rng = np.random.default_rng(3)
    a = rng.integers(0, len(seqs), 1500).astype(np.uint32)
    b = rng.integers(0, len(seqs), 1500).astype(np.uint32)
    a[:len(seqs)] = np.arange(len(seqs)); b[:len(seqs)] = np.arange(len(seqs))[::-1]      # every sequence as an owner and as a partner
    got = st.block_bound_pairs(a, b)
    want = np.array([qgram_ref.block_count(seqs[x], seqs[y]) for x, y in zip(a, b)])
    assert (got == want).all(), np.flatnonzero(got != want)[:10]
    got2 = st.block_bound_pairs(a, b, probe_stride=2)
    want2 = np.array([qgram_ref.block_count(seqs[x], seqs[y], s=2) for x, y in zip(a, b)])
    assert (got2 == want2).all(), np.flatnonzero(got2 != want2)[:10]
    d = st.ed_pairs(a, b, None)
    assert (got <= d).all() and (got2 <= d).all()
    assert got[a == b].max() == 0
    # near pairs: the count comes close to the distance (that is why it rejects what the q-gram bound lets through)
    near = d < 40
    assert near.sum() > 50 and (got[near] >= 0.6 * d[near] - 2).mean() > 0.9


def test_long_reads_and_the_ont_profile():
    from isocon_amd import synth
    from isocon_amd.store import SeqStore
    seqs = _reads(120, 5000, 2, 5, profile=synth.ONT_PROFILE)
    st = SeqStore(seqs)
    rng = np.random.default_rng(4)
    a = rng.integers(0, len(seqs), 300).astype(np.uint32)
    b = rng.integers(0, len(seqs), 300).astype(np.uint32)
    d = st.ed_pairs(a, b, None)
    for stride in (4, 2):
        got = st.block_bound_pairs(a, b, probe_stride=stride)
        want = np.array([qgram_ref.block_count(seqs[x], seqs[y], s=stride) for x, y in zip(a, b)])
        assert (got == want).all()
        assert (got <= d).all()


@pytest.mark.parametrize("n,length,iso,seed", [(3000, 900, 4, 21), (2500, 2500, 3, 22)])
def test_graph_with_and_without_the_filter(n, length, iso, seed, monkeypatch):
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    seqs = _reads(n, length, iso, seed)
    st = SeqStore(seqs)
    best, rp, cols, stats = st.nn_graph()
    assert stats["pairs_block_rejected"] > 0 and stats["filter_kernel_ms"] > 0
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "nn_filter_one_pass")          # without the second pass (a probe every 2 bases on what the first leaves)
    best1, rp1, cols1, stats1 = SeqStore(seqs).nn_graph()
    assert (best == best1).all() and (rp == rp1).all() and (cols == cols1).all()
    assert 0 < stats1["pairs_block_rejected"] <= stats["pairs_block_rejected"] and stats1["pairs_evaluated"] >= stats["pairs_evaluated"]
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "nn_no_block_filter")
    best0, rp0, cols0, stats0 = SeqStore(seqs).nn_graph()
    monkeypatch.delenv("ISOCON_DEBUG_VARIANT")
    assert stats0["pairs_block_rejected"] == 0
    assert (best == best0).all() and (rp == rp0).all() and (cols == cols0).all()
    # the filter removes work, nothing else: fewer pairs reach the alignment kernels
    assert stats["pairs_evaluated"] < stats0["pairs_evaluated"]
    # rows of the oracle's statement of the reference loop, sampled: same minimum, same neighbours in the same order
    rng = np.random.default_rng(1)
    packed = O.pack(seqs)
    conv = np.zeros(len(seqs), np.uint8)
    for q in rng.choice(len(seqs), 60, replace=False):
        rpq, c, e, _ = O.nn_1set(seqs, conv, int(q), 1, packed=packed)
        assert cols[rp[q]:rp[q + 1]].tolist() == c.tolist()
        assert (len(c) == 0 and best[q] < 0) or (e == best[q]).all()
