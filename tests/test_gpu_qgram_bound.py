"""The q-gram pre-filter of the NN main pass (isocon_amd/csrc/qgram.hpp): its bound equals a numpy restatement, never exceeds the
oracle's edit distance, and the graph with the filter is the graph without it."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q = 8
BINS = 6144


def profile(s):
    """min(255, occurrences) of every 6-gram, indexed like the kernel: low code bits of the bases | high code bits << 6."""
    code = np.zeros(256, np.int64)
    code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
    c = code[np.frombuffer(s.encode(), np.uint8)]
    ng = len(c) - Q + 1
    if ng <= 0:
        return np.zeros(BINS, np.int64)
    idx = np.zeros(ng, np.int64)
    for i in range(Q):
        idx |= (c[i:i + ng] & 1) << i
        idx |= (c[i:i + ng] >> 1) << (Q + i)
    if BINS != 4 ** Q:
        idx = (((idx * 0x9E3779B1) & 0xffffffff) >> 7) % BINS
    return np.minimum(np.bincount(idx, minlength=BINS), 255)


def bound(pa, pb):
    return int((np.abs(pa - pb).sum() + abs(int(pa.sum()) - int(pb.sum())) + 2 * Q - 1) // (2 * Q))


def _mixed_set():
    import random
    from isocon_amd import synth
    rng = random.Random(11)
    accs, seqs, _ = synth.make_reads(300, 900, 4, seed=77)
    seqs = list(dict.fromkeys(seqs))
    seqs += ["", "A", "ACGTA", "ACGTAC", "ACGTACG", "A" * 700, "AC" * 400, "".join(rng.choice("ACGT") for _ in range(900)),
             "T" * 300 + "".join(rng.choice("ACGT") for _ in range(500))]
    return sorted(seqs, key=len)


def test_bound_equals_the_restatement_and_never_exceeds_the_distance():
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    seqs = _mixed_set()
    rng = np.random.default_rng(5)
    a = rng.integers(0, len(seqs), 3000).astype(np.uint32)
    b = rng.integers(0, len(seqs), 3000).astype(np.uint32)
    a[:len(seqs)] = np.arange(len(seqs)); b[:len(seqs)] = np.arange(len(seqs))[::-1]        # every sequence at least once, incl. the odd ones
    st = SeqStore(seqs)
    try:
        got = st.qgram_bound_pairs(a, b)
    finally:
        st.close()
    prof = [profile(s) for s in seqs]
    want = np.array([bound(prof[i], prof[j]) for i, j in zip(a, b)])
    assert (got == want).all()
    d = O.ed_pairs(seqs, a, b, None)
    assert (got <= d).all()
    same = a == b
    assert (got[same] == 0).all()
    assert (got > 0).sum() > 1000          # the bound is not vacuous


def test_graph_with_the_filter_is_the_graph_without_it_and_the_filter_is_used():
    from isocon_amd import synth
    from isocon_amd.store import SeqStore
    accs, seqs, _ = synth.make_reads(3000, 1500, 3, seed=20001)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    conv = np.zeros(len(seqs), np.uint8); conv[::17] = 1
    st = SeqStore(seqs)
    try:
        best, row_ptr, cols, stats = st.nn_graph(is_converged=conv)
        assert stats["pairs_prefiltered"] > 0 and stats["bound_kernel_ms"] > 0
        os.environ["ISOCON_NN_NO_QGRAM"] = "1"
        try:
            best0, row_ptr0, cols0, stats0 = st.nn_graph(is_converged=conv)
        finally:
            del os.environ["ISOCON_NN_NO_QGRAM"]
        assert stats0["pairs_prefiltered"] == 0
        assert stats["pairs_evaluated"] < stats0["pairs_evaluated"]
        assert (best == best0).all() and (row_ptr == row_ptr0).all() and (cols == cols0).all()
        # the bounds with the 64-neighbour seed pass instead of the smallest-bound seeds, and the other workgroup shapes / orders
        for env in ({"ISOCON_NN_OLD_SEED": "1"}, {"ISOCON_NN_WAVES": "8"}, {"ISOCON_NN_ORDER": "0"}, {"ISOCON_NN_ORDER": "1"}):
            os.environ.update(env)
            try:
                b2, r2, c2, s2 = st.nn_graph(is_converged=conv)
            finally:
                for k in env:
                    del os.environ[k]
            assert s2["pairs_prefiltered"] > 0
            assert (best == b2).all() and (row_ptr == r2).all() and (cols == c2).all(), env
        # the same with a finite depth and a strided shard (the row layout follows the launch slots)
        is_t = np.zeros(len(seqs), np.uint8); is_t[::5] = 1
        g1 = st.nn_graph(is_target=is_t)
        os.environ["ISOCON_NN_NO_QGRAM"] = "1"
        try:
            g0 = st.nn_graph(is_target=is_t)
        finally:
            del os.environ["ISOCON_NN_NO_QGRAM"]
        assert all((x == y).all() for x, y in zip(g1[:3], g0[:3]))
    finally:
        st.close()


def test_sharded_phases_with_the_filter():
    from isocon_amd import _lib, synth
    from isocon_amd.store import SeqStore, nn_finalize
    accs, seqs, _ = synth.make_reads(1500, 1200, 3, seed=9)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    try:
        ref = st.nn_graph(depth=400)
        n = st.n
        best = np.full(n, _lib.NN_INF, np.int32)
        hits_all = []
        filtered = 0
        reused = 0
        for phase in (0, 1, 2):
            parts = []
            for r in ((0, 1, 2) if phase != 1 else (2, 0, 1)):
                b = best.copy()
                hits, stats = st.nn_partial(r, n, phase, b, depth=400, q_stride=3)
                filtered += stats.get("pairs_prefiltered", 0)
                # the main phase that directly follows its own shard's seed phase finds the bound matrix still in place
                reused += phase == 1 and stats["pairs_prefiltered"] > 0 and stats["bound_kernel_ms"] == 0
                hits_all.append(hits); parts.append(b)
            best = np.minimum.reduce(parts)
        assert reused == 1
        hits = np.concatenate(hits_all)
        out = nn_finalize(n, best, hits)
        assert filtered > 0
        assert all((x == y).all() for x, y in zip(out[:3], ref[:3]))
    finally:
        st.close()
