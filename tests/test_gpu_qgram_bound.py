"""The q-gram pre-filter of the NN main pass (isocon_amd/csrc/qgram_mm.hpp): its bound equals a numpy restatement (pair list and every
byte of the matrix the main pass reads), never exceeds the oracle's edit distance, and the graph with the filter is the graph without it."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from qgram_ref import bound, profile


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


def _check_matrix(st, seqs, B, q_begin, q_end, q_stride, depth, q_block=1):
    """every byte of the matrix the main pass reads (isocon_qgram_bound_matrix) against the restatement"""
    lens = np.array([len(s) for s in seqs])
    n = len(seqs)
    from isocon_amd.store import shard_entries
    row_ptr, got = st.qgram_bound_matrix(q_begin, q_end, q_stride, depth, q_block)
    qs = [int(x) for x in shard_entries(q_begin, min(q_end, n), q_stride, q_block)]
    assert len(row_ptr) == len(qs) + 1
    total = 0
    for r, q in enumerate(qs):
        hi = int(np.searchsorted(lens, lens[q] + 63, "right"))
        hi = min(hi, q + 1 + depth)
        js = np.arange(q + 1, max(hi, q + 1))
        assert int(row_ptr[r + 1] - row_ptr[r]) == len(js), (q, len(js))
        if len(js):
            want = np.minimum(B[q, js], 255)
            have = got[int(row_ptr[r]):int(row_ptr[r + 1])]
            bad = np.nonzero(have != want)[0]
            assert len(bad) == 0, (q_begin, q_stride, depth, q, int(js[bad[0]]), int(have[bad[0]]), int(want[bad[0]]), len(bad))
        total += len(js)
    return total


def test_every_byte_of_the_bound_matrix():
    """The product kernel k_qgram_mm (tiles of 256 x 256 pairs on the matrix cores, rows laid out for the main pass) against the numpy
    restatement: 1-set, strided and block-cyclic shards, a last row block that is not full, finite depths, n not a multiple of the tile."""
    import random
    import qgram_ref as R
    from isocon_amd import synth
    from isocon_amd.store import SeqStore
    rng = random.Random(3)
    accs, seqs, _ = synth.make_reads(2900, 1000, 5, seed=41)
    seqs = list(dict.fromkeys(seqs))
    seqs += ["", "A", "ACGTACGTA", "ACGTACGTAC", "A" * 990, "AC" * 500, "ACG" * 340] + ["".join(rng.choice("ACGT") for _ in range(rng.randrange(900, 1100))) for _ in range(150)]
    seqs = sorted(seqs, key=len)
    assert len(seqs) % 256 not in (0, 255)
    B = R.bound_matrix(seqs)
    st = SeqStore(seqs)
    try:
        n = len(seqs)
        total = _check_matrix(st, seqs, B, 0, n, 1, 2 ** 32)
        assert total > 1000000
        for q_begin, q_stride in ((0, 3), (1, 3), (2, 3), (5, 7)):
            _check_matrix(st, seqs, B, q_begin, n, q_stride, 2 ** 32)
        # block-cyclic shards (isocon_amd/dist.py: blocks of one tile row dealt round-robin), a smaller block, a last block cut by q_end
        for q_begin, q_stride, q_block in ((0, 768, 256), (256, 768, 256), (512, 768, 256), (64, 256, 64), (8, 24, 8)):
            _check_matrix(st, seqs, B, q_begin, n, q_stride, 2 ** 32, q_block)
        _check_matrix(st, seqs, B, 256, n - 100, 512, 150, 256)
        _check_matrix(st, seqs, B, 0, n, 1, 300)
        _check_matrix(st, seqs, B, 1, n - 7, 2, 37)
        _check_matrix(st, seqs, B, n - 40, n, 1, 2 ** 32)
    finally:
        st.close()


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
        os.environ["ISOCON_DEBUG_VARIANT"] = "nn_no_qgram=1"
        try:
            best0, row_ptr0, cols0, stats0 = st.nn_graph(is_converged=conv)
        finally:
            del os.environ["ISOCON_DEBUG_VARIANT"]
        assert stats0["pairs_prefiltered"] == 0
        assert stats["pairs_evaluated"] < stats0["pairs_evaluated"]
        assert (best == best0).all() and (row_ptr == row_ptr0).all() and (cols == cols0).all()
        # the bounds with the 64-neighbour seed pass instead of the smallest-bound seeds, and the other workgroup shapes / orders
        # ... the survivor lists switched off (the kernel's own admission), every pair through a table / one pair per lane, and lists
        # that do not fit (fallback to the kernel's own admission)
        for env in ("nn_old_seed", "nn_waves=8", "nn_order=0", "nn_order=1", "nn_no_list", "nn_list_min=1", "nn_list_min=1000000", "nn_list_cap=1000",
                    "nn_list_waves=4", "nn_list_waves=8", "nn_host_finalize", "nn_narrow=1", "nn_narrow=0", "nn_narrow=1,nn_list_waves=4", "nn_narrow=0,nn_list_waves=4",
                    "nn_seed_classes=1", "nn_seed_classes=2"):
            # ... each of them behind the block filter (csrc/nn_filter.hpp: default, the survivors one per lane), with the filter and the table
            # launches on whatever it leaves, and without the filter (every survivor of the q-gram bound through the tables)
            for tables in ("", ",nn_filter_one_pass", ",nn_table_chunks=0", ",nn_no_block_filter"):
                os.environ["ISOCON_DEBUG_VARIANT"] = env + tables          # (the one switch behind which the A/B variants sit: DESIGN.md section 8)
                try:
                    b2, r2, c2, s2 = st.nn_graph(is_converged=conv)
                finally:
                    del os.environ["ISOCON_DEBUG_VARIANT"]
                assert s2["pairs_prefiltered"] > 0
                assert (s2["pairs_block_rejected"] > 0) == (tables != ",nn_no_block_filter" and env not in ("nn_no_list", "nn_list_cap=1000")), (env, tables)
                assert (best == b2).all() and (row_ptr == r2).all() and (cols == c2).all(), (env, tables)
        # the same with a finite depth and a strided shard (the row layout follows the launch slots)
        is_t = np.zeros(len(seqs), np.uint8); is_t[::5] = 1
        g1 = st.nn_graph(is_target=is_t)
        os.environ["ISOCON_DEBUG_VARIANT"] = "nn_no_qgram=1"
        try:
            g0 = st.nn_graph(is_target=is_t)
        finally:
            del os.environ["ISOCON_DEBUG_VARIANT"]
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
                hits, stats = st.nn_partial(r * 64, n, phase, b, depth=400, q_stride=192, q_block=64)
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
