"""GPU, BASELINE.json's full size (config C3: 50 k reads x ~2.5 kb, 10 isoforms): size-independent properties of the
exact NN graph plus oracle spot checks -- the CPU oracle needs ~0.2 s per query at this size, so only a sample of rows
is recomputed with the reference loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3():
    from isocon_amd import synth
    from isocon_amd.store import SeqStore
    accs, seqs, _ = synth.make_reads(50000, 2500, 10, seed=30001)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    best, row_ptr, cols, stats = st.nn_graph()
    return seqs, st, best, row_ptr, cols, stats


def test_c3_structure_and_symmetry(c3):
    seqs, st, best, row_ptr, cols, stats = c3
    n = len(seqs)
    lens = st.lens
    assert row_ptr[-1] == len(cols) and (np.diff(row_ptr) >= 0).all()
    has = np.diff(row_ptr) > 0
    assert has.all()                                   # dense isoform clusters: every read has a neighbour
    assert (best[has] > 0).all() and (best <= lens).all()
    rows = np.repeat(np.arange(n), np.diff(row_ptr))
    assert (cols != rows).all()
    assert (np.abs(lens[cols] - lens[rows]) <= best[rows]).all()       # a neighbour at distance d differs by <= d in length
    assert (best[cols] <= best[rows]).all()                            # symmetry of the metric: my NN's NN is at least as close
    # reference insertion order: ascending |offset|, lower index first
    off = np.abs(cols.astype(np.int64) - rows)
    same = rows[1:] == rows[:-1]
    assert ((off[1:] > off[:-1]) | ((off[1:] == off[:-1]) & (cols[1:] > cols[:-1])))[same].all()


def test_c3_idempotent(c3):
    seqs, st, best, row_ptr, cols, stats = c3
    b2, r2, c2, _ = st.nn_graph()
    assert (b2 == best).all() and (r2 == row_ptr).all() and (c2 == cols).all()


def test_c3_edges_are_true_distances(c3):
    """every reported edge re-aligned through the independent pair-list entry point (unbounded)"""
    seqs, st, best, row_ptr, cols, stats = c3
    rows = np.repeat(np.arange(len(seqs)), np.diff(row_ptr))
    ed = st.ed_pairs(rows[:20000], cols[:20000], None)
    assert (ed == best[rows[:20000]]).all()


def test_c3_sampled_rows_equal_reference_loop(c3):
    from oracle import oracle as O
    seqs, st, best, row_ptr, cols, stats = c3
    packed = O.pack(seqs)
    conv = np.zeros(len(seqs), np.uint8)
    rng = np.random.default_rng(3)
    for i in rng.choice(len(seqs), 24, replace=False).tolist():
        rp, c, e, _ = O.nn_1set(seqs, conv, i, 1, packed=packed)
        assert cols[row_ptr[i]:row_ptr[i + 1]].tolist() == c.tolist(), i
        assert (e == best[i]).all()


def test_c3_every_row_equals_the_reference_loop_fixture(c3):
    """ALL 50 000 rows -- neighbours, their order, the distance -- against tests/golden/g17_c3_graph.npz, whose rows were each
    recomputed with the oracle's statement of the reference loop (NNG:110-198) on the CPU; and the digest bench.py asserts."""
    import bench
    from conftest import g17
    seqs, st, best, row_ptr, cols, stats = c3
    fseqs, fbest, frow_ptr, fcols = g17("c3")
    assert fseqs == seqs
    assert (np.diff(frow_ptr) > 0).all() and (fbest > 0).all()
    assert (best == fbest).all()
    assert (row_ptr == frow_ptr).all()
    assert (cols == fcols).all()
    assert bench.graph_digest(best, row_ptr, cols) == bench.EXPECTED_GRAPH_DIGEST_C3


def test_c3_pruning_counters(c3):
    """The work the search does at C3, not only its result: the three counters below are deterministic (the seeds finish before the lists are
    built, hub scores and seed keys come out of the bound kernel's epilogue) and have been the same through every rewrite of that epilogue
    (profiles/r06j_bench.json).  A change of the seeds, of the hub scores that decide which end owns a pair, of either bound or of a
    threshold moves them without touching the graph -- this is where it shows.  A DELIBERATE change of a tuning constant (bins, gram
    length, probe stride) updates the numbers here."""
    seqs, st, best, row_ptr, cols, stats = c3
    assert int(stats["pairs_prefiltered"]) == 293119143          # window pairs rejected by the q-gram bound
    assert int(stats["pairs_block_rejected"]) == 13913023        # survivors rejected by the block bound (both passes)
    assert int(stats["pairs_evaluated"]) == 418136               # pairs aligned (seeds not counted)
    assert int(stats["pairs_lanes"]) == int(stats["pairs_evaluated"])          # no table launch at C3: what is left is aligned one pair per lane


def test_c3_alignments_roundtrip(c3):
    """SW on 256 (read, NN) pairs at full length: un-gapped alignment == input (correction_module.py:273-275),
    counts consistent, score identity."""
    from isocon_amd import SW_alignment_module as SWM
    seqs, st, best, row_ptr, cols, stats = c3
    q = np.arange(0, len(seqs), len(seqs) // 256)[:256]
    t = cols[row_ptr[q]]
    mm = np.full(len(q), -2, dtype=np.int8)
    ops, ptr, res = st.sg_trace(t, q, mm)
    # the same pairs with their distances as band hints (what sw_align_sequences hands down): the diagonal-band kernel
    ops_b, ptr_b, res_b = st.sg_trace(t, q, mm, ed_upper=best[q])
    assert (res_b == res).all() and (ptr_b == ptr).all() and (ops_b == ops).all()
    for p in range(len(q)):
        s1, s2 = seqs[int(t[p])], seqs[int(q[p])]
        a1, a2 = SWM._ops_to_alignment(ops[ptr[p]:ptr[p + 1]].tolist(), s1, s2)
        assert a1.replace("-", "") == s1 and a2.replace("-", "") == s2 and len(a1) == len(a2)
        m = sum(1 for x, y in zip(a1, a2) if x == y and x != "-")
        x = sum(1 for x, y in zip(a1, a2) if x != y and x != "-" and y != "-")
        assert (m, x, len(a1) - m - x) == tuple(res[p, 3:6])
        assert x + (len(a1) - m - x) >= best[q[p]]          # an alignment cannot beat the edit distance


def test_c3_partitions_cover_and_follow_the_graph(c3):
    """partition_strings at full size (config C3 asks for the partitions): every unique string in exactly one
    partition, weights add up, partitions are connected pieces of the nearest-neighbour graph, deterministic."""
    from isocon_amd import partitions

    class P(object):
        nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None

    seqs = c3[0]
    S = {"read_%d" % i: s for i, s in enumerate(seqs[::5])}          # 10 k reads keep the Python dict work short
    S.update({"dup_%d" % i: s for i, s in enumerate(seqs[::500])})  # some multiplicity-2 ("converged") strings
    G, partition, M, converged = partitions.partition_strings(S, P())
    assert not converged
    members = [m for c in partition for m in partition[c]] + list(partition)
    assert len(members) == len(set(members)) == len(set(S.values()))
    assert sum(M.values()) == len(S)
    for c in list(partition)[:50]:            # a partition is one connected piece of the nearest-neighbour graph
        comp = set(partition[c]) | {c}
        seen, stack = {c}, [c]
        while stack:
            v = stack.pop()
            for w in list(G.successors(v)) + list(G.predecessors(v)):
                if w in comp and w not in seen:
                    seen.add(w); stack.append(w)
        assert seen == comp
    G2, partition2, M2, _ = partitions.partition_strings(dict(reversed(list(S.items()))), P())
    assert partition2 == partition and M2 == M


def test_c3_infix_alignments_properties(c3):
    """isocon_hw_pairs at full read length (2.5 kb), size-independent properties: a read inside itself (distance 0, whole
    range); a planted exact infix (distance 0, the planted start unless an earlier occurrence exists -- none in random
    2.5 kb sequences, first end, no terminal insertions); a read against its nearest neighbour: the infix distance never
    exceeds the global distance, equals -1 only above k, and the location spans a stretch of the target whose length differs
    from the read's by at most the distance."""
    from isocon_amd.store import SeqStore
    seqs, st, best, row_ptr, cols, stats = c3
    n = len(seqs)
    rng = np.random.default_rng(4)
    q = rng.choice(n, 3000, replace=False)
    # 1. self
    r = st.hw_pairs(q, q, 10)
    assert (r[:, 0] == 0).all() and (r[:, 1] == 0).all() and (r[:, 2] == st.lens[q] - 1).all() and (r[:, 3:] == 0).all()
    # 2. nearest neighbour, k = 63: compare with the global distance of the graph
    t = cols[row_ptr[q]]
    r = st.hw_pairs(q, t, 63)
    ed = best[q]
    hit = r[:, 0] >= 0
    assert (r[hit, 0] <= ed[hit]).all() and hit[ed <= 63].all()
    span = r[hit, 2] - r[hit, 1] + 1
    assert (np.abs(span - st.lens[q][hit]) <= r[hit, 0]).all() and (r[hit, 1] >= 0).all() and (r[hit, 2] < st.lens[t][hit]).all()
    assert (r[~hit, 1:] == np.array([-1, -1, 0, 0])).all()
    # 3. planted infixes: prefix + read + suffix
    sub = q[:400]
    pre = rng.integers(0, 25, len(sub))
    suf = rng.integers(0, 25, len(sub))
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    planted = [alphabet[rng.integers(0, 4, int(a))].tobytes().decode() + seqs[i] + alphabet[rng.integers(0, 4, int(b))].tobytes().decode()
               for i, a, b in zip(sub.tolist(), pre.tolist(), suf.tolist())]
    st2 = SeqStore([seqs[i] for i in sub.tolist()] + planted)
    m = len(sub)
    r = st2.hw_pairs(np.arange(m), np.arange(m, 2 * m), 25)
    assert (r[:, 0] == 0).all() and (r[:, 1] == pre).all() and (r[:, 2] == pre + st.lens[sub] - 1).all() and (r[:, 3:] == 0).all()
    st2.close()
