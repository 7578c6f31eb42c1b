"""GPU: BASELINE.json configs[1] at its FULL size (C2: 5 000 reads x ~1.5 kb, 3 isoforms, seed 20001) and a set of
configs[4]'s SHAPE (C5: ONT error profile, 6 %, lengths 1-5 kb, 50 isoforms in 5 gene families, seed 50001; 20 000 of its
200 000 reads) through the public compute_nearest_neighbor_graph (NNG:237-296): rows of sampled queries equal the
reference loop (oracle restatement of NNG:110-198, neighbour order included), every row satisfies the size-independent
properties, and the sliced entry points (NNG:110-198 / :341-424 as the reference's Pool calls them) return the rows of the
whole graph."""
import numpy as np
import pytest

from conftest import Params, golden, list_to_dd, ordered

pytestmark = pytest.mark.gpu


def _graph_arrays(S, graph):
    """public dict -> (sorted unique sequences, index of acc, best[], rows, cols)"""
    seq_to_acc = {seq: acc for acc, seq in S.items()}
    seqs = sorted(seq_to_acc, key=len)
    accs = [seq_to_acc[s] for s in seqs]
    pos = {a: i for i, a in enumerate(accs)}
    assert list(graph) == accs                                  # outer key order = length-sorted order
    best = np.full(len(seqs), -1, dtype=np.int64)
    rows, cols = [], []
    for a, nbrs in graph.items():
        i = pos[a]
        for b, d in nbrs.items():
            rows.append(i); cols.append(pos[b]); best[i] = d
        assert len(set(nbrs.values())) <= 1                     # an arg-min set: one distance per row
    return seqs, accs, best, np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)


def _check_properties(seqs, best, rows, cols):
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    assert (cols != rows).all()
    assert (best[rows] > 0).all() and (best[rows] <= lens[rows]).all()
    assert (np.abs(lens[cols] - lens[rows]) <= best[rows]).all()     # a neighbour at distance d differs by <= d in length
    has = best >= 0
    assert (best[cols][has[cols]] <= best[rows][has[cols]]).all()    # symmetry: my NN's NN is at least as close
    off = np.abs(cols - rows)
    same = rows[1:] == rows[:-1]
    assert ((off[1:] > off[:-1]) | ((off[1:] == off[:-1]) & (cols[1:] > cols[:-1])))[same].all()   # NNG:155-178 insertion order


def _check_rows_against_the_reference_loop(seqs, accs, graph, sample):
    from oracle import oracle as O
    packed = O.pack(seqs)
    conv = np.zeros(len(seqs), np.uint8)
    for i in sample:
        rp, c, e, _ = O.nn_1set(seqs, conv, int(i), 1, packed=packed)
        assert list(graph[accs[i]].items()) == [(accs[j], int(d)) for j, d in zip(c.tolist(), e.tolist())], i


@pytest.fixture(scope="module")
def c2():
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    accs, seqs, iso = synth.make_reads(5000, 1500, 3, seed=20001)
    S = dict(zip(accs, seqs))
    graph, isolated = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    return S, graph, isolated


def test_c2_full_size_rows_equal_reference_loop(c2):
    S, graph, isolated = c2
    seqs, accs, best, rows, cols = _graph_arrays(S, graph)
    assert not isolated and len(seqs) == len(set(S.values()))
    _check_properties(seqs, best, rows, cols)
    rng = np.random.default_rng(11)
    _check_rows_against_the_reference_loop(seqs, accs, graph, rng.choice(len(seqs), 240, replace=False).tolist())


def test_c2_full_size_distances_and_alignments_of_every_edge(c2):
    """EAM + SWM on C2's whole partition pair list: distances equal the graph's, alignments round-trip, counts add up."""
    from isocon_amd import SW_alignment_module as SWM
    from isocon_amd import edlib_alignment_module as EAM
    S, graph, isolated = c2
    matches = {}
    for a, nbrs in list(graph.items())[::3]:
        if nbrs:
            matches[S[a]] = set(S[b] for b in nbrs)
    ed = EAM.edlib_align_sequences(matches)
    inv = {seq: acc for acc, seq in S.items()}
    for s1, inner in ed.items():
        for s2, d in inner.items():
            assert d == graph[inv[s1]][inv[s2]]
    sw = SWM.sw_align_sequences(ed)
    n = 0
    for s1, inner in sw.items():
        for s2, (a1, a2, (m, x, ind)) in inner.items():
            assert a1.replace("-", "") == s1 and a2.replace("-", "") == s2 and len(a1) == len(a2)
            assert m + x + ind == len(a1) and x + ind >= ed[s1][s2]
            n += 1
    assert n == sum(len(v) for v in ed.values())


@pytest.fixture(scope="module")
def c5_shape():
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    accs, seqs, iso = synth.make_reads(20000, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
    S = dict(zip(accs, seqs))
    graph, isolated = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    return S, graph, isolated, dict(NNG.LAST_STATS)


def test_c5_shape_rows_equal_reference_loop(c5_shape):
    """1-5 kb reads at 6 % errors: nearest neighbours lie hundreds of edits away, so the 128/256/512-row lane-refill kernels
    (k_nn_scan_refill<16, 2|4|8>) at up to 5 kb and the un-banded tail (k_ed_full) produce these rows."""
    S, graph, isolated, stats = c5_shape
    seqs, accs, best, rows, cols = _graph_arrays(S, graph)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    assert lens.max() > 4000 and np.median(best[best >= 0]) > 63 and stats["fallback_queries"] > 0
    _check_properties(seqs, best, rows, cols)
    rng = np.random.default_rng(12)
    longest = np.argsort(lens)[-3:].tolist()                         # the 5 kb end explicitly
    sample = rng.choice(len(seqs), 13, replace=False).tolist() + longest
    _check_rows_against_the_reference_loop(seqs, accs, graph, sample)


def test_c5_shape_edges_are_true_distances(c5_shape):
    from isocon_amd.store import SeqStore
    S, graph, isolated, stats = c5_shape
    seqs, accs, best, rows, cols = _graph_arrays(S, graph)
    st = SeqStore(seqs)
    try:
        pick = np.random.default_rng(13).choice(len(rows), min(len(rows), 6000), replace=False)
        ed = st.ed_pairs(rows[pick], cols[pick], None)              # independent pair-list entry point, unbounded
        assert (ed == best[rows[pick]]).all()
    finally:
        st.close()


def test_sliced_entry_points_return_the_rows_of_the_whole_graph():
    """get_nearest_neighbors / get_nearest_neighbors_2set called chunk by chunk, as the reference's Pool does
    (NNG:33-65: chunks of max(n / (10 nr_cores), 20) queries), against the reference-generated fixtures."""
    from isocon_amd import nearest_neighbor_graph as NNG
    case = golden("g2_nn_graph_n200.json")
    S = dict(case["S"])
    seq_to_acc = {seq: acc for acc, seq in S.items()}
    lst = sorted(seq_to_acc.items(), key=lambda x: len(x[0]))
    merged = {}
    for start in range(0, len(lst), 20):
        batch = lst[start:start + 20]
        part = NNG.get_nearest_neighbors(batch, start, start, lst, set(), 2 ** 32)
        assert list(part) == [a for _, a in batch]
        merged.update(part)
    assert ordered(merged) == ordered(list_to_dd(case["graph"]))
    for case in golden("g2_nn_graph_2set.json")["cases"]:
        if case["depth"] < 2 ** 31:
            continue
        X, C = dict(case["X"]), dict(case["C"])
        lst = sorted([(s, a) for a, s in X.items()] + [(s, a) for a, s in C.items()], key=lambda x: len(x[0]))
        merged = {}
        for start in range(0, len(lst), 20):
            merged.update(NNG.get_nearest_neighbors_2set(lst[start:start + 20], start, lst, set(C), 2 ** 32))
        assert ordered(merged) == ordered(list_to_dd(case["graph"]))


def test_c2_every_row_equals_the_reference_loop_fixture(c2):
    """all 5 000 rows of configs[1] against tests/golden/g17_c2_graph.npz (every row recomputed with the oracle's reference loop)"""
    from conftest import g17
    S, graph, isolated = c2
    seqs, accs, best, rows, cols = _graph_arrays(S, graph)
    fseqs, fbest, frow_ptr, fcols = g17("c2")
    assert fseqs == seqs
    assert (best == fbest).all()
    counts = np.bincount(rows, minlength=len(seqs)) if len(rows) else np.zeros(len(seqs), np.int64)
    assert (np.concatenate([[0], np.cumsum(counts)]) == frow_ptr).all()
    assert (cols == fcols).all()          # (rows come out in key order, neighbours in the reference's insertion order)


def _check_2set_fixture(which):
    """compute_2set_nearest_neighbor_graph (NNG:201-234) on a configuration's reads against the seeded candidate set of g19: EVERY read's
    row -- candidates, their order, the distance -- equals the fixture the oracle's reference loop produced on the CPU; outer keys = the
    reads in merged-sorted order; reads without an admissible candidate map to {}."""
    from conftest import g19
    from isocon_amd import nearest_neighbor_graph as NNG
    X, C, merged, fx = g19(which)
    graph = NNG.compute_2set_nearest_neighbor_graph(X, C, Params(1))
    is_t, fbest, frp, fcols = fx["is_target"], fx["best"], fx["row_ptr"], fx["cols"]
    accs = [a for _, a in merged]
    assert list(graph) == [a for a, t in zip(accs, is_t.tolist()) if not t]
    pos = {a: i for i, a in enumerate(accs)}
    n_edges = 0
    bad = []
    for a, nbrs in graph.items():
        i = pos[a]
        exp_cols = fcols[frp[i]:frp[i + 1]].tolist()
        got_cols = [pos[b] for b in nbrs]
        ok = got_cols == exp_cols and all(d == int(fbest[i]) for d in nbrs.values()) and (len(nbrs) > 0 or int(fbest[i]) == -1)
        if not ok:
            bad.append(i)
        n_edges += len(nbrs)
    assert not bad, "%d rows differ from the reference loop, first %s" % (len(bad), bad[:5])
    assert n_edges == len(fcols)
    assert (fbest[is_t == 0] == 0).sum() >= 1          # (a read that IS a candidate: distance 0 is admitted, NNG:388)
    return len(graph), n_edges


def test_c2_2set_every_row_equals_the_reference_loop_fixture():
    rows, edges = _check_2set_fixture("c2")
    assert rows == 5000


def test_c3_2set_every_row_equals_the_reference_loop_fixture():
    """the search the metric is named after (read x candidate alignments) at the size it is quoted on: 50 000 reads x 1 030 candidates"""
    rows, edges = _check_2set_fixture("c3")
    assert rows == 50000
