"""GPU: the alignment (SW / CIGAR) half of the metric at configuration scale against the oracle-made fixtures g18
(tests/golden/make_golden_g18.py): EVERY pair of C3's partition pair list (49 990 pairs, configs[2]) and of C2's partition pair list
and edge list (configs[1]) -- distances, penalty buckets, score, end cell, matches / mismatches / indels, the run-length ops (as a
64-bit hash per pair), the exon flag -- through the public functions (partition_strings, get_partition_alignments,
edlib_align_sequences, sw_align_sequences: isocon_get_candidates.py:37-81, EAM:10-49, SWM:89-164) and through the C ABI directly
(banded with distance hints AND the full-matrix kernel).  The 2 000 sampled C5 pairs are in test_gpu_c5_full.py."""
import numpy as np
import pytest

from conftest import Params, g17, g18, ops_of_alignment

pytestmark = pytest.mark.gpu
CONFIGS = {"c2": (5000, 1500, 3, 20001), "c3": (50000, 2500, 10, 30001)}


class P(Params):
    min_exon_diff = 20
    ignore_ends_len = 15


def _pipeline(which):
    from isocon_amd import partitions, synth
    accs, seqs_all, _ = synth.make_reads(*CONFIGS[which])
    S = dict(zip(accs, seqs_all))
    entries = sorted(dict.fromkeys(seqs_all), key=len)
    G_star, partition, M, converged = partitions.partition_strings(S, P())
    return S, entries, {s: i for i, s in enumerate(entries)}, G_star, partition, M


@pytest.fixture(scope="module")
def c3():
    return _pipeline("c3")


@pytest.fixture(scope="module")
def c2():
    return _pipeline("c2")


def _sorted_pairs(pairs, idx):
    a = np.fromiter((idx[m] for m, s in pairs), dtype=np.int64, count=len(pairs))
    b = np.fromiter((idx[s] for m, s in pairs), dtype=np.int64, count=len(pairs))
    o = np.lexsort((b, a))
    return a, b, o


def _check_partition(fx, entries, idx, partition, M):
    pairs = [(m, s) for m, members in partition.items() for s in members]
    a, b, o = _sorted_pairs(pairs, idx)
    assert len(pairs) == len(fx["part_a"])
    assert (a[o] == fx["part_a"]).all() and (b[o] == fx["part_b"]).all()          # bit-exact partitions at configuration scale
    assert sorted(idx[m] for m in partition) == fx["centres"].tolist()
    assert [M[entries[c]] for c in fx["centres"].tolist()] == fx["weights"].tolist()


def _check_partition_alignments(fx, S, entries, idx, G_star, partition, M, digest=None):
    import bench
    from isocon_amd import isocon_get_candidates as IGC
    exon_filtered = set()
    pa = IGC.get_partition_alignments(partition, M, G_star, exon_filtered, P())
    batch = pa.batch
    assert batch is not None, "the alignments did not stay CIGAR ops"
    a, b, o = _sorted_pairs(batch.pairs, idx)
    assert (a[o] == fx["part_a"]).all() and (b[o] == fx["part_b"]).all()
    res = np.asarray(batch.res)[o]
    bad = np.flatnonzero((res != fx["part_res"]).any(axis=1))
    assert len(bad) == 0, "score / end cell / counts differ from the oracle at %d pairs, first %s: %s vs %s" % (
        len(bad), bad[:5], res[bad[:1]], fx["part_res"][bad[:1]])
    h = bench.sw_pair_hashes(batch.ops, batch.ops_ptr)[o]
    assert (np.diff(batch.ops_ptr)[o] == fx["part_n_ops"]).all()
    assert (h == fx["part_ops_hash"]).all(), "CIGAR ops differ from the oracle at %d pairs" % int((h != fx["part_ops_hash"]).sum())
    if digest is not None:
        assert bench.sw_digest(a, b, batch.res, bench.sw_pair_hashes(batch.ops, batch.ops_ptr)) == digest == str(fx["digest"])
    # the exon filter and the values the correction reads (isocon_get_candidates.py:56-76)
    flagged = np.flatnonzero(fx["part_exon"])
    assert exon_filtered == set(entries[j] for j in fx["part_b"][flagged].tolist())
    kept = fx["part_exon"] == 0
    assert sum(len(v) - 1 for v in pa.values()) == int(kept.sum())
    edit = fx["part_res"][:, 4] + fx["part_res"][:, 5]
    for p in np.flatnonzero(kept)[:: max(1, int(kept.sum()) // 3000)].tolist():
        v = pa[entries[int(fx["part_a"][p])]][entries[int(fx["part_b"][p])]]
        assert v[0] == edit[p] and v[3] == 1
    return pa


def _check_wrappers(fx, prefix, entries, matches):
    """edlib_align_sequences + sw_align_sequences on {s1: iterable(s2)}: every distance, every alignment"""
    import bench
    from isocon_amd import SW_alignment_module as SWM
    from isocon_amd import edlib_alignment_module as EAM
    fa, fb = fx[prefix + "a"], fx[prefix + "b"]
    pos = {(int(x), int(y)): p for p, (x, y) in enumerate(zip(fa.tolist(), fb.tolist()))}
    idx = {s: i for i, s in enumerate(entries)}
    ed = EAM.edlib_align_sequences(matches)
    n = 0
    for s1, inner in ed.items():
        for s2, d in inner.items():
            assert d == fx[prefix + "ed"][pos[(idx[s1], idx[s2])]]
            n += 1
    assert n == len(fa)
    sw = SWM.sw_align_sequences(ed)
    ops_list, order, counts = [], [], []
    for s1, inner in sw.items():
        for s2, (a1, a2, cnt) in inner.items():
            ops_list.append(ops_of_alignment(a1, a2))
            order.append(pos[(idx[s1], idx[s2])])
            counts.append(cnt)
    assert sorted(order) == list(range(len(fa)))
    ptr = np.zeros(len(ops_list) + 1, np.int64)
    np.cumsum([len(o) for o in ops_list], out=ptr[1:])
    h = bench.sw_pair_hashes(np.concatenate(ops_list), ptr)
    order = np.asarray(order)
    assert (np.asarray(counts, dtype=np.int32) == fx[prefix + "res"][order, 3:6]).all()
    assert (h == fx[prefix + "ops_hash"][order]).all(), "gapped strings differ from the oracle's at %d pairs" % int((h != fx[prefix + "ops_hash"][order]).sum())


def _check_abi(fx, prefix, entries, full_sample=None):
    """isocon_ed_pairs + isocon_sg_trace_batch on the fixture's pairs: banded with distance hints on all of them; the full-matrix kernel
    (no hints) on a sample"""
    import bench
    from isocon_amd.store import SeqStore
    st = SeqStore(entries)
    try:
        a, b = fx[prefix + "a"], fx[prefix + "b"]
        ed = st.ed_pairs(a, b, None)
        assert (ed == fx[prefix + "ed"]).all()
        rate = ed.astype(np.float64) / np.minimum(st.lens[a], st.lens[b]).astype(np.float64)
        mm = np.where(rate <= 0.01, -1, np.where(rate <= 0.09, -2, -4)).astype(np.int8)
        assert (mm == fx[prefix + "mismatch"]).all()                                   # SWM:102-109's buckets
        ops, ptr, res = st.sg_trace(a, b, mm, ed_upper=ed)
        assert (res == fx[prefix + "res"]).all()
        assert (bench.sw_pair_hashes(ops, ptr) == fx[prefix + "ops_hash"]).all()
        pick = np.arange(len(a)) if full_sample is None else np.random.default_rng(5).choice(len(a), full_sample, replace=False)
        ops, ptr, res = st.sg_trace(a[pick], b[pick], mm[pick])
        assert (res == fx[prefix + "res"][pick]).all()
        assert (bench.sw_pair_hashes(ops, ptr) == fx[prefix + "ops_hash"][pick]).all()
    finally:
        st.close()


def test_c3_partition_equals_the_fixture(c3):
    S, entries, idx, G_star, partition, M = c3
    assert entries == g17("c3")[0]
    _check_partition(g18("c3"), entries, idx, partition, M)


def test_c3_get_partition_alignments_every_pair(c3):
    """all 49 990 pairs of configs[2]'s step-1 partition; and the digest bench.py asserts for its wrappers leg"""
    import bench
    S, entries, idx, G_star, partition, M = c3
    _check_partition_alignments(g18("c3"), S, entries, idx, G_star, partition, M, digest=bench.EXPECTED_SW_DIGEST_C3)


def test_c3_wrappers_every_pair(c3):
    S, entries, idx, G_star, partition, M = c3
    _check_wrappers(g18("c3"), "part_", entries, partition)


def test_c3_abi_every_pair_banded_and_a_full_matrix_sample(c3):
    _check_abi(g18("c3"), "part_", c3[1], full_sample=3000)


def test_c2_partition_and_alignments_every_pair(c2):
    S, entries, idx, G_star, partition, M = c2
    fx = g18("c2")
    assert entries == g17("c2")[0]
    _check_partition(fx, entries, idx, partition, M)
    _check_partition_alignments(fx, S, entries, idx, G_star, partition, M)
    _check_wrappers(fx, "part_", entries, partition)
    _check_abi(fx, "part_", entries)


def test_c2_every_edge_of_the_graph(c2):
    """configs[1]'s whole edge list (query, neighbour): 11 548 alignments"""
    S, entries, idx, G_star, partition, M = c2
    fx = g18("c2")
    matches = {}
    for x, y in zip(fx["edge_a"].tolist(), fx["edge_b"].tolist()):
        matches.setdefault(entries[x], []).append(entries[y])
    _check_wrappers(fx, "edge_", entries, matches)
    _check_abi(fx, "edge_", entries)
