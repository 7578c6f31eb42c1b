"""GPU: BASELINE.json configs[4] at its FULL size on one GPU -- 200 000 reads with the ONT error profile (6 %), lengths 1-5 kb, 50
isoforms in 5 gene families (seed 50001): step 1's exact nearest-neighbour graph through the public
compute_nearest_neighbor_graph (modules/nearest_neighbor_graph.py:237-296).  Every row satisfies the size-independent properties,
sampled rows -- the 5 kb end included -- equal the reference loop (oracle restatement of NNG:110-198, neighbour order included),
sampled edges are true distances; and find_candidate_transcripts (modules/isocon_get_candidates.py:85-312) on ALL 200 000 reads -- the
candidate phase of the config at full size, ~150 s on one GPU -- keeps its invariants (the un-gapped alignments are the inputs, the
partition covers the reads, most true isoforms are among the candidates)."""
import numpy as np
import pytest

from conftest import Params
from test_gpu_configs import _check_properties, _check_rows_against_the_reference_loop, _graph_arrays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c5_reads():
    from isocon_amd import synth
    return synth.make_reads(200000, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))


@pytest.fixture(scope="module")
def c5(c5_reads):
    from isocon_amd import nearest_neighbor_graph as NNG
    accs, seqs, iso = c5_reads
    S = dict(zip(accs, seqs))
    graph, isolated = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    return S, graph, isolated, dict(NNG.LAST_STATS)


def test_c5_full_size_graph(c5):
    S, graph, isolated, stats = c5
    seqs, accs, best, rows, cols = _graph_arrays(S, graph)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    assert len(seqs) > 199000 and not isolated
    assert lens.min() < 1100 and lens.max() > 4000 and np.median(best[best >= 0]) > 63 and stats["fallback_queries"] > 0
    assert (best >= 0).mean() > 0.999                                # practically every read has a neighbour
    _check_properties(seqs, best, rows, cols)
    rng = np.random.default_rng(21)
    sample = rng.choice(len(seqs), 6, replace=False).tolist() + np.argsort(lens)[-2:].tolist()       # ... and the 5 kb end
    _check_rows_against_the_reference_loop(seqs, accs, graph, sample)
    # sampled edges through the independent pair-list entry point, unbounded
    from isocon_amd.store import SeqStore
    st = SeqStore(seqs)
    try:
        pick = rng.choice(len(rows), 4000, replace=False)
        assert (st.ed_pairs(rows[pick], cols[pick], None) == best[rows[pick]]).all()
    finally:
        st.close()


@pytest.mark.timeout(1500)
def test_c5_candidates_at_full_size_keep_their_invariants(c5_reads, tmp_path):
    from isocon_amd import isocon_get_candidates as IGC

    class P(Params):
        def __init__(self, out):
            Params.__init__(self, 1)
            self.outfolder = str(out)
            self.ignore_ends_len = 15
            self.min_candidate_support = 2
            self.is_fastq = False
            self.ccs = None
            self.logfile = None
            self.min_exon_diff = 20

    accs, seqs, iso = c5_reads
    n = len(seqs)
    assert n == 200000
    fa = tmp_path / "reads.fa"
    with open(fa, "w") as f:
        for a, s in zip(accs[:n], seqs[:n]):
            f.write(">%s\n%s\n" % (a, s))
    (tmp_path / "out").mkdir()
    cand_file, read_partition, to_realign = IGC.find_candidate_transcripts(str(fa), P(tmp_path / "out"))
    assigned = 0
    reads = dict(zip(accs[:n], seqs[:n]))
    for c_acc, members in read_partition.items():
        for r_acc, (c_aln, r_aln, (m, x, ind)) in members.items():
            assert len(c_aln) == len(r_aln) and m + x + ind == len(c_aln)
            assert r_aln.replace("-", "") == reads[r_acc]                        # correction_module.py:273-275: the un-gapped alignment is the input
            assigned += 1
    assert assigned + len(to_realign) == n                                       # isocon_get_candidates.py:293
    assert len(read_partition) >= 10 and assigned > 0.9 * n
    from isocon_amd.input_output import fasta_parser
    cands = set(s for _, s in fasta_parser.read_fasta(open(cand_file)))
    assert sum(1 for t in set(iso) if t in cands) >= 25                         # of the 50 true isoforms (34 in profiles/r04k_c5_200k_get_candidates.log)


def test_c5_sampled_alignments_equal_the_oracle_fixture(c5_reads):
    """2 000 sampled (read, read of the same isoform) pairs of the 200 000-read set, the 5 kb end included (tests/golden/g18_c5_sw.npz: aligned
    on the CPU by the oracle, SWM:64-86 tie policy 0): distances, SWM:102-109's buckets (mostly -4 at 6 % errors), score, end cell, counts and
    ops of every pair -- banded with hints (bands beyond 256 diagonals: the strip kernel), the full-matrix kernel on a sample, and the public
    sw_align_sequences."""
    import hashlib
    import bench
    from conftest import g18, ops_of_alignment
    from isocon_amd import SW_alignment_module as SWM
    from isocon_amd.store import SeqStore
    fx = g18("c5")
    accs, seqs_all, iso = c5_reads
    entries = sorted(dict.fromkeys(seqs_all), key=len)
    fa, fb = fx["part_a"].astype(np.int64), fx["part_b"].astype(np.int64)
    sub = sorted(set(fa.tolist()) | set(fb.tolist()))
    pos = {v: k for k, v in enumerate(sub)}
    sseqs = [entries[v] for v in sub]
    h = hashlib.sha1()
    for s in sseqs:
        h.update(s.encode()); h.update(b"\n")
    assert h.hexdigest() == str(fx["inputs_sha1"])
    assert max(len(s) for s in sseqs) > 4000
    a = np.array([pos[v] for v in fa.tolist()], dtype=np.uint32)
    b = np.array([pos[v] for v in fb.tolist()], dtype=np.uint32)
    st = SeqStore(sseqs)
    try:
        ed = st.ed_pairs(a, b, None)
        assert (ed == fx["part_ed"]).all()
        rate = ed.astype(np.float64) / np.minimum(st.lens[a], st.lens[b]).astype(np.float64)
        mm = np.where(rate <= 0.01, -1, np.where(rate <= 0.09, -2, -4)).astype(np.int8)
        assert (mm == fx["part_mismatch"]).all() and (mm == -4).any()
        ops, ptr, res = st.sg_trace(a, b, mm, ed_upper=ed)
        assert (res == fx["part_res"]).all()
        assert (bench.sw_pair_hashes(ops, ptr) == fx["part_ops_hash"]).all()
        pick = np.concatenate([np.arange(0, len(a), 8), np.arange(len(a) - 10, len(a))])
        ops, ptr, res = st.sg_trace(a[pick], b[pick], mm[pick])
        assert (res == fx["part_res"][pick]).all()
        assert (bench.sw_pair_hashes(ops, ptr) == fx["part_ops_hash"][pick]).all()
    finally:
        st.close()
    matches = {}
    for p in range(0, len(a), 4):
        matches.setdefault(sseqs[int(a[p])], {})[sseqs[int(b[p])]] = int(fx["part_ed"][p])
    sw = SWM.sw_align_sequences(matches)
    where = {(int(a[p]), int(b[p])): p for p in range(len(a))}
    index = {s: i for i, s in enumerate(sseqs)}
    for s1, inner in sw.items():
        for s2, (a1, a2, cnt) in inner.items():
            p = where[(index[s1], index[s2])]
            o = ops_of_alignment(a1, a2)
            assert tuple(cnt) == tuple(fx["part_res"][p, 3:6]) and bench.sw_pair_hashes(o, [0, len(o)])[0] == fx["part_ops_hash"][p]

