"""Consensus correction against outputs of the reference's own correction_module / functions
(tests/golden/g11_correction.json): multi-alignment matrices (width + digest) and corrected sequences (digests)."""
import hashlib
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G7 = {c["name"]: c for c in json.load(open(os.path.join(HERE, "golden", "g7_partitions.json")))["cases"]}
G11 = json.load(open(os.path.join(HERE, "golden", "g11_correction.json")))["cases"]


class Params(object):
    nr_cores = 1
    neighbor_search_depth = 2 ** 32
    verbose = False
    develop_logfile = None
    min_exon_diff = 20
    ignore_ends_len = 15


def sha(s):
    return hashlib.sha1(s.encode()).hexdigest()[:16]


def run_chain(S):
    from isocon_amd import correction_module as COR
    from isocon_amd import functions as FUN
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import partitions
    G, partition, M, converged = partitions.partition_strings(S, Params())
    pa = IGC.get_partition_alignments(partition, M, G, set(), Params())
    seq_to_acc = IGC.get_unique_seq_accessions(S)
    S_prime, qual = COR.correct_strings(pa, seq_to_acc, {}, 1, nr_cores=1, verbose=False)
    assert qual == {}
    uid = {}
    for seq in S.values():
        uid.setdefault(seq, len(uid))
    msa = []
    for m in sorted(pa, key=lambda x: uid[x]):
        if len(pa[m]) > 1:
            am = FUN.create_multialignment_matrix(m, pa[m])
            msa.append([uid[m], len(am[m]), sha("".join("".join(am[s]) for s in sorted(am, key=lambda x: uid[x])))])
    return {"S_prime": sorted([acc, sha(s), len(s)] for acc, s in S_prime.items()), "msa": msa}


def test_insertion_placement_rules():
    from isocon_amd import functions as FUN
    assert FUN.get_best_solution("-GACG-", "-") == list("------")
    assert FUN.get_best_solution("-GACG-", "AC") == list("--AC--")            # substring: placed where it occurs
    assert FUN.get_best_solution("-GACG-", "AG") == list("--A-G-")            # threaded along an alignment without deletions
    assert FUN.min_ed("-GA-", "C") == "C---"
    vec, a, b = FUN.position_query_to_alignment("AC-GTT", "A-CG-T", 0)
    assert vec == ["-", "A", "C", "-", "-", "G", "T", "T", "-"] and (a, b) == (0, 8)


def test_pfm_counts_degrees():
    from isocon_amd import functions as FUN
    am = {"x": list("AC-"), "y": list("AG-")}
    pfm = FUN.create_position_frequency_matrix(am, {"x": (0, "", "", 3), "y": (1, "", "", 1)})
    assert pfm[0]["A"] == 4 and pfm[1] == {"A": 0, "C": 3, "G": 1, "T": 0, "-": 0} and pfm[2]["-"] == 4


@pytest.mark.parametrize("case", [c for c in G11 if c["name"] != "synth_300x600_4iso_dups"], ids=[c["name"] for c in G11 if c["name"] != "synth_300x600_4iso_dups"])
def test_correction_with_the_oracle_kernels(case, monkeypatch):
    from isocon_amd import graphs
    from isocon_amd import isocon_get_candidates as IGC
    from oracle import oracle as O
    from isocon_amd import correction_module as COR
    from oracle import correction as OC
    monkeypatch.setattr(COR, "_correct_on_device", OC.correct_rows)     # no GPU here: the numpy checker stands in for the kernels
    monkeypatch.setattr(graphs, "nearest_neighbor_graph", O)
    monkeypatch.setattr(IGC, "edlib_align_sequences", O.edlib_align_sequences)
    monkeypatch.setattr(IGC, "sw_align_sequences", O.sw_align_sequences)
    assert run_chain(dict(G7[case["name"]]["S"])) == case["expect"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", G11, ids=[c["name"] for c in G11])
def test_gpu_correction_chain(case):
    assert run_chain(dict(G7[case["name"]]["S"])) == case["expect"]


@pytest.mark.gpu
def test_device_correction_equals_host_statement(monkeypatch):
    """isocon_msa_correct vs the numpy statement of the same step on a partition with many reads and real indels."""
    import numpy as np
    from isocon_amd import correction_module as COR
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import partitions, synth
    accs, seqs, _ = synth.make_reads(1500, 900, 2, seed=61)
    S = dict(zip(accs, seqs))
    S.update({"dup%d" % i: seqs[i * 40] for i in range(10)})
    G, partition, M, converged = partitions.partition_strings(S, Params())
    pa = IGC.get_partition_alignments(partition, M, G, set(), Params())
    seq_to_acc = IGC.get_unique_seq_accessions(S)
    from oracle import correction as OC
    dev, _ = COR.correct_strings(pa, seq_to_acc, {}, 1)
    monkeypatch.setattr(COR, "_correct_on_device", OC.correct_rows)
    host, _ = COR.correct_strings(pa, seq_to_acc, {}, 1)
    assert dev == host and len(dev) > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("profile", ["ccs", "ont"])
def test_device_built_matrix_equals_the_host_matrix(profile):
    """isocon_msa_build_ops (the multi-alignment matrix from CIGAR ops and the packed store, csrc/msa_build.hpp) + the wide-slot patches
    against functions.msa_matrix on the gapped strings (the reference's layout, functions.py:543-588,679-767): every cell of every
    partition; CCS-like reads (single-base insertions) and reads with many multi-base insertions (wide slots)."""
    import numpy as np
    from isocon_amd import correction_module as COR
    from isocon_amd import functions as FUN
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import partitions, synth
    if profile == "ccs":
        accs, seqs, _ = synth.make_reads(900, 700, 3, seed=62)
    else:
        accs, seqs, _ = synth.make_reads(400, 500, 3, seed=63, profile=synth.ONT_PROFILE)
    S = dict(zip(accs, seqs))
    G, partition, M, converged = partitions.partition_strings(S, Params())
    pa = IGC.get_partition_alignments(partition, M, G, set(), Params())
    batch = pa.batch
    assert batch is not None and batch.alive()
    checked = wide_total = 0
    for m in pa:
        if len(pa[m]) < 2:
            continue
        rows = batch.rows_of[m]
        members = [batch.pairs[p][1] for p in rows]
        assert [m] + members == list(pa[m])
        row_ids, ops, ops_ptr = COR._partition_rows(batch, m)
        n_cols, col_slot, longest, wide = batch.store.msa_build_ops(row_ids, ops, ops_ptr)
        dev = batch.store.msa_read_built(len(row_ids), n_cols)
        p_row, p_col, p_ptr, p_bytes = COR._wide_slot_patches(members, wide, col_slot, longest)
        for i in range(0 if p_row is None else len(p_row)):
            dev[p_row[i], p_col[i]:p_col[i] + int(p_ptr[i + 1] - p_ptr[i])] = p_bytes[int(p_ptr[i]):int(p_ptr[i + 1])]
        keys, host = FUN.msa_matrix(m, pa[m])          # (expands the gapped strings of the lazy values)
        assert keys == [m] + members and host.shape == dev.shape
        assert (host == dev).all()
        checked += 1
        wide_total += len(wide)
    assert checked >= 3 and (profile == "ccs" or wide_total > 50)


@pytest.mark.gpu
def test_batched_correction_equals_one_partition_at_a_time():
    """correct_strings corrects ALL partitions of a step in one batched build + correct (isocon_msa_*_batch); the same partitions one at
    a time through isocon_msa_build_ops / _correct_built, and the string path with the numpy checker, give the same reads.  Many small
    partitions (40 isoforms, ONT-profile reads: wide slots) and a few large ones."""
    from isocon_amd import correction_module as COR
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import partitions, synth
    from oracle import correction as OC
    for args, kw in (((1200, 600, 40), dict(seed=71, profile=synth.ONT_PROFILE)), ((1500, 900, 3), dict(seed=72))):
        accs, seqs, _ = synth.make_reads(*args, **kw)
        S = dict(zip(accs, seqs))
        G, partition, M, converged = partitions.partition_strings(S, Params())
        pa = IGC.get_partition_alignments(partition, M, G, set(), Params())
        seq_to_acc = IGC.get_unique_seq_accessions(S)
        batched, _ = COR.correct_strings(pa, seq_to_acc, {}, 1)
        one_by_one = {}
        n_parts = 0
        for m in sorted(pa):
            if len(pa[m]) > 1 and sum(t[3] for t in pa[m].values()) > 2:
                one_by_one.update(COR._correct_partition_from_ops(pa.batch, m, pa[m], seq_to_acc))
                n_parts += 1
        assert batched == one_by_one and len(batched) > 500
        assert n_parts >= (20 if args[2] == 40 else 3)
        kernel = COR._correct_on_device
        COR._correct_on_device = OC.correct_rows
        try:
            host, _ = COR.correct_strings(pa, seq_to_acc, {}, 1)
        finally:
            COR._correct_on_device = kernel
        assert batched == host


@pytest.mark.gpu
def test_lazy_alignments_behave_like_the_tuples():
    """partition_alignments values of the ops path: indexing, iteration, equality with the tuples of the string path"""
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import partitions, synth
    accs, seqs, _ = synth.make_reads(300, 400, 2, seed=64)
    S = dict(zip(accs, seqs))
    G, partition, M, converged = partitions.partition_strings(S, Params())
    ex1, ex2 = set(), set()
    fast = IGC.get_partition_alignments(partition, M, G, ex1, Params())
    assert isinstance(fast, IGC.PartitionAlignments) and fast.batch is not None
    IGC._EDLIB_ALIGN = None          # (forces the string path)
    try:
        plain = IGC.get_partition_alignments(partition, M, G, ex2, Params())
    finally:
        IGC._EDLIB_ALIGN = IGC.edlib_align_sequences
    assert not hasattr(plain, "batch") or plain.batch is None
    assert ex1 == ex2 and list(fast) == list(plain)
    n = 0
    for m in plain:
        assert set(fast[m]) == set(plain[m])
        for s in plain[m]:
            t, u = plain[m][s], fast[m][s]
            assert u == t and tuple(u) == t and len(u) == 4 and u[0] == t[0] and u[1] == t[1] and u[2] == t[2] and u[3] == t[3]
            n += 1
    assert n > 250


@pytest.mark.gpu
def test_device_correction_rows_beyond_the_lds_list():
    """Rows with more than 2048 correctable positions take the second launch (list in HBM): same result as the checker."""
    import numpy as np
    from isocon_amd import correction_module as COR
    from oracle import correction as OC
    rng = np.random.default_rng(5)
    nr, ncols = 40, 9000
    base = rng.integers(0, 4, ncols)
    M = np.frombuffer(b"ACGT", dtype=np.uint8)[np.tile(base, (nr, 1))].copy()
    for r in (3, 17, 29):                           # three very noisy rows: ~1/3 of their positions differ from the majority
        pos = rng.choice(ncols, 3000 + 100 * r, replace=False)
        M[r, pos] = np.frombuffer(b"ACGT-", dtype=np.uint8)[rng.integers(0, 5, len(pos))]
    for r in range(nr):                             # and a little noise everywhere (varied frequencies, gaps)
        pos = rng.choice(ncols, 30, replace=False)
        M[r, pos] = np.frombuffer(b"ACGT-", dtype=np.uint8)[rng.integers(0, 5, 30)]
    deg = np.ones(nr, dtype=np.int64); deg[0] = 4
    p1, o1, n1 = COR._correct_on_device(M, deg)
    p2, o2, n2 = OC.correct_rows(M, deg)
    assert n1.max() > 2048 and (n1 == n2).all() and (o1 == o2).all() and (p1[:o1[-1]] == p2).all()
