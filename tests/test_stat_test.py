"""The whole pipeline (IsoCon:145-177: candidate inference, then the statistical filter) against what the reference's own
find_candidate_transcripts + stat_filter_candidates write (tests/golden/g15_stat_test.json): final candidates with
support / p-value / partition size / variants, the read -> candidate table, the p-value table of every round.
Floats are compared with a relative tolerance of 1e-9 (the reference's own last digits depend on PYTHONHASHSEED,
tests/golden/make_golden_stat_test.py::same_up_to_float_digits)."""
import glob
import hashlib
import json
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_G15 = json.load(open(os.path.join(HERE, "golden", "g15_stat_test.json")))
_G12 = json.load(open(os.path.join(HERE, "golden", "g12_candidates.json")))


def _input_of(name):
    """the reads of a case: stored in g15, shared with g12, or (the larger public test sets) a gzipped FASTA next to them"""
    if name in _G15["inputs"]:
        return _G15["inputs"][name]
    if name in _G12["inputs"]:
        return _G12["inputs"][name]
    import gzip
    lines = gzip.open(os.path.join(HERE, "golden", "inputs_%s.fa.gz" % name), "rt").read().split("\n")
    return [[lines[i][1:], lines[i + 1]] for i in range(0, len(lines) - 1, 2)]


G15 = [dict(c, S=_input_of(c["input"])) for c in _G15["cases"]]


def sha(s):
    return hashlib.sha1(s.encode()).hexdigest()[:16]


_FLOAT = re.compile(r"\d+\.\d+(?:e-?\d+)?")


def same_up_to_float_digits(a, b, rel=1e-9):
    """Structural equality; decimal numbers inside strings (p-values in accessions / tsv cells) may differ by `rel` relative:
    the reference's own last digits move with PYTHONHASHSEED (it sums per-read terms in set order)."""
    if isinstance(a, list) and isinstance(b, list):
        return len(a) == len(b) and all(same_up_to_float_digits(x, y, rel) for x, y in zip(a, b))
    if isinstance(a, dict) and isinstance(b, dict):
        return list(a) == list(b) and all(same_up_to_float_digits(a[k], b[k], rel) for k in a)
    if isinstance(a, str) and isinstance(b, str):
        if _FLOAT.sub("#", a) != _FLOAT.sub("#", b):
            return False
        return all(abs(float(x) - float(y)) <= rel * max(abs(float(x)), abs(float(y))) for x, y in zip(_FLOAT.findall(a), _FLOAT.findall(b)))
    return a == b


def run(case, tmp_path):
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import isocon_statistical_test as IST
    tmp = str(tmp_path)
    fastq = case["name"].endswith("_fastq")          # records [acc, seq, qualities]: the tests then use the base qualities
    read_file = os.path.join(tmp, "reads.fq" if fastq else "reads.fa")
    with open(read_file, "w") as fh:
        if fastq:
            fh.write("".join("@%s\n%s\n+\n%s\n" % (a, s, q) for a, s, q in case["S"]))
        else:
            fh.write("".join(">%s\n%s\n" % (a, s) for a, s in case["S"]))

    class Params(object):
        nr_cores = 1
        neighbor_search_depth = 2 ** 32
        verbose = False
        develop_logfile = None
        logfile = None
        min_exon_diff = 20
        ignore_ends_len = 15
        min_candidate_support = 2
        p_value_threshold = 0.01
        min_test_ratio = 5
        max_phred_q_trusted = 43
        is_fastq = fastq
        ccs = None
        outfolder = tmp

    cand_file, read_partition, to_realign = IGC.find_candidate_transcripts(read_file, Params())
    C = IST.stat_filter_candidates(read_file, cand_file, read_partition, to_realign, Params())
    finals, acc = [], None
    for line in open(os.path.join(tmp, "final_candidates.fa")):
        if line.startswith(">"):
            acc = line[1:].strip()
        else:
            finals.append([acc, sha(line.strip()), len(line.strip())])
    assert sorted(len(s) for s in C.values()) == sorted(f[2] for f in finals)
    info = [l.rstrip("\n").split("\t") for l in open(os.path.join(tmp, "cluster_info.tsv"))]
    pv = {os.path.basename(f): [l.rstrip("\n").split("\t") for l in open(f)] for f in sorted(glob.glob(os.path.join(tmp, "p_values_*.tsv")))}
    return {"final_candidates": finals, "cluster_info": sorted(info), "p_values": pv}


class OracleStore(object):
    def __init__(self, seqs):
        self.seqs = list(seqs)

    def hw_pairs(self, q, t, k, **_unused):
        import numpy as np
        from oracle import oracle as O
        from test_all_nn import hw_row
        return np.asarray([hw_row(O, self.seqs[a], self.seqs[b], int(kk)) for a, b, kk in zip(q, t, k)], dtype=np.int32).reshape(-1, 5)


def oracle_align_pairs(pairs, mismatch, match_score=2, opening_penalty=2, gap_ext=0, ed_upper=None):
    from oracle import oracle as O
    return [O.parasail_alignment(a, b, 0, 0, match_score=match_score, mismatch_penalty=int(mm), opening_penalty=opening_penalty, gap_ext=gap_ext)[2]
            for (a, b), mm in zip(pairs, mismatch)]


@pytest.mark.parametrize("case", [c for c in G15 if c["name"].startswith("synth")], ids=[c["name"] for c in G15 if c["name"].startswith("synth")])
def test_pipeline_with_the_oracle_kernels(case, tmp_path, monkeypatch):
    import sys
    sys.path.insert(0, HERE)
    import isocon_amd.SW_alignment_module as SWM
    import isocon_amd.edlib_alignment_module as EAM
    from isocon_amd import correction_module as COR
    from isocon_amd import end_invariant_functions as END
    from isocon_amd import graphs
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import isocon_statistical_test as IST
    from oracle import correction as OC
    from oracle import oracle as O
    monkeypatch.setattr(COR, "_correct_on_device", OC.correct_rows)
    monkeypatch.setattr(graphs, "nearest_neighbor_graph", O)
    monkeypatch.setattr(IGC, "edlib_align_sequences", O.edlib_align_sequences)
    monkeypatch.setattr(IGC, "sw_align_sequences", O.sw_align_sequences)
    monkeypatch.setattr(EAM, "edlib_align_sequences_keeping_accession", O.edlib_align_sequences_keeping_accession)
    monkeypatch.setattr(SWM, "sw_align_sequences_keeping_accession", O.sw_align_sequences_keeping_accession)
    monkeypatch.setattr(IST, "edlib_align_sequences_keeping_accession", O.edlib_align_sequences_keeping_accession)
    monkeypatch.setattr(IST, "sw_align_sequences_keeping_accession", O.sw_align_sequences_keeping_accession)
    monkeypatch.setattr(SWM, "_align_pairs", oracle_align_pairs)
    monkeypatch.setattr(END, "SeqStore", OracleStore)
    assert same_up_to_float_digits(run(case, tmp_path), case["expect"])


def test_raghavan_bound_known_values():
    """Closed-form checks of the bound: no supporter -> 1; y == m -> 0.5; otherwise e^k / (1 + d)^(k + k / d)."""
    import math
    from isocon_amd import hypothesis_test_module as H
    p = {"a": 0.01, "b": 0.01, "c": 0.01, "d": 0.01}
    assert H.raghavan_upper_pvalue_bound(p, []) == 1.0
    m, y = 0.04, 2.0
    d = y / m - 1
    k = m * d
    assert abs(H.raghavan_upper_pvalue_bound(p, ["a", "b"]) / (math.exp(k) / (1 + d) ** (k + k / d)) - 1) < 1e-12
    assert H.get_correction_factor("ACGTACGTAC", "c", {3: ("S", "A", 1), 5: ("D", "-", 2), 7: ("I", "G", 1)}) == (4 * 11) * 10 * (3 * 9)


@pytest.mark.gpu
@pytest.mark.parametrize("case", G15, ids=[c["name"] for c in G15])
def test_gpu_pipeline(case, tmp_path):
    assert same_up_to_float_digits(run(case, tmp_path), case["expect"])


def test_read_tables_equal_the_per_read_functions():
    """The vectorised per-candidate read tables (hypothesis_test_module._ReadTable / _test_on_tables) against the per-read
    statements in isocon_amd.functions (get_support, get_read_errors, get_empirical_error_probabilities) and
    raghavan_upper_pvalue_bound: identical variants, supporter counts, reads used and bit-equal p-values."""
    import random
    from isocon_amd import hypothesis_test_module as H
    from oracle import oracle as O
    rng = random.Random(11)

    def mut(b, n, homopolymer=0.5):
        v = list(b)
        for _ in range(n):
            p = rng.randrange(len(v))
            r = rng.random()
            if r < 0.4:
                v[p] = rng.choice("ACGT")
            elif r < 0.7:
                del v[p]
            else:
                v.insert(p, v[p] if rng.random() < homopolymer else rng.choice("ACGT"))
        return "".join(v)

    def aln(a, b, **kw):
        return O.parasail_alignment(a, b, 0, 0, **kw)[2]

    tested = 0
    for trial in range(120):
        t = "".join(rng.choice("AACGTT") for _ in range(rng.randint(40, 160)))
        c = mut(t, rng.randint(0, 4))
        if rng.random() < 0.3:
            c = c[rng.randint(0, 6):]
        if rng.random() < 0.3:
            c = c + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 6)))
        if c == t:
            continue
        reads_c = {"c%d" % k: aln(c, mut(c, rng.randint(0, 4))[rng.randint(0, 3):]) for k in range(rng.randint(0, 7))}
        reads_t = {"t%d" % k: aln(t, mut(t if rng.random() < 0.6 else c, rng.randint(0, 4))) for k in range(rng.randint(0, 9))}
        tc = aln(t, c, opening_penalty=3, mismatch_penalty=-3, gap_ext=1)
        ct = aln(c, t, opening_penalty=3, mismatch_penalty=-3, gap_ext=1)
        try:
            slow = H._test_on_alignments(t, c, tc, ct, reads_c, reads_t)
        except IndexError:
            with pytest.raises(IndexError):
                H._test_on_tables(t, c, tc, ct, H._ReadTable(len(c), reads_c), H._ReadTable(len(t), reads_t))
            continue
        fast = H._test_on_tables(t, c, tc, ct, H._ReadTable(len(c), reads_c), H._ReadTable(len(t), reads_t))
        assert list(slow[0].items()) == list(fast[0].items())
        assert slow[1] == fast[1] and len(slow[2]) == fast[2] and slow[3] == fast[3], (trial, slow[1], fast[1])
        tested += slow[1] not in (0.0, 1.0)
    assert tested > 20


_G16 = json.load(open(os.path.join(HERE, "golden", "g16_stat_helpers.json")))


@pytest.mark.parametrize("ci", range(len(_G16["cases"])))
def test_stat_helpers_equal_reference(ci):
    """The helper functions of the test, called directly, against the reference's own (fixture g16): variant coordinates,
    supporting reads, per-read errors, empirical and quality-based error probabilities (floats bit-equal)."""
    from isocon_amd import ccs_info as CI
    from isocon_amd import functions as F
    g = _G16["cases"][ci]
    variants = [tuple(v) for v in g["variants"]]
    vt, vc, ac2t, at2c = F.get_variant_coordinates(g["t"], g["c"], g["aln_t"], g["aln_c"], variants)
    assert [[k, list(v)] for k, v in vt.items()] == g["variant_coords_t"] and [[k, list(v)] for k, v in vc.items()] == g["variant_coords_c"]
    assert [[k, v] for k, v in ac2t.items()] == g["alignment_c_to_t"] and [[k, v] for k, v in at2c.items()] == g["alignment_t_to_c"]
    rc = {a: (v[0], v[1], tuple(v[2])) for a, v in g["reads_c"].items()}
    rt = {a: (v[0], v[1], tuple(v[2])) for a, v in g["reads_t"].items()}
    if g["support"] == "IndexError":
        with pytest.raises(IndexError):
            F.get_support(rc, vc, rt, vt, ac2t)
    else:
        assert F.get_support(rc, vc, rt, vt, ac2t) == g["support"]
    errors = F.get_read_errors(rc, rt)
    assert [[a, list(e)] for a, e in errors.items()] == g["errors"]
    assert [[a, repr(p)] for a, p in F.get_empirical_error_probabilities(len(g["t"]), errors, vt).items()] == g["empirical"]
    reads = {a: v[1].replace("-", "") for a, v in list(rc.items()) + list(rt.items())}
    ccs = {a: CI.CCS(a, reads[a], g["qual"][a], "NA") for a in reads}
    for key, fn, ra, v, sn in (("ccs_c", F.get_read_ccs_probabilities_c, rc, vc, at2c), ("ccs_t", F.get_read_ccs_probabilities_t, rt, vt, ac2t)):
        if isinstance(g[key], str):
            with pytest.raises((AssertionError, IndexError)):
                fn(ra, v, sn, ccs, errors, 43)
        else:
            pr, non = fn(ra, v, sn, ccs, errors, 43)
            assert [[a, repr(p)] for a, p in pr.items()] == g[key]["prob"] and sorted(non) == g[key]["non_informative"]


def test_fix_quality_values_equal_reference():
    from isocon_amd import ccs_info as CI
    for s, q, expect in _G16["fix_quality_values"]:
        assert CI.fix_quality_values(s, q) == expect


def test_empty_candidate_file_writes_empty_outputs_and_exits(tmp_path):
    """isocon_statistical_test.py:166-171: no candidate -> empty final_candidates.fa / cluster_info.tsv, then sys.exit(0)
    (reached before any kernel is needed)."""
    from isocon_amd import isocon_statistical_test as IST
    reads = tmp_path / "reads.fa"
    reads.write_text(">r1\nACGTACGT\n>r2\nACGTTCGT\n")
    cands = tmp_path / "candidates_converged.fa"
    cands.write_text("")

    class Params(object):
        is_fastq = False
        ccs = None
        outfolder = str(tmp_path)

    with pytest.raises(SystemExit) as e:
        IST.stat_filter_candidates(str(reads), str(cands), {}, {"r1": "ACGTACGT"}, Params())
    assert e.value.code == 0
    assert (tmp_path / "final_candidates.fa").read_text() == "" and (tmp_path / "cluster_info.tsv").read_text() == ""


def test_read_tables_equal_the_per_read_functions_with_qualities():
    """The quality-based probabilities on the read tables (hypothesis_test_module._ccs_probabilities_on_table) against the
    per-read statement (functions.get_read_ccs_probabilities_c / _t + raghavan_upper_pvalue_bound): bit-equal p-values,
    the same number of informative reads, the same exceptions."""
    import random
    from isocon_amd import ccs_info as CI
    from isocon_amd import hypothesis_test_module as H
    from oracle import oracle as O
    rng = random.Random(12)

    def mut(b, n):
        v = list(b)
        for _ in range(n):
            p = rng.randrange(len(v))
            r = rng.random()
            if r < 0.4:
                v[p] = rng.choice("ACGT")
            elif r < 0.7:
                del v[p]
            else:
                v.insert(p, v[p] if rng.random() < 0.5 else rng.choice("ACGT"))
        return "".join(v)

    def aln(a, b, **kw):
        return O.parasail_alignment(a, b, 0, 0, **kw)[2]

    compared = dropped = 0
    for trial in range(150):
        t = "".join(rng.choice("AACGTT") for _ in range(rng.randint(40, 150)))
        c = mut(t, rng.randint(1, 3))
        if c == t:
            continue
        reads, reads_c, reads_t = {}, {}, {}
        for k in range(rng.randint(1, 8)):
            x = c if rng.random() < 0.5 else mut(c, rng.randint(0, 2))
            reads["c%d" % k] = x
            reads_c["c%d" % k] = aln(c, x)
        for k in range(rng.randint(1, 9)):
            src = t if rng.random() < 0.6 else c
            x = src if rng.random() < 0.5 else mut(src, rng.randint(0, 2))
            reads["t%d" % k] = x
            reads_t["t%d" % k] = aln(t, x)
        ccs = {a: CI.CCS(a, s, [rng.randint(3, 70) for _ in s], "NA") for a, s in reads.items()}
        tc = aln(t, c, opening_penalty=3, mismatch_penalty=-3, gap_ext=1)
        ct = aln(c, t, opening_penalty=3, mismatch_penalty=-3, gap_ext=1)

        def call(fn, *a):
            try:
                return fn(*a)
            except (AssertionError, IndexError, SystemExit) as e:
                return type(e).__name__

        slow = call(H._test_on_alignments, t, c, tc, ct, reads_c, reads_t, ccs, 43)
        fast = call(H._test_on_tables, t, c, tc, ct, H._ReadTable(len(c), reads_c), H._ReadTable(len(t), reads_t), ccs, 43)
        if isinstance(slow, str) or isinstance(fast, str):
            assert slow == fast, (trial, slow, fast)
            continue
        assert list(slow[0].items()) == list(fast[0].items())
        assert slow[1] == fast[1] and len(slow[2]) == fast[2] and slow[3] == fast[3], (trial, slow[1:], fast[1:])
        compared += slow[1] not in (0.0, 1.0)
        dropped += slow[3] < len(reads)
    assert compared > 25 and dropped > 10
