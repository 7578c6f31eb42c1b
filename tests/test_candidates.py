"""The candidate-inference phase end to end (BASELINE config[0]: the reference's test data through the pipeline's first
phase) against what the reference's own find_candidate_transcripts produces (tests/golden/g12_candidates.json,
ignore_ends_len = 0 and the default 15): converged candidates, read -> candidate alignments, reads to realign, number of steps."""
import glob
import hashlib
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_G12 = json.load(open(os.path.join(HERE, "golden", "g12_candidates.json")))
G12 = [dict(c, S=_G12["inputs"][c["input"]]) for c in _G12["cases"]]


def sha(s):
    return hashlib.sha1(s.encode()).hexdigest()[:16]


def run(case, tmp_path):
    from isocon_amd import isocon_get_candidates as IGC
    read_file = tmp_path / "reads.fa"
    read_file.write_text("".join(">%s\n%s\n" % (a, s) for a, s in case["S"]))

    class Params(object):
        nr_cores = 1
        neighbor_search_depth = 2 ** 32
        verbose = False
        develop_logfile = None
        logfile = None
        min_exon_diff = 20
        ignore_ends_len = case.get("ignore_ends_len", 0)
        min_candidate_support = 2
        is_fastq = False
        ccs = None
        outfolder = str(tmp_path)

    cand_file, read_partition, to_realign = IGC.find_candidate_transcripts(str(read_file), Params())
    cands, acc = [], None
    for line in open(cand_file):
        if line.startswith(">"):
            acc = line[1:].strip()
        else:
            cands.append([acc, sha(line.strip()), len(line.strip())])
    steps = 1 + len(glob.glob(os.path.join(str(tmp_path), "candidates_step_*.fa")))
    rp = sorted([c, r, sha(t[0]), sha(t[1]), list(t[2])] for c in read_partition for r, t in read_partition[c].items())
    return {"candidates": cands, "read_partition": rp, "to_realign": sorted(to_realign), "steps": steps}


@pytest.mark.parametrize("case", [c for c in G12 if c["name"].startswith("synth")], ids=[c["name"] for c in G12 if c["name"].startswith("synth")])
def test_candidate_inference_with_the_oracle_kernels(case, tmp_path, monkeypatch):
    import isocon_amd.SW_alignment_module as SWM
    import isocon_amd.edlib_alignment_module as EAM
    from isocon_amd import graphs
    from isocon_amd import isocon_get_candidates as IGC
    from oracle import oracle as O
    from isocon_amd import correction_module as COR
    from oracle import correction as OC
    monkeypatch.setattr(COR, "_correct_on_device", OC.correct_rows)     # no GPU here: the numpy checker stands in for the kernels
    monkeypatch.setattr(graphs, "nearest_neighbor_graph", O)
    monkeypatch.setattr(IGC, "edlib_align_sequences", O.edlib_align_sequences)
    monkeypatch.setattr(IGC, "sw_align_sequences", O.sw_align_sequences)
    monkeypatch.setattr(EAM, "edlib_align_sequences_keeping_accession", O.edlib_align_sequences_keeping_accession)
    monkeypatch.setattr(SWM, "sw_align_sequences_keeping_accession", O.sw_align_sequences_keeping_accession)
    assert run(case, tmp_path) == case["expect"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", G12, ids=[c["name"] for c in G12])
def test_gpu_candidate_inference(case, tmp_path):
    assert run(case, tmp_path) == case["expect"]


@pytest.mark.gpu
def test_fastq_input_gives_the_same_candidates(tmp_path):
    """is_fastq = True reads the same records through readfq; nothing else may change."""
    from isocon_amd import isocon_get_candidates as IGC
    from isocon_amd import synth
    accs, seqs, _ = synth.make_reads(120, 400, 2, seed=97)
    (tmp_path / "a").mkdir(); (tmp_path / "q").mkdir()
    fa, fq = tmp_path / "reads.fa", tmp_path / "reads.fq"
    fa.write_text("".join(">%s\n%s\n" % (a, s) for a, s in zip(accs, seqs)))
    fq.write_text("".join("@%s\n%s\n+\n%s\n" % (a, s, "I" * len(s)) for a, s in zip(accs, seqs)))

    def params(is_fastq, out):
        class P(object):
            nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; logfile = None
            min_exon_diff = 20; ignore_ends_len = 15; min_candidate_support = 2; ccs = None
        P.is_fastq = is_fastq; P.outfolder = str(out)
        return P

    c1, rp1, tr1 = IGC.find_candidate_transcripts(str(fa), params(False, tmp_path / "a"))
    c2, rp2, tr2 = IGC.find_candidate_transcripts(str(fq), params(True, tmp_path / "q"))
    assert open(c1).read() == open(c2).read() and rp1 == rp2 and tr1 == tr2
