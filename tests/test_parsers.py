"""FASTA / FASTQ readers against what the reference's own parsers return for the same texts
(tests/golden/g10_parsers.json)."""
import io
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G10 = json.load(open(os.path.join(HERE, "golden", "g10_parsers.json")))


@pytest.mark.parametrize("case", G10["fasta"], ids=[c["name"] for c in G10["fasta"]])
def test_read_fasta(case):
    from isocon_amd.input_output import fasta_parser
    assert [list(r) for r in fasta_parser.read_fasta(io.StringIO(case["text"]))] == case["expect"]


@pytest.mark.parametrize("case", G10["fastq"], ids=[c["name"] for c in G10["fastq"]])
def test_readfq(case):
    from isocon_amd.input_output import fastq_parser
    assert [list(r) for r in fastq_parser.readfq(io.StringIO(case["text"]))] == case["expect"]


@pytest.mark.gpu
def test_store_from_fasta(tmp_path):
    from isocon_amd.input_output import fasta_parser
    p = tmp_path / "reads.fa"
    p.write_text(">a x\nACGTACGTAA\n>b\nACGT\nACGT\n>c\nACGTACGTAA\n>d\nACGTACG\n")
    accs, st = fasta_parser.store_from_fasta(str(p))
    assert accs == ["d", "b", "c"] and st.lens.tolist() == [7, 8, 10]
    best, rp, cols, _ = st.nn_graph()
    assert best.tolist() == [1, 1, 2]
