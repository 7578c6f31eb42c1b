#!/usr/bin/env python3
"""Golden fixture tests/golden/g10_parsers.json: small FASTA / FASTQ texts (written here) and what the reference's own
modules/input_output/{fasta_parser,fastq_parser}.py return for them.  Build container only."""
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from modules.input_output import fasta_parser as R_FA  # noqa: E402
from modules.input_output import fastq_parser as R_FQ  # noqa: E402

FASTA = {
    "plain": ">r1\nACGT\n>r2 with blanks  x\nAC\nGT\n\nTT\n>empty\n>last\nGGG\n",
    "no_trailing_newline": ">a\nACGT\n>b\nTTTT",
    "crlf_and_spaces": ">a b\r\nAC GT \r\n>c\r\n  TT\r\n",
    "only_header": ">solo\n",
    "empty_file": "",
}
FASTQ = {
    "plain": "@r1 desc\nACGT\n+\nIIII\n@r2\nAC\nGT\n+r2\nII\nII\n",
    "quality_starts_with_at_and_plus": "@r1\nACGTACGT\n+\n@III+III\n@r2\nTTTT\n+\n+@+@\n",
    "fasta_in_fastq_reader": ">f1 x\nACGT\nAC\n>f2\nGG\n",
    "mixed": ">f1\nACGT\n@q1\nAAAA\n+\n!!!!\n>f2\nCC\n",
    "truncated_quality": "@r1\nACGTACGT\n+\nIII\n",
    "no_trailing_newline": "@r1\nACGT\n+\nIIII\n@r2\nGGGG\n+\nJJJJ",
    "leading_junk": "junk line\n\n@r1\nAC\n+\nII\n",
    "empty_file": "",
}
out = {"generator": "tests/golden/make_golden_parsers.py", "fasta": [], "fastq": []}
for name, text in FASTA.items():
    out["fasta"].append({"name": name, "text": text, "expect": [list(r) for r in R_FA.read_fasta(io.StringIO(text))]})
for name, text in FASTQ.items():
    out["fastq"].append({"name": name, "text": text, "expect": [list(r) for r in R_FQ.readfq(io.StringIO(text))]})
json.dump(out, open(os.path.join(HERE, "g10_parsers.json"), "w"), indent=1)
for k in ("fasta", "fastq"):
    for c in out[k]:
        print(k, c["name"], c["expect"])
