"""FASTA reader with the reference's record semantics (/root/reference/modules/input_output/fasta_parser.py:1-19):
accession = header without '>' , stripped, blanks replaced by '_'; sequence lines stripped and concatenated; a record is
emitted when the next header (or the end of the file) is reached, also when its sequence is empty.  Written from the
behaviour, not from the source; pinned by tests/golden/g10_parsers.json."""
from __future__ import annotations


def read_fasta(fasta_file):
    """Generator of (accession, sequence) from an iterable of lines."""
    accession = None
    chunks = []
    for line in fasta_file:
        if line[:1] == ">":
            if accession is not None:
                yield accession, "".join(chunks)
            accession = line[1:].strip().replace(" ", "_")
            chunks = []
        elif accession is not None:
            chunks.append(line.strip())
        else:
            # text before the first header: the reference accumulates it into the first record's sequence
            chunks.append(line.strip())
    if accession is not None and accession != "":
        yield accession, "".join(chunks)


def store_from_fasta(path):
    """(accessions, SeqStore) of the unique sequences of a FASTA file in the length-sorted order the NN search needs
    (ties keep first appearance, the last accession of a duplicated sequence wins -- NNG:243-246)."""
    from ..store import SeqStore
    with open(path) as fh:
        seq_to_acc = {}
        for acc, seq in read_fasta(fh):
            seq_to_acc[seq] = acc
    seqs = sorted(seq_to_acc, key=len)
    return [seq_to_acc[s] for s in seqs], SeqStore(seqs)
