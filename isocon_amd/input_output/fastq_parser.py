"""FASTA/FASTQ reader with the record semantics of the reader the reference ships
(/root/reference/modules/input_output/fastq_parser.py:5-34, itself Heng Li's readfq): yields (name, seq, qual) with
qual = None for FASTA records; name = header without its first character, blanks replaced by '_'; multi-line sequences
and qualities; a quality block is as long as its sequence, so '@' and '+' may start a quality line.  Every line loses its
LAST character (the newline) -- including a final line that has none, exactly like the reference.  Written as an
explicit state machine from that behaviour; pinned by tests/golden/g10_parsers.json."""
from __future__ import annotations


def readfq(fp):
    """Generator of (name, sequence, quality-or-None) from an iterable of lines."""
    it = iter(fp)
    pending = None                 # a header line already consumed while reading the previous record
    while True:
        header = pending
        pending = None
        if header is None:
            for line in it:
                if line[:1] in (">", "@"):
                    header = line[:-1]
                    break
            if header is None:
                return
        name = header[1:].replace(" ", "_")
        seq_parts = []
        stop = None
        for line in it:
            if line[:1] in ("@", "+", ">"):
                stop = line[:-1]
                break
            seq_parts.append(line[:-1])
        seq = "".join(seq_parts)
        if stop is None:                      # end of file inside / after the sequence
            yield name, seq, None
            return
        if stop[:1] != "+":                   # next record starts: this one was FASTA
            yield name, seq, None
            pending = stop
            continue
        qual_parts, have, complete = [], 0, False
        for line in it:
            qual_parts.append(line[:-1])
            have += len(line) - 1
            if have >= len(seq):
                complete = True
                break
        if not complete:                      # file ended before the quality was complete: reported without quality
            yield name, seq, None
            return
        yield name, seq, "".join(qual_parts)
