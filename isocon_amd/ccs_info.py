"""Base qualities of the reads for the statistical test: mirror of
/root/reference/modules/ccs_info.py:9-56 (`CCS`, `read_aln_to_ccs_coord`), :131-151 (`fix_quality_values`) and :153-216
(`modify_strings_and_acc_fastq`).  Host-side bookkeeping (one object per read).  Qualities from a BAM file
(`get_ccs`, `modify_strings_and_acc`; pysam) are not provided."""
from __future__ import annotations

import itertools
import sys


class CCS(object):
    """ccs_info.py:9-24: a read with its PHRED qualities (integers 0..93)."""

    def __init__(self, name, seq, qual, np):
        self.name = name
        self.seq = seq
        self.qual = qual
        if any(val < 0 or val > 93 for val in qual):
            print(name, "has a quality value outside 0..93")
            sys.exit()
        self.np = np
        self.subreads = {}

    def read_aln_to_ccs_coord(self, read_aln, pos):
        """ccs_info.py:37-56: position in this record of base `pos` of the read whose gapped row is read_aln (one past the
        last base falls back on the last base)."""
        fasta_seq = read_aln.replace("-", "")
        index = self.seq.index(fasta_seq)
        if index + pos < len(self.seq):
            return index + pos
        if index + pos == len(self.seq):
            return index + pos - 1
        print("Index error:", index, "length seq_piece:", len(fasta_seq), "length sequence:", len(self.seq))
        sys.exit()


def fix_quality_values(seq, qualities):
    """ccs_info.py:131-151: inside every homopolymer run the qualities in ascending order (the uncertainty of a run's
    length is booked on its first bases whatever the strand)."""
    assert len(seq) == len(qualities)
    out, at = [], 0
    for _, run in itertools.groupby(seq):
        n = sum(1 for _ in run)
        out.extend(sorted(qualities[at:at + n]))
        at += n
    return out


def modify_strings_and_acc_fastq(ccs_dict_raw, X_ids, X):
    """ccs_info.py:153-216: keep the records of the reads in X (X_ids: first word of the accession -> accession), cut to
    the read's sequence, keyed by the full accession; strand=- records get fix_quality_values."""
    assert len(X_ids) == len(X)
    for q_id in list(ccs_dict_raw.keys()):
        if q_id not in X_ids:
            del ccs_dict_raw[q_id]
            continue
        q_acc = X_ids[q_id]
        record = ccs_dict_raw[q_id]
        qualities = fix_quality_values(record.seq, record.qual) if "strand=-" in q_acc else list(record.qual)
        start = record.seq.index(X[q_acc])
        stop = start + len(X[q_acc])
        record.seq = record.seq[start:stop]
        record.qual = qualities[start:stop]
        assert record.seq == X[q_acc] and len(record.seq) == len(record.qual)
        del ccs_dict_raw[q_id]
        record.name = q_acc
        ccs_dict_raw[q_acc] = record
    assert len(ccs_dict_raw) == len(X_ids)
    return ccs_dict_raw
