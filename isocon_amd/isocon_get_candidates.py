"""The two helpers of the candidate-inference loop that sit directly on the hot path:
mirror of /root/reference/modules/isocon_get_candidates.py:22-35 (`get_unique_seq_accessions`) and :37-81
(`get_partition_alignments`: exact edit distances -> semi-global alignments -> exon-difference filter -> the
partition_alignments structure consumed by correction_module).  Same signatures, same return shapes; progress prints
of the reference are not reproduced.  Everything heavy runs on the GPU through the sibling modules; the exon filter
works on the CIGAR ops the aligner just produced (isocon_amd.functions)."""
from __future__ import annotations

from . import functions
from .SW_alignment_module import sw_align_sequences
from .edlib_alignment_module import edlib_align_sequences


def get_unique_seq_accessions(S):
    """isocon_get_candidates.py:22-35: {seq: [acc, ...]} in first-appearance order."""
    seq_to_acc = {}
    for acc, seq in S.items():
        seq_to_acc.setdefault(seq, []).append(acc)
    return seq_to_acc


def get_partition_alignments(graph_partition, M, G_star, exon_filtered, params):
    """isocon_get_candidates.py:37-81.  graph_partition: {centre: set(members)}, M: {centre: weight}, G_star: the NN
    graph (node attribute `degree`).  Returns {centre: {seq: (edit_distance, aln_centre, aln_seq, weight)}} and adds the
    sequences dropped for exon-sized differences to `exon_filtered`."""
    exact_edit_distances = edlib_align_sequences(graph_partition, nr_cores=params.nr_cores)
    exact_alignments = sw_align_sequences(exact_edit_distances, nr_cores=params.nr_cores)
    filtered = functions.filter_exon_differences(exact_alignments, params.min_exon_diff, params.ignore_ends_len)
    exon_filtered.update(filtered)

    partition_alignments = {}
    for m in M:
        selfdegree = G_star.nodes[m]["degree"]
        partition_alignments[m] = {m: (0, m, m, selfdegree)}
        if m not in exact_alignments:
            continue
        for s in exact_alignments[m]:
            aln_m, aln_s, (matches, mismatches, indels) = exact_alignments[m][s]
            partition_alignments[m][s] = (mismatches + indels, aln_m, aln_s, 1)
    return partition_alignments
