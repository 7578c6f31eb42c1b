"""The statistical-test phase: mirror of
/root/reference/modules/isocon_statistical_test.py:152-536 (`stat_filter_candidates`, `product_with_check_overflow`).

Rounds until nothing changes: (re)assign unassigned reads to the surviving candidates (2-set NN search, exact distances,
semi-global alignments, exon filter -- the hot-path kernels), pick for every candidate the closest surviving candidates
of the static candidate-vs-candidate graph (built once with the infix kernel, end_invariant_functions), test each such
edge (hypothesis_test_module), drop the candidates whose largest corrected p-value is at or above the threshold (the
larger of p_value_threshold and the median corrected p-value), hand their reads to the next round.  One extra round
re-aligns every read ("to avoid local maxima") and adds homopolymer-equivalent candidates as references.  Same files
written: temp_candidates_step_<k>.fa, remaining_to_align.fa, p_values_<k>.tsv, candidates_after_step_<k>.fa,
final_candidates.fa, cluster_info.tsv.  FASTQ input (params.is_fastq) brings base qualities into the tests (ccs_info);
qualities from a BAM file (params.ccs, pysam) are not provided and raise.

Where the reference iterates a Python set (the reads of a candidate's partition), this module iterates the sorted set:
the reference's p-values depend on that order in their last digits (tests/golden/make_golden_stat_test.py)."""
from __future__ import annotations

import os
import sys

from . import ccs_info, end_invariant_functions, functions, hypothesis_test_module, partitions
from .SW_alignment_module import sw_align_sequences_keeping_accession
from .edlib_alignment_module import edlib_align_sequences_keeping_accession
from .input_output import fasta_parser, fastq_parser, write_output


def product_with_check_overflow(p_value, mult_factor_inv):
    """isocon_statistical_test.py:143-149: p-value x (integer) correction factor; 1.0 when the product overflows a float."""
    try:
        return p_value * mult_factor_inv
    except OverflowError:
        return 1.0


def _snapshot(read_partition):
    """What the reference gets from copy.deepcopy(read_partition) (:209, :483): the values are tuples of strings and
    integers, never modified in place, so copying the two dict levels is the same thing."""
    return {c_acc: dict(reads) for c_acc, reads in read_partition.items()}


def _assign_reads(to_realign, C, X, read_partition, params, remaining_file, candidate_file_name):
    """:269-313: reads without a candidate go to their nearest candidate; candidates left without reads disappear."""
    write_output.print_reads(remaining_file, to_realign)
    _, partition_of_realigned_reads = partitions.partition_strings_2set(to_realign, C, remaining_file, candidate_file_name, params)
    reassigned = {c_acc: {read_acc: (C[c_acc], X[read_acc]) for read_acc in sorted(reads)} for c_acc, reads in partition_of_realigned_reads.items()}
    edit_distances = edlib_align_sequences_keeping_accession(reassigned, nr_cores=params.nr_cores)
    alignments = sw_align_sequences_keeping_accession(edit_distances, nr_cores=params.nr_cores)
    functions.filter_exon_differences(alignments, params.min_exon_diff, params.ignore_ends_len)
    for c_acc in alignments:
        for read_acc in alignments[c_acc]:
            read_partition[c_acc][read_acc] = alignments[c_acc][read_acc]
    for c_acc in list(read_partition.keys()):
        if len(read_partition[c_acc]) == 0:
            del C[c_acc]
            del read_partition[c_acc]


def _tests_of_this_round(C, static_graph, add_homopolymer_edges):
    """:327-366: for every candidate its closest surviving candidates in the static graph."""
    graph = {}
    min_ed = None
    for c_acc in C:
        graph[c_acc] = {}
        if len(static_graph[c_acc]) > 0:
            alive = [ed for nbr, ed in static_graph[c_acc].items() if nbr in C]
            if alive:
                min_ed = min(alive)
            for nbr in static_graph[c_acc]:
                if nbr in C and static_graph[c_acc][nbr] == min_ed:
                    graph[c_acc][nbr] = min_ed
    if add_homopolymer_edges:
        for c_acc, nbrs in functions.get_homopolymer_invariants(C).items():
            graph.setdefault(c_acc, {})
            for t_acc in nbrs:
                graph[c_acc].setdefault(t_acc, 1)
    return graph


def _prune_tests(tests, read_partition, last_round_reads, last_round_tests, results, min_test_ratio):
    """:373-400.  Drops from `tests` (in place) the edges c -> t where c has at least min_test_ratio times the reads of t
    (it would pass anyway) and those that were scheduled last round with both read sets unchanged since (their result is
    carried over).  Returns (carried results per candidate, the edges scheduled this round per candidate)."""
    carried, scheduled = {}, dict(last_round_tests)
    for c_acc, row in tests.items():
        for t_acc in [t for t in row if len(read_partition[c_acc]) >= min_test_ratio * len(read_partition[t])]:
            del row[t_acc]
        same_c = None
        carried[c_acc] = {}
        for t_acc in list(row):
            if (c_acc, t_acc) not in last_round_tests[c_acc]:
                continue
            if same_c is None:
                same_c = last_round_reads[c_acc] == read_partition[c_acc]
            if same_c and last_round_reads[t_acc] == read_partition[t_acc]:
                carried[c_acc][t_acc] = results[c_acc][t_acc]
        scheduled[c_acc] = set((c_acc, t_acc) for t_acc in row)
        for t_acc in carried[c_acc]:
            del row[t_acc]
    return carried, scheduled


def _worst_test(c_acc, rows, n_reads):
    """:421-431: the test of a candidate with the largest corrected p-value (the last one among equals), as the tuple
    (c_acc, t_acc, p_value, correction factor, supporting reads, reads used, variants) the writers expect."""
    worst, worst_corrected = (c_acc, "", "not_tested", 1.0, n_reads, n_reads, ""), 0.0
    for t_acc, (p_value, factor, support, N_t, variants) in rows.items():
        corrected = product_with_check_overflow(p_value, factor)
        if corrected >= worst_corrected:
            worst, worst_corrected = (c_acc, t_acc, p_value, factor, support, N_t, variants), corrected
    return worst


def _rejection_threshold(worst, p_value_threshold):
    """:433-446: the larger of p_value_threshold and the median corrected p-value over the tested candidates."""
    tested = sorted(product_with_check_overflow(w[2], w[3]) for w in worst.values() if w[2] != "not_tested")
    if not tested:
        return p_value_threshold
    half = int(len(tested) / 2)
    median = (tested[half - 1] + tested[half]) / 2.0 if len(tested) % 2 == 0 else tested[half]
    return median if median > p_value_threshold else p_value_threshold


def stat_filter_candidates(read_file, candidate_file, read_partition, to_realign, params):
    """isocon_statistical_test.py:152-536.  read_partition: {c_acc: {read_acc: (c_aln, read_aln, (matches, mismatches,
    indels))}} and to_realign as returned by find_candidate_transcripts.  Returns the surviving candidates {acc: seq}."""
    if getattr(params, "ccs", None) and not params.is_fastq:       # the reference checks is_fastq first (:177-191): a FASTQ run uses its own qualities
        raise NotImplementedError("stat_filter_candidates: CCS quality values from a BAM file are not provided")
    if params.is_fastq:
        X_original = {acc: seq for (acc, seq, qual) in fastq_parser.readfq(open(read_file, "r"))}
    else:
        X_original = {acc: seq for (acc, seq) in fasta_parser.read_fasta(open(read_file, "r"))}
    assigned = set(x_acc for c_acc in read_partition for x_acc in read_partition[c_acc])
    X = {acc: seq for (acc, seq) in X_original.items() if acc in assigned or acc in to_realign}
    final_out_file_name = os.path.join(params.outfolder, "final_candidates.fa")
    tsv_info = os.path.join(params.outfolder, "cluster_info.tsv")
    if os.stat(candidate_file).st_size == 0:
        write_output.print_candidates(final_out_file_name, {}, {}, {}, {}, params, final=True, reads_to_consensus_tsv=tsv_info)
        sys.exit(0)
    C = {acc: seq for (acc, seq) in fasta_parser.read_fasta(open(candidate_file, "r"))}
    ccs_dict = {}
    if params.is_fastq:         # :180-192: the reads' base qualities, cut and keyed like X
        ccs_dict_raw = {x_acc.split(" ")[0]: ccs_info.CCS(x_acc.split(" ")[0], seq, [ord(ch) - 33 for ch in qual], "NA")
                        for (x_acc, seq, qual) in fastq_parser.readfq(open(read_file, "r"))}
        X_ids = {x_acc.split(" ")[0]: x_acc for x_acc in X}
        for x_acc in X:
            assert X_ids[x_acc.split(" ")[0]] == x_acc
        ccs_dict = ccs_info.modify_strings_and_acc_fastq(ccs_dict_raw, X_ids, X)
        for x_acc in X:
            assert X[x_acc] == ccs_dict[x_acc].seq

    candidates_nn_graph_static = end_invariant_functions.get_NN_graph_ignored_ends_edlib(C, params)

    modified = True
    step = 1
    last_round_reads = _snapshot(read_partition)
    last_round_tests = {c_acc: set() for c_acc in C}
    hypothesis_test_module.clear_tables()
    results = {}            # {c_acc: {t_acc: (p_value, correction factor, supporting reads, reads used, variants)}}
    worst = {}
    realignment_to_avoid_local_max = 0
    remaining_to_align_read_file = os.path.join(params.outfolder, "remaining_to_align.fa")
    while modified:
        modified = False
        temp_candidate_name = os.path.join(params.outfolder, "temp_candidates_step_{0}.fa".format(step))
        with open(temp_candidate_name, "w") as fh:
            for c_acc, c_seq in C.items():
                fh.write(">{0}\n{1}\n".format(c_acc, c_seq))

        if realignment_to_avoid_local_max == 1:         # the final round: every read is placed again
            to_realign = X
            read_partition = {c_acc: {} for c_acc in C}
        if to_realign:
            _assign_reads(to_realign, C, X, read_partition, params, remaining_to_align_read_file, temp_candidate_name)

        tests = _tests_of_this_round(C, candidates_nn_graph_static, realignment_to_avoid_local_max > 0)

        # :373-400: which of these tests are actually run
        kept_results, last_round_tests = _prune_tests(tests, read_partition, last_round_reads, last_round_tests, results, params.min_test_ratio)
        if any(tests.values()):
            for c_acc, row in hypothesis_test_module.do_statistical_tests_per_edge(tests, C, X, read_partition, ccs_dict, params).items():
                kept_results[c_acc].update(row)
        results = kept_results
        assert len(results) == len(C)

        worst = {c_acc: _worst_test(c_acc, rows, len(read_partition[c_acc])) for c_acc, rows in results.items()}     # :421-431
        limit = _rejection_threshold(worst, params.p_value_threshold)                                                 # :433-446

        # :448-480: candidates without support or not significant leave, their reads are placed again in the next round
        to_realign = {}
        with open(os.path.join(params.outfolder, "p_values_{0}.tsv".format(step)), "w") as tsv:
            for (c_acc, t_acc, p_value, factor, support, N_t, variants) in list(worst.values()):
                if p_value == "not_tested":
                    continue
                corrected = product_with_check_overflow(p_value, factor)
                if support == 0 or corrected >= limit:
                    to_realign.update((x_acc, X[x_acc]) for x_acc in read_partition.pop(c_acc))
                    del C[c_acc]
                    modified = True
                label = "_".join([c_acc, str(support), str(1.0 if support == 0 else min(1.0, corrected)), str(N_t), str(len(variants))])
                tsv.write(label + "\t" + str(p_value) + "\n")

        last_round_reads = _snapshot(read_partition)
        candidate_file = os.path.join(params.outfolder, "candidates_after_step_{0}.fa".format(step))
        step += 1
        if len(C) == 0:
            break
        write_output.print_candidates(candidate_file, C, worst, read_partition, X, params)

        if realignment_to_avoid_local_max == 1:
            realignment_to_avoid_local_max = 2
        elif not modified and realignment_to_avoid_local_max == 0:
            realignment_to_avoid_local_max = 1
            modified = True
        write_output.logger("Statistical test, step {0} done".format(step), getattr(params, "logfile", None))

    if params.ignore_ends_len > 0:          # :509-528: candidates equal up to their ends collapse once more
        c_acc_to_support = {c_acc: len(reads) for c_acc, reads in read_partition.items()}
        remaining = end_invariant_functions.collapse_candidates_under_ends_invariant(C, c_acc_to_support, params)
        for c_acc in remaining:
            for removed_c_acc in remaining[c_acc]:
                for read_acc, aln in read_partition[removed_c_acc].items():
                    read_partition[c_acc][read_acc] = aln
                del C[removed_c_acc]
                del c_acc_to_support[removed_c_acc]
                del read_partition[removed_c_acc]

    write_output.print_candidates(final_out_file_name, C, worst, read_partition, X, params, final=True, reads_to_consensus_tsv=tsv_info)
    hypothesis_test_module.clear_tables()
    return C
