"""The per-edge hypothesis test of the statistical-test phase: mirror of
/root/reference/modules/hypothesis_test_module.py:20-77 (`do_statistical_tests_per_edge`), :92-171
(`arrange_alignments_new_no_realign`), :217-247 (`statistical_test`), :253-329 (`raghavan_upper_pvalue_bound`) and
:331-343 (`get_correction_factor`).

For a candidate c and its reference t: align the two (semi-global, match 2, mismatch -3, gap open 3, extend 1 -- the
reference's second gap model, hypothesis_test_module.py:99,103) in both orders and keep the order with fewer differing
columns outside the end gaps; locate the variants on both sequences; count the reads (of c and of t, from their stored
alignments) that carry all of them; bound the probability of that many supporters arising from sequencing errors alone.

The reference aligns the two orders of every edge one call at a time (or one Pool task per edge); here ALL edges of a
round go to the GPU as one batch of isocon_sg_strings_batch (2 x edges alignments), everything after that is string
bookkeeping on the host as in the reference.  Quality-value based probabilities (a `ccs_dict`, FASTQ / BAM input of the
reference) are not provided: a non-empty ccs_dict raises."""
from __future__ import annotations

import decimal
import math

from . import functions
from . import SW_alignment_module as SWM


def _variants_of(aln_t, aln_c):
    start, end = functions.get_mask_start_and_end(aln_t, aln_c)           # indels in the ends are length differences, not variants
    return [(i, p_t, p_c) for i, (p_t, p_c) in enumerate(zip(aln_t, aln_c)) if p_t != p_c and start <= i < end]


def _candidate_vs_reference(alignment_tc, alignment_ct):
    """hypothesis_test_module.py:99-110: (aln_t, aln_c, variants) from the t-vs-c alignment unless the c-vs-t alignment has
    strictly fewer variants."""
    aln_t, aln_c = alignment_tc[0], alignment_tc[1]
    variants = _variants_of(aln_t, aln_c)
    aln_c_flip, aln_t_flip = alignment_ct[0], alignment_ct[1]
    variants_flipped = _variants_of(aln_t_flip, aln_c_flip)
    if len(variants_flipped) < len(variants):
        return aln_t_flip, aln_c_flip, variants_flipped
    return aln_t, aln_c, variants


def _test_on_alignments(t_seq, c_seq, alignment_tc, alignment_ct, read_alignments_to_c, read_alignments_to_t):
    """hypothesis_test_module.py:92-171 after the two alignments: (variant_coords_t, p_value, supporting reads, reads used)."""
    aln_t, aln_c, variants = _candidate_vs_reference(alignment_tc, alignment_ct)
    variant_coords_t, variant_coords_c, alignment_c_to_t, alignment_t_to_c = functions.get_variant_coordinates(t_seq, c_seq, aln_t, aln_c, variants)
    reads_support = functions.get_support(read_alignments_to_c, variant_coords_c, read_alignments_to_t, variant_coords_t, alignment_c_to_t)
    if len(variants) == 0:      # identical up to the ignored ends
        return variant_coords_t, 0.0, reads_support, len(read_alignments_to_c) + len(read_alignments_to_t)
    errors = functions.get_read_errors(read_alignments_to_c, read_alignments_to_t)
    probability = functions.get_empirical_error_probabilities(len(t_seq), errors, variant_coords_t)
    if len(probability) == 0:
        assert len(reads_support) == 0
        p_value = 0.0
    else:
        p_value = raghavan_upper_pvalue_bound(probability, reads_support)
    return variant_coords_t, p_value, reads_support, len(probability)


def arrange_alignments_new_no_realign(t_acc, c_acc, t_seq, c_seq, read_alignments_to_c, read_alignments_to_t, ccs_dict, ignore_ends_len, max_phred_q_trusted):
    """hypothesis_test_module.py:92-171 (single edge; the batch entry point is do_statistical_tests_per_edge)."""
    if ccs_dict:
        raise NotImplementedError("quality-value based error probabilities (ccs_dict) are not provided")
    tc, ct = SWM._align_pairs([(t_seq, c_seq), (c_seq, t_seq)], [-3, -3], 2, 3, 1)
    return _test_on_alignments(t_seq, c_seq, tc, ct, read_alignments_to_c, read_alignments_to_t)


def _result(c_acc, t_acc, t_seq, variant_coords_t, p_value, reads_support, nr_reads_used):
    variant_types = ";".join("(" + str(v[0]) + "," + str(j) + "," + str(v[2]) + ")" for j, v in variant_coords_t.items())
    return (c_acc, t_acc, p_value, get_correction_factor(t_seq, c_acc, variant_coords_t), len(reads_support), nr_reads_used, variant_types)


def statistical_test(c_acc, t_acc, c_seq, t_seq, reads_to_c, read_alignments_to_t, read_alignments_to_c, ignore_ends_len, ccs_dict, max_phred_q_trusted):
    """hypothesis_test_module.py:217-247: (c_acc, t_acc, p_value, correction factor, supporting reads, reads used, variants)."""
    assert not (set(reads_to_c) & set(read_alignments_to_t))
    N_t = len(set(reads_to_c) | set(read_alignments_to_t))
    if N_t == 0:    # all reads of both went elsewhere in the realignment
        return c_acc, t_acc, 1.0, 1.0, 0, N_t, ""
    delta_t, p_value, reads_support, used = arrange_alignments_new_no_realign(t_acc, c_acc, t_seq, c_seq, read_alignments_to_c, read_alignments_to_t,
                                                                              ccs_dict, ignore_ends_len, max_phred_q_trusted)
    return _result(c_acc, t_acc, t_seq, delta_t, p_value, reads_support, used)


def do_statistical_tests_per_edge(nearest_neighbor_graph, C, X, read_partition, ccs_dict, params):
    """hypothesis_test_module.py:20-77: {c_acc: {t_acc: (p_value, correction factor, support, reads used, variants)}} for
    every edge c -> t of the graph (nr_cores is irrelevant here: one device batch for all edges)."""
    if ccs_dict:
        raise NotImplementedError("quality-value based error probabilities (ccs_dict) are not provided")
    edges = [(c_acc, t_acc) for c_acc in nearest_neighbor_graph for t_acc in nearest_neighbor_graph[c_acc]]
    live = [(c, t) for c, t in edges if len(read_partition[c]) + len(read_partition[t]) > 0]
    pairs = []
    for c, t in live:
        pairs.append((C[t], C[c]))
        pairs.append((C[c], C[t]))
    alignments = SWM._align_pairs(pairs, [-3] * len(pairs), 2, 3, 1) if pairs else []
    of_edge = {e: (alignments[2 * i], alignments[2 * i + 1]) for i, e in enumerate(live)}
    p_values = {c_acc: {} for c_acc in nearest_neighbor_graph}
    for c_acc, t_acc in edges:
        if (c_acc, t_acc) not in of_edge:
            p_values[c_acc][t_acc] = (1.0, 1.0, 0, 0, "")
            continue
        assert not (set(read_partition[c_acc]) & set(read_partition[t_acc]))
        tc, ct = of_edge[(c_acc, t_acc)]
        delta_t, p_value, reads_support, used = _test_on_alignments(C[t_acc], C[c_acc], tc, ct, read_partition[c_acc], read_partition[t_acc])
        p_values[c_acc][t_acc] = _result(c_acc, t_acc, C[t_acc], delta_t, p_value, reads_support, used)[2:]
    return p_values


def raghavan_upper_pvalue_bound(probability, x_equal_to_one):
    """hypothesis_test_module.py:253-329: Raghavan's bound for a weighted sum of independent Bernoulli variables,
    P(Y > m (1 + d)) < (e^d / (1 + d)^(1 + d))^m, with weights w_i = log10(p_i) / min log10(p) in (0, 1], Y = the summed
    weights of the supporting reads, m = E[Y]; evaluated as e^k / (1 + d)^(k + k / d), k = m d, in 100-digit decimals
    (the reference sets that precision module-wide)."""
    assert max(probability.values()) <= 1.0
    assert min(probability.values()) > 0.0
    log_probabilities = {acc: -math.log(p_i, 10) for acc, p_i in probability.items()}
    log_p_i_max = max(log_probabilities.values())
    assert log_p_i_max > 0
    weight = {acc: log_probabilities[acc] / log_p_i_max for acc in log_probabilities}
    with decimal.localcontext() as ctx:
        ctx.prec = 100
        m = decimal.Decimal(sum([weight[acc] * probability[acc] for acc in probability]))
        y = decimal.Decimal(sum([weight[x_i] for x_i in x_equal_to_one]))
        d = y / m - 1
        k = m * d
        if y == 0:
            bound = 1.0
        elif d == 0:
            bound = 0.5
        else:
            bound = k.exp() / (d + 1) ** (k + k / d)
        return float(bound)


def get_correction_factor(t_seq, c_acc, delta_t):
    """hypothesis_test_module.py:331-343: the number of candidates with the same numbers of substitutions, deletions and
    insertions relative to t (multiple-testing factor)."""
    m = len(t_seq)
    n_S = sum(1 for v in delta_t.values() if v[0] == "S")
    n_D = sum(1 for v in delta_t.values() if v[0] == "D")
    n_I = sum(1 for v in delta_t.values() if v[0] == "I")
    return ((4 * (m + 1)) ** n_I) * functions.choose(m, n_D) * functions.choose(3 * (m - n_D), n_S)
