"""GPU drop-in for /root/reference/modules/get_best_alignments.py (GBA): edit distances -> keep the minima ->
semi-global alignments -> keep the minima.  (Dead code in IsoCon v0.3.3; kept for the interface.)"""
from __future__ import annotations

from . import SW_alignment_module, edlib_alignment_module


def find_best_matches(approximate_matches, params, edge_creating_min_treshold=-1, edge_creating_max_treshold=2 ** 30):
    """GBA:5-119.  {s1: [s2, ...]} -> {s1: {s2: (edit_distance, s1_alignment, s2_alignment)}} (symmetrised)."""
    exact_edit_distances = edlib_alignment_module.edlib_align_sequences(approximate_matches, nr_cores=params.nr_cores)
    best_exact_edit_distances = {}
    for s1 in exact_edit_distances:
        for s2 in exact_edit_distances[s1]:
            edit_distance = exact_edit_distances[s1][s2]
            if edit_distance < edge_creating_max_treshold:
                best_exact_edit_distances.setdefault(s1, {})[s2] = edit_distance
                best_exact_edit_distances.setdefault(s2, {})[s1] = edit_distance
    for s1 in list(best_exact_edit_distances.keys()):
        min_edit_distance = min(best_exact_edit_distances[s1].values())
        for s2 in list(best_exact_edit_distances[s1].keys()):
            ed = best_exact_edit_distances[s1][s2]
            if ed > min_edit_distance and ed > edge_creating_min_treshold:
                del best_exact_edit_distances[s1][s2]
    cntrr = sum(len(v) for v in best_exact_edit_distances.values())
    filtered_tot_ed = sum(sum(v.values()) for v in best_exact_edit_distances.values())
    filtered_tot_ed / float(cntrr)  # GBA:60 raises ZeroDivisionError when no edge is left

    exact_alignments = SW_alignment_module.sw_align_sequences(best_exact_edit_distances, nr_cores=params.nr_cores)
    best_exact_matches = {}
    for s1 in exact_alignments:
        for s2 in exact_alignments[s1]:
            s1_alignment, s2_alignment, (matches, mismatches, indels) = exact_alignments[s1][s2]
            edit_distance = mismatches + indels
            if edit_distance < edge_creating_max_treshold:
                best_exact_matches.setdefault(s1, {})[s2] = (edit_distance, s1_alignment, s2_alignment)
                best_exact_matches.setdefault(s2, {})[s1] = (edit_distance, s2_alignment, s1_alignment)
    for s1 in list(best_exact_matches.keys()):
        min_edit_distance = min(v[0] for v in best_exact_matches[s1].values())
        for s2 in list(best_exact_matches[s1].keys()):
            ed = best_exact_matches[s1][s2][0]
            if ed > min_edit_distance and ed > edge_creating_min_treshold:
                del best_exact_matches[s1][s2]
    return best_exact_matches


def find_best_matches_2set(highest_paf_scores, X, C, params):
    """GBA:121-203.  {read_acc: [(score, t_acc), ...]} -> {x_acc: {c_acc: (edit_distance, x_alignment, c_alignment)}}."""
    approximate_matches = {}
    for read_acc, best_hits in highest_paf_scores.items():
        approximate_matches[read_acc] = {}
        for score, t_acc in best_hits:
            approximate_matches[read_acc][t_acc] = (X[read_acc], C[t_acc])
    exact_edit_distances = edlib_alignment_module.edlib_align_sequences_keeping_accession(approximate_matches, nr_cores=params.nr_cores)
    best_exact_edit_distances = {a1: dict(inner) for a1, inner in exact_edit_distances.items()}
    for s1_acc in list(best_exact_edit_distances.keys()):
        min_edit_distance = min(v[2] for v in best_exact_edit_distances[s1_acc].values())
        for s2_acc in list(best_exact_edit_distances[s1_acc].keys()):
            if best_exact_edit_distances[s1_acc][s2_acc][2] > min_edit_distance:
                del best_exact_edit_distances[s1_acc][s2_acc]
    exact_alignments = SW_alignment_module.sw_align_sequences_keeping_accession(best_exact_edit_distances, nr_cores=params.nr_cores)
    best_exact_matches = {}
    for x_acc in exact_alignments:
        for c_acc in exact_alignments[x_acc]:
            x_alignment, c_alignment, (matches, mismatches, indels) = exact_alignments[x_acc][c_acc]
            edit_distance = mismatches + indels
            if x_acc in best_exact_matches:
                current = next(iter(best_exact_matches[x_acc].values()))[0]
                if edit_distance < current:
                    best_exact_matches[x_acc] = {c_acc: (edit_distance, x_alignment, c_alignment)}
                elif edit_distance == current:
                    best_exact_matches[x_acc][c_acc] = (edit_distance, x_alignment, c_alignment)
            else:
                best_exact_matches[x_acc] = {c_acc: (edit_distance, x_alignment, c_alignment)}
    return best_exact_matches
