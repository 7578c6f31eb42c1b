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
bookkeeping on the host as in the reference.  With base qualities (a `ccs_dict`: FASTQ input) the error probabilities
come from the qualities (functions.get_read_ccs_probabilities_c / _t), computed on the same read tables."""
from __future__ import annotations

import decimal
import math

import numpy as np

from . import functions
from . import SW_alignment_module as SWM


def _clamp(a, hi):
    """a limited to [0, hi] (np.clip with less call overhead: this runs per variant and window column of every edge)"""
    return np.minimum(np.maximum(a, 0), hi)


class _ReadTable(object):
    """The stored alignments of one candidate's reads as flat arrays, so that the per-read quantities of a test --
    alignment column of a candidate position, error counts, window comparisons -- are computed once per round and for all
    reads at a time instead of once per (edge, read) as functions.get_support / get_read_errors do.  Same results
    (tests/test_stat_test.py::test_read_tables_equal_the_per_read_functions)."""

    def __init__(self, ref_len, read_alignments):
        self.__dict__.update(_build_tables([(ref_len, read_alignments)])[0].__dict__)

    def column_of(self, i, still_ok):
        """alignment column of candidate position i, per read (functions.get_support: c_seq_to_coord_in_almnt[i])"""
        if not -self.ref_len <= i < self.ref_len:
            if still_ok.any():
                raise IndexError("list index out of range")         # what the per-read statement raises for these reads
            return np.zeros(self.n, dtype=np.int64)
        if i < 0:
            i += self.ref_len
        return i + np.bincount(self.gap_row[self.gap_h <= i], minlength=self.n)

    def agree_with_candidate(self, variant_coords):
        """reads whose row equals the candidate's row over every variant window (get_support, reads of c)"""
        ok = np.ones(self.n, dtype=bool)
        for i, (_, _, u_v) in variant_coords.items():
            pos = self.column_of(i, ok)
            for w in range(-1, u_v + 1):
                col = pos + w
                valid = (col >= 0) & (col < self.len)
                ok &= ~(valid & self.diff[_clamp(self.off0 + col, self.total - 1)])
        return ok

    def show_snippets(self, variant_coords, snippets):
        """reads that show the other sequence's snippet at every variant (get_support, reads of t)"""
        ok = np.ones(self.n, dtype=bool)
        for i, (v_type, _, u_v) in variant_coords.items():
            pos = self.column_of(i, ok)
            snippet = np.frombuffer(snippets[i].encode("ascii"), dtype=np.uint8)
            before, after = (2, u_v) if v_type == "I" else (1, u_v + 1)
            lo = np.maximum(0, pos - before)
            hi = np.minimum(self.len, pos + after)
            match = np.maximum(hi - lo, 0) == len(snippet)
            for j in range(len(snippet)):
                match &= self.read[_clamp(self.off0 + lo + j, self.total - 1)] == snippet[j]
            ok &= match
        return ok


    # ---- pieces of the quality-based probabilities (functions._ccs_probabilities), per read and per variant ----
    def own_window_equal(self, pos, u_v):
        """aln_read[lo:hi] == aln_own[lo:hi] with lo = max(0, pos - 1), hi = pos + u_v + 1"""
        ok = np.ones(self.n, dtype=bool)
        for w in range(-1, u_v + 1):
            col = pos + w
            valid = (col >= 0) & (col < self.len)
            ok &= ~(valid & self.diff[_clamp(self.off0 + col, self.total - 1)])
        return ok

    def window_equals(self, pos, before, after, text):
        """aln_read[max(0, pos - before): pos + after] == text"""
        snippet = np.frombuffer(text.encode("ascii"), dtype=np.uint8)
        lo = np.maximum(0, pos - before)
        hi = np.minimum(self.len, pos + after)
        match = np.maximum(hi - lo, 0) == len(snippet)
        for j in range(len(snippet)):
            match &= self.read[_clamp(self.off0 + lo + j, self.total - 1)] == snippet[j]
        return match

    def read_bases_upto(self, pos):
        """number of read bases in aln_read[:pos + 1]"""
        if getattr(self, "_rgap_key", None) is None:
            g = np.flatnonzero(self.read == 45)
            row = np.searchsorted(self.off0, g, side="right") - 1
            self._rgap_key = row * (np.int64(1) << 32) + (g - self.off0[row])          # sorted: by row, then column
            self._rgap_first = np.searchsorted(row, np.arange(self.n + 1))
        rows = np.arange(self.n, dtype=np.int64)
        upto = np.searchsorted(self._rgap_key, rows * (np.int64(1) << 32) + np.clip(pos, -1, (1 << 31) - 1), side="right") - self._rgap_first[:-1]
        return pos + 1 - np.where(pos >= 0, upto, 0)

    def qualities(self, ccs_dict):
        """(flat qualities, offset of every read's record, record length, start of the read inside its record)"""
        if getattr(self, "_qual_of", None) is not ccs_dict:
            recs = [ccs_dict[acc] for acc in self.accs]
            lens = np.fromiter((len(r.qual) for r in recs), dtype=np.int64, count=self.n)
            off = np.zeros(self.n + 1, dtype=np.int64)
            np.cumsum(lens, out=off[1:])
            for r in recs:                                  # the record's quality list as an array, made once per record
                if getattr(r, "_qual_np_of", None) is not r.qual:
                    r._qual_np, r._qual_np_of = np.asarray(r.qual, dtype=np.int64), r.qual
            flat = np.concatenate([r._qual_np for r in recs]) if recs else np.zeros(0, dtype=np.int64)
            reads = self.read_rows_without_gaps()
            start = np.fromiter((r.seq.index(x) for r, x in zip(recs, reads)), dtype=np.int64, count=self.n)
            self._qual = (flat, off[:-1], np.fromiter((len(r.seq) for r in recs), dtype=np.int64, count=self.n), start)
            self._qual_of = ccs_dict
        return self._qual

    def read_rows_without_gaps(self):
        raw = self.read.tobytes().decode("ascii")
        o = self.off0.tolist() + [self.total]
        return [raw[o[r]:o[r + 1]].replace("-", "") for r in range(self.n)]


def _ccs_probabilities_on_table(tab, variant_coords, other_snippets, ccs_dict, ratios, max_phred_q_trusted, shifted_type, coord_when_other):
    """functions._ccs_probabilities for all reads of a table at once: (informative mask, probability per read)."""
    subs_ratio, ins_ratio, del_ratio = ratios
    assert len(variant_coords) > 0
    alive = np.ones(tab.n, dtype=bool)
    prob = np.ones(tab.n, dtype=np.float64)
    if tab.n == 0:
        return alive, prob
    flat, qoff, rec_len, rec_start = tab.qualities(ccs_dict)
    base = np.asarray([10 ** (-((q - 3) * (max_phred_q_trusted - 3.0) / (90.0) + 3) / 10.0) for q in range(94)], dtype=np.float64)
    for i, (v_type, _, u_v) in variant_coords.items():
        pos = tab.column_of(i, alive)
        shows_own = tab.own_window_equal(pos, u_v)
        if v_type == shifted_type:
            shows_other = tab.window_equals(pos, 2, u_v, other_snippets[i])
        else:
            shows_other = tab.window_equals(pos, 1, u_v + 1, other_snippets[i])
        assert not (alive & shows_own & shows_other).any()
        seen = tab.read_bases_upto(pos)
        read_coord = np.where(shows_own, seen - 1, seen + coord_when_other.get(v_type, -1))
        alive &= shows_own | shows_other
        coord = rec_start + read_coord                     # CCS.read_aln_to_ccs_coord
        if (alive & (coord > rec_len)).any():
            raise SystemExit("Index error: read position beyond its quality record")
        coord = np.where(coord == rec_len, coord - 1, coord)
        coord = np.where(coord < 0, coord + rec_len, coord)        # a negative list index counts from the end
        if (alive & ((coord < 0) | (coord >= rec_len))).any():
            raise IndexError("list index out of range")
        q = flat[np.clip(qoff + np.clip(coord, 0, np.maximum(rec_len - 1, 0)), 0, max(len(flat) - 1, 0))]
        p10 = base[np.clip(q, 0, 93)]
        if u_v > 1:
            p_error = p10
        elif v_type == "S":
            p_error = (p10 * subs_ratio) / 3.0
        elif v_type == "I":
            p_error = (p10 * ins_ratio) / 4.0
        else:
            p_error = p10 * del_ratio
        prob = np.where(alive, prob * p_error, prob)
    assert ((prob[alive] > 0.0) & (prob[alive] < 1.0)).all()
    return alive, prob


def _variants_of(aln_t, aln_c):
    start, end = functions.get_mask_start_and_end(aln_t, aln_c)           # indels in the ends are length differences, not variants
    a = np.frombuffer(aln_t.encode("ascii"), dtype=np.uint8)
    b = np.frombuffer(aln_c.encode("ascii"), dtype=np.uint8)
    return [(i, aln_t[i], aln_c[i]) for i in np.flatnonzero(a != b).tolist() if start <= i < end]


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


def _test_on_alignments(t_seq, c_seq, alignment_tc, alignment_ct, read_alignments_to_c, read_alignments_to_t, ccs_dict=None, max_phred_q_trusted=None):
    """hypothesis_test_module.py:92-171 after the two alignments: (variant_coords_t, p_value, supporting reads, reads used)."""
    aln_t, aln_c, variants = _candidate_vs_reference(alignment_tc, alignment_ct)
    variant_coords_t, variant_coords_c, alignment_c_to_t, alignment_t_to_c = functions.get_variant_coordinates(t_seq, c_seq, aln_t, aln_c, variants)
    reads_support = functions.get_support(read_alignments_to_c, variant_coords_c, read_alignments_to_t, variant_coords_t, alignment_c_to_t)
    if len(variants) == 0:      # identical up to the ignored ends
        return variant_coords_t, 0.0, reads_support, len(read_alignments_to_c) + len(read_alignments_to_t)
    errors = functions.get_read_errors(read_alignments_to_c, read_alignments_to_t)
    if ccs_dict:            # base qualities decide (FASTQ input): reads showing neither sequence at a variant drop out
        probability, _ = functions.get_read_ccs_probabilities_c(read_alignments_to_c, variant_coords_c, alignment_t_to_c, ccs_dict, errors, max_phred_q_trusted)
        probability.update(functions.get_read_ccs_probabilities_t(read_alignments_to_t, variant_coords_t, alignment_c_to_t, ccs_dict, errors, max_phred_q_trusted)[0])
    else:
        probability = functions.get_empirical_error_probabilities(len(t_seq), errors, variant_coords_t)
    if len(probability) == 0:
        assert len(reads_support) == 0
        p_value = 0.0
    else:
        p_value = raghavan_upper_pvalue_bound(probability, reads_support)
    return variant_coords_t, p_value, reads_support, len(probability)


def _build_tables(items):
    """_ReadTable objects for [(ref_len, read_alignments)]: one pass of array operations over the reads of all of them,
    the tables are slices of the shared arrays."""
    per, rows_ref, rows_read, end_ref, end_read = [0], [], [], [], []
    for _, ra in items:
        for v in ra.values():
            a, b = v[0], v[1]
            rows_ref.append(a)
            rows_read.append(b)
            # columns of the leading + trailing gap run of either row (they are not errors: read_errors_from_alignment)
            end_ref.append(2 * len(a) - len(a.lstrip("-")) - len(a.rstrip("-")))
            end_read.append(2 * len(b) - len(b.lstrip("-")) - len(b.rstrip("-")))
        per.append(len(rows_ref))
    n = len(rows_ref)
    lens = np.fromiter((len(r) for r in rows_ref), dtype=np.int64, count=n)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    off0 = off[:-1]
    ref = np.frombuffer("".join(rows_ref).encode("ascii"), dtype=np.uint8)
    read = np.frombuffer("".join(rows_read).encode("ascii"), dtype=np.uint8)
    diff = ref != read
    # gaps of the candidate's row: a gap with h candidate characters before it shifts every position >= h by one column
    g = np.flatnonzero(ref == 45)
    gap_row = np.searchsorted(off, g, side="right") - 1
    first_gap = np.searchsorted(gap_row, np.arange(n + 1))
    gap_h = (g - off0[gap_row]) - (np.arange(len(g), dtype=np.int64) - first_gap[gap_row])
    # Error counts between the end gaps (functions.read_errors_from_alignment).  A column never holds two gaps and an end
    # run of one row faces bases of the other, so: insertions = gaps of the candidate's row outside its end runs,
    # deletions = the same for the read's row, substitutions = the remaining differing columns.
    if n:
        gaps_ref = first_gap[1:] - first_gap[:-1]
        gaps_read = np.fromiter((r.count("-") for r in rows_read), dtype=np.int64, count=n)
        ins = gaps_ref - np.asarray(end_ref, dtype=np.int64)
        dele = gaps_read - np.asarray(end_read, dtype=np.int64)
        sub = np.add.reduceat(diff.view(np.uint8), off0, dtype=np.int64) - gaps_ref - gaps_read
    else:
        ins = dele = sub = np.zeros(0, dtype=np.int64)
    tables = []
    for k, (ref_len, ra) in enumerate(items):
        r0, r1 = per[k], per[k + 1]
        b0, b1 = int(off[r0]), int(off[r1])
        g0, g1 = int(first_gap[r0]), int(first_gap[r1])
        t = _ReadTable.__new__(_ReadTable)
        t.accs = list(ra)
        t.n = r1 - r0
        t.ref_len = ref_len
        t.len = lens[r0:r1]
        t.off0 = off0[r0:r1] - b0
        t.total = b1 - b0
        t.read = read[b0:b1]
        t.diff = diff[b0:b1]
        t.gap_row = gap_row[g0:g1] - r0
        t.gap_h = gap_h[g0:g1]
        t.ins, t.dele, t.sub = ins[r0:r1], dele[r0:r1], sub[r0:r1]
        tables.append(t)
    return tables


_TABLES = {}        # id(read-alignment dict) -> (the dict, its alignment tuples, table); reset by clear_tables()


def clear_tables():
    _TABLES.clear()


def _tables_for(wanted):
    """{id(read_alignments): table} for [(ref_seq, read_alignments)]; a table is rebuilt only when the candidate's reads or
    their alignments have changed since the last round (the loop keeps most partitions untouched from round to round)."""
    out, todo = {}, []
    for ref_seq, ra in wanted:
        if id(ra) in out:
            continue
        stamp = list(ra.values())           # the alignment tuples themselves (kept alive: identities cannot be recycled)
        hit = _TABLES.get(id(ra))
        if (hit is not None and hit[0] is ra and len(hit[1]) == len(stamp) and all(a is b for a, b in zip(hit[1], stamp))
                and hit[2].ref_len == len(ref_seq)):
            out[id(ra)] = hit[2]
        else:
            out[id(ra)] = None
            todo.append((ref_seq, ra, stamp))
    if len(_TABLES) > 200000:
        _TABLES.clear()
    for lo in range(0, len(todo), 2048):                 # bounded batches: the shared arrays stay small
        part = todo[lo:lo + 2048]
        for (ref_seq, ra, stamp), tab in zip(part, _build_tables([(len(r), a) for r, a, _ in part])):
            _TABLES[id(ra)] = (ra, stamp, tab)
            out[id(ra)] = tab
    return out


def _test_on_tables(t_seq, c_seq, alignment_tc, alignment_ct, tab_c, tab_t, ccs_dict=None, max_phred_q_trusted=None):
    """_test_on_alignments on the read tables of c and t: same tuple, the supporting reads as a count."""
    aln_t, aln_c, variants = _candidate_vs_reference(alignment_tc, alignment_ct)
    variant_coords_t, variant_coords_c, alignment_c_to_t, alignment_t_to_c = functions.get_variant_coordinates(t_seq, c_seq, aln_t, aln_c, variants)
    sup_c = np.flatnonzero(tab_c.agree_with_candidate(variant_coords_c))
    sup_t = np.flatnonzero(tab_t.show_snippets(variant_coords_t, alignment_c_to_t))
    n_support = len(sup_c) + len(sup_t)
    if len(variants) == 0:
        return variant_coords_t, 0.0, n_support, tab_c.n + tab_t.n
    if ccs_dict:
        # base qualities (functions.get_read_ccs_probabilities_c / _t): c's informative reads first, then t's
        subs = float(max(1.0, int(tab_t.sub.sum() + tab_c.sub.sum())))
        ins = float(max(1.0, int(tab_t.ins.sum() + tab_c.ins.sum())))
        del_ = float(max(1.0, int(tab_t.dele.sum() + tab_c.dele.sum())))
        tot_errors = subs + ins + del_
        ratios = (subs / tot_errors, ins / tot_errors, del_ / tot_errors)
        alive_c, prob_c = _ccs_probabilities_on_table(tab_c, variant_coords_c, alignment_t_to_c, ccs_dict, ratios, max_phred_q_trusted, "D", {"I": 0})
        alive_t, prob_t = _ccs_probabilities_on_table(tab_t, variant_coords_t, alignment_c_to_t, ccs_dict, ratios, max_phred_q_trusted, "I", {"D": 0, "I": -2})
        prob = np.concatenate([prob_c[alive_c], prob_t[alive_t]])
        if len(prob) == 0:
            assert n_support == 0
            return variant_coords_t, 0.0, n_support, 0
        assert alive_c[sup_c].all() and alive_t[sup_t].all()
        slot_c = np.cumsum(alive_c) - 1                       # position of a read of c among the informative ones
        slot_t = int(alive_c.sum()) + np.cumsum(alive_t) - 1
        return variant_coords_t, _raghavan_on_arrays(prob, np.concatenate([slot_c[sup_c], slot_t[sup_t]]) if n_support else None), n_support, len(prob)
    # error probabilities per read, t's reads first (functions.get_read_errors / get_empirical_error_probabilities)
    n_reads = tab_t.n + tab_c.n
    if n_reads == 0:
        assert n_support == 0
        return variant_coords_t, 0.0, n_support, 0
    delta_size = float(len(variant_coords_t))
    seg = float(len(t_seq))
    p_S = (np.maximum(np.concatenate([tab_t.sub, tab_c.sub]), delta_size) / seg) / 3.0
    p_I = (np.maximum(np.concatenate([tab_t.ins, tab_c.ins]), delta_size) / seg) / 4.0
    p_D = np.maximum(np.concatenate([tab_t.dele, tab_c.dele]), delta_size) / seg
    prob = np.ones(n_reads, dtype=np.float64)
    for v_type, _, u_v in variant_coords_t.values():
        if v_type == "S":
            prob *= p_S * u_v
        elif v_type == "I":
            prob *= np.minimum(0.5, p_I * u_v)
        elif v_type == "D":
            prob *= np.minimum(0.5, p_D * u_v)
    prob[prob >= 1.0] = 0.99999
    return variant_coords_t, _raghavan_on_arrays(prob, np.concatenate([tab_t.n + sup_c, sup_t]) if n_support else None), n_support, n_reads


def _raghavan_on_arrays(prob, supporters):
    """raghavan_upper_pvalue_bound for probabilities in dict order and the indices of the supporting reads (in the order of
    functions.get_support): logarithms with math.log per distinct probability, sums left to right like the reference's."""
    assert prob.max() <= 1.0 and prob.min() > 0.0
    uniq, inv = np.unique(prob, return_inverse=True)
    logs = [-math.log(p, 10) for p in uniq.tolist()]
    log_max = max(logs)
    assert log_max > 0
    w_u = np.asarray([l / log_max for l in logs], dtype=np.float64)
    weight = w_u[inv]
    m_sum = sum((w_u * uniq)[inv].tolist())
    y_sum = sum(weight[supporters].tolist()) if supporters is not None else 0
    return _raghavan_from_sums(m_sum, y_sum)


def arrange_alignments_new_no_realign(t_acc, c_acc, t_seq, c_seq, read_alignments_to_c, read_alignments_to_t, ccs_dict, ignore_ends_len, max_phred_q_trusted):
    """hypothesis_test_module.py:92-171 (single edge; the batch entry point is do_statistical_tests_per_edge)."""
    tc, ct = SWM._align_pairs([(t_seq, c_seq), (c_seq, t_seq)], [-3, -3], 2, 3, 1)
    return _test_on_alignments(t_seq, c_seq, tc, ct, read_alignments_to_c, read_alignments_to_t, ccs_dict, max_phred_q_trusted)


def _result(c_acc, t_acc, t_seq, variant_coords_t, p_value, reads_support, nr_reads_used, with_qualities=False):
    variant_types = ";".join("(" + str(v[0]) + "," + str(j) + "," + str(v[2]) + ")" for j, v in variant_coords_t.items())
    factor = 1.0 if with_qualities else get_correction_factor(t_seq, c_acc, variant_coords_t)      # :242-246
    return (c_acc, t_acc, p_value, factor, len(reads_support), nr_reads_used, variant_types)


def statistical_test(c_acc, t_acc, c_seq, t_seq, reads_to_c, read_alignments_to_t, read_alignments_to_c, ignore_ends_len, ccs_dict, max_phred_q_trusted):
    """hypothesis_test_module.py:217-247: (c_acc, t_acc, p_value, correction factor, supporting reads, reads used, variants)."""
    assert not (set(reads_to_c) & set(read_alignments_to_t))
    N_t = len(set(reads_to_c) | set(read_alignments_to_t))
    if N_t == 0:    # all reads of both went elsewhere in the realignment
        return c_acc, t_acc, 1.0, 1.0, 0, N_t, ""
    if ccs_dict:
        for x_acc in reads_to_c:
            assert reads_to_c[x_acc] == ccs_dict[x_acc].seq
    delta_t, p_value, reads_support, used = arrange_alignments_new_no_realign(t_acc, c_acc, t_seq, c_seq, read_alignments_to_c, read_alignments_to_t,
                                                                              ccs_dict, ignore_ends_len, max_phred_q_trusted)
    return _result(c_acc, t_acc, t_seq, delta_t, p_value, reads_support, used, bool(ccs_dict))


def do_statistical_tests_per_edge(nearest_neighbor_graph, C, X, read_partition, ccs_dict, params):
    """hypothesis_test_module.py:20-77: {c_acc: {t_acc: (p_value, correction factor, support, reads used, variants)}} for
    every edge c -> t of the graph (nr_cores is irrelevant here: one device batch for all edges)."""
    edges = [(c_acc, t_acc) for c_acc in nearest_neighbor_graph for t_acc in nearest_neighbor_graph[c_acc]]
    live = [(c, t) for c, t in edges if len(read_partition[c]) + len(read_partition[t]) > 0]
    pairs = []
    for c, t in live:
        pairs.append((C[t], C[c]))
        pairs.append((C[c], C[t]))
    alignments = SWM._align_pairs(pairs, [-3] * len(pairs), 2, 3, 1) if pairs else []
    of_edge = {e: (alignments[2 * i], alignments[2 * i + 1]) for i, e in enumerate(live)}
    p_values = {c_acc: {} for c_acc in nearest_neighbor_graph}
    tables = _tables_for([(C[acc], read_partition[acc]) for e in live for acc in e])
    for c_acc, t_acc in edges:
        if (c_acc, t_acc) not in of_edge:
            p_values[c_acc][t_acc] = (1.0, 1.0, 0, 0, "")
            continue
        assert not (set(read_partition[c_acc]) & set(read_partition[t_acc]))
        tc, ct = of_edge[(c_acc, t_acc)]
        if ccs_dict:
            for x_acc in read_partition[c_acc]:
                assert X[x_acc] == ccs_dict[x_acc].seq
        delta_t, p_value, n_support, used = _test_on_tables(C[t_acc], C[c_acc], tc, ct, tables[id(read_partition[c_acc])],
                                                            tables[id(read_partition[t_acc])], ccs_dict, getattr(params, "max_phred_q_trusted", None))
        p_values[c_acc][t_acc] = _result(c_acc, t_acc, C[t_acc], delta_t, p_value, range(n_support), used, bool(ccs_dict))[2:]
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
    return _raghavan_from_sums(sum([weight[acc] * probability[acc] for acc in probability]), sum([weight[x_i] for x_i in x_equal_to_one]))


def _raghavan_from_sums(m_sum, y_sum):
    with decimal.localcontext() as ctx:
        ctx.prec = 100
        m = decimal.Decimal(m_sum)
        y = decimal.Decimal(y_sum)
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
