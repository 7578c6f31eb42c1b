"""The candidate-inference phase: mirror of
/root/reference/modules/isocon_get_candidates.py:22-35 (`get_unique_seq_accessions`), :37-81 (`get_partition_alignments`:
exact edit distances -> semi-global alignments -> exon-difference filter -> the partition_alignments structure consumed
by correction_module) and :85-312 (`find_candidate_transcripts`: the loop around partitioning, alignment and correction,
the naming and collapse of the candidates, the read-to-candidate alignments).  Same signatures, same return shapes,
same files written; progress prints of the reference are not reproduced.  Everything heavy runs on the GPU through the sibling modules; the exon filter
works on the CIGAR ops the aligner just produced (isocon_amd.functions)."""
from __future__ import annotations

from . import functions
from .SW_alignment_module import sw_align_sequences
from .edlib_alignment_module import edlib_align_sequences

_EDLIB_ALIGN, _SW_ALIGN = edlib_align_sequences, sw_align_sequences


def get_unique_seq_accessions(S):
    """isocon_get_candidates.py:22-35: {seq: [acc, ...]} in first-appearance order."""
    from . import _lib
    H = _lib.pyhelp()
    if H is not None and hasattr(H, "group_keys_by_value") and type(S) is dict:
        try:
            return H.group_keys_by_value(S)          # (the same loop in C: 50 000 reads per correction step)
        except TypeError:                            # (values that are not exact str: the plain loop)
            pass
    seq_to_acc = {}
    for acc, seq in S.items():
        seq_to_acc.setdefault(seq, []).append(acc)
    return seq_to_acc


class AlignmentBatch(object):
    """The alignments of one get_partition_alignments call as arrays: pair p = (centre a[p], member b[p]) of `store`, its run-length
    CIGAR ops[ops_ptr[p] : ops_ptr[p + 1]] (isocon_sg_trace_batch), counts res[p, 3:6].  The gapped strings are only made when somebody
    reads them (LazyAlignment); the correction builds its multi-alignment matrix from the ops on the device (correction_module)."""

    def __init__(self, store, pairs, a, b, ops, ops_ptr, res):
        self.store, self.pairs, self.a, self.b, self.ops, self.ops_ptr, self.res = store, pairs, a, b, ops, ops_ptr, res
        self.rows_of = {}          # centre sequence -> indices of its pairs that survived the exon filter

    def strings(self, p):
        from .SW_alignment_module import _ops_to_alignment
        m, s = self.pairs[p]
        return _ops_to_alignment(self.ops[int(self.ops_ptr[p]):int(self.ops_ptr[p + 1])].tolist(), m, s)

    def alive(self):
        return getattr(self.store, "_h", None) is not None


class LazyAlignment(object):
    """partition_alignments[m][s] = (edit_distance, m_alignment, s_alignment, 1) (isocon_get_candidates.py:74) whose two gapped strings are
    expanded from the CIGAR ops when first read.  Indexing, iteration, len() and comparison behave like the tuple."""
    __slots__ = ("_batch", "_p", "_edit", "_strings")

    def __init__(self, batch, p, edit):
        self._batch, self._p, self._edit, self._strings = batch, p, edit, None

    def _tuple(self):
        if self._strings is None:
            self._strings = self._batch.strings(self._p)
        return (self._edit, self._strings[0], self._strings[1], 1)

    def __getitem__(self, i):
        if i == 0:
            return self._edit
        if i == 3:
            return 1
        return self._tuple()[i]

    def __iter__(self):
        return iter(self._tuple())

    def __len__(self):
        return 4

    def __eq__(self, other):
        return self._tuple() == (other._tuple() if isinstance(other, LazyAlignment) else other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        return repr(self._tuple())


class PartitionAlignments(dict):
    """{centre: {s: (edit_distance, aln_centre, aln_s, weight)}} as get_partition_alignments returns it, plus `.batch`: the same
    alignments as arrays (AlignmentBatch) for consumers inside this package."""
    batch = None


def _partition_alignments_from_ops(graph_partition, M, G_star, exon_filtered, params):
    """get_partition_alignments without gapped strings: distances, CIGAR ops and counts from the device, the exon filter on the ops
    (functions.py:23-50 through isocon_exon_filter_from_ops), LazyAlignment values.  Same content as the string path."""
    import numpy as np
    from . import _lib
    from .SW_alignment_module import TIE_POLICY
    from .store import store_for_pairs
    H = _lib.pyhelp()
    pairs = None
    if H is not None and hasattr(H, "flatten_pairs") and type(graph_partition) is dict:
        try:
            pairs = H.flatten_pairs(graph_partition)[0]          # [(centre, member)] in the dict's / the sets' iteration order, in C
        except TypeError:
            pairs = None
    if pairs is None:
        pairs = [(m, s) for m, members in graph_partition.items() for s in members]
    out = PartitionAlignments()
    if pairs:
        st, a, b, owned = store_for_pairs(pairs, only_remembered=True)
        if st is None:          # (not the remembered store: nothing to keep resident -- the plain path, and nothing was packed to find out)
            return None
        if bool((st.lens[a] == 0).any() or (st.lens[b] == 0).any()):
            raise ValueError("empty sequence in an alignment pair")
        ed = st.ed_pairs(a, b, None)                                   # EAM:111, unbounded
        rate = ed.astype(np.float64) / np.minimum(st.lens[a], st.lens[b]).astype(np.float64)          # SWM:102-109
        mismatch = np.where(rate <= 0.01, -1, np.where(rate <= 0.09, -2, -4)).astype(np.int8)
        ops, ops_ptr, res = st.sg_trace(a, b, mismatch, tie_policy=TIE_POLICY, ed_upper=ed)
        L = _lib.load()
        flags = np.zeros(len(pairs), dtype=np.uint8)
        ops_c = np.ascontiguousarray(ops if len(ops) else np.zeros(1, np.uint32), dtype=np.uint32)
        ptr_c = np.ascontiguousarray(ops_ptr, dtype=np.uint64)
        _lib.check(L.isocon_exon_filter_from_ops(ops_c.ctypes.data_as(_lib.u32p), ptr_c.ctypes.data_as(_lib.u64p), len(pairs), int(params.min_exon_diff),
                                                 int(params.ignore_ends_len), flags.ctypes.data_as(_lib.u8p)), "isocon_exon_filter_from_ops")
        batch = AlignmentBatch(st, pairs, a, b, ops, np.asarray(ops_ptr, dtype=np.int64), res)
        out.batch = batch
        edit = (res[:, 4] + res[:, 5]).tolist()                        # mismatches + indels (isocon_get_candidates.py:74)
        dropped = np.flatnonzero(flags).tolist()
        for p in dropped:
            exon_filtered.add(pairs[p][1])
        keep = flags == 0
    for m in M:
        out[m] = {m: (0, m, m, G_star.nodes[m]["degree"])}
    if pairs:
        if H is not None and hasattr(H, "lazy_rows") and all(m in out for m in graph_partition):
            keep_c = np.ascontiguousarray(keep, dtype=np.uint8)
            edit_c = np.ascontiguousarray(res[:, 4] + res[:, 5], dtype=np.int32)
            H.lazy_rows(LazyAlignment, batch, pairs, keep_c.ctypes.data, edit_c.ctypes.data, min(len(keep_c), len(edit_c)), out, batch.rows_of)
        else:
            for p in np.flatnonzero(keep).tolist():
                m, s = pairs[p]
                out[m][s] = LazyAlignment(batch, p, edit[p])
                batch.rows_of.setdefault(m, []).append(p)
    return out


def get_partition_alignments(graph_partition, M, G_star, exon_filtered, params):
    """isocon_get_candidates.py:37-81.  graph_partition: {centre: set(members)}, M: {centre: weight}, G_star: the NN
    graph (node attribute `degree`).  Returns {centre: {seq: (edit_distance, aln_centre, aln_seq, weight)}} and adds the
    sequences dropped for exon-sized differences to `exon_filtered`.
    When the partition comes from the store the NN search just remembered (the pipeline's case) the alignments stay CIGAR ops: the
    gapped strings of a value are expanded when somebody reads them, and correct_strings builds its matrices from the ops on the device."""
    if edlib_align_sequences is _EDLIB_ALIGN and sw_align_sequences is _SW_ALIGN:          # (tests swap these for the CPU oracle)
        fast = _partition_alignments_from_ops(graph_partition, M, G_star, exon_filtered, params)
        if fast is not None:
            return fast
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


def _log(message, logfile):
    if logfile is not None:
        logfile.write(message + "\n")


def find_candidate_transcripts(read_file, params):
    """isocon_get_candidates.py:85-312: the candidate-inference phase -- read the reads, then partition / align / correct
    until the reads have converged, name the resulting candidates, and align every read to its candidate.

    params: .is_fastq .nr_cores .neighbor_search_depth .min_exon_diff .ignore_ends_len .min_candidate_support .outfolder
    .verbose (.logfile / .develop_logfile optional).  Writes candidates_step_<k>.fa, candidates_converged.fa and an empty
    not_converged.fa into params.outfolder like the reference and returns (candidates_file_name, read_partition,
    to_realign).  Not provided: the CCS quality variant the reference has switched off."""
    import os

    from . import correction_module, partitions
    from .SW_alignment_module import sw_align_sequences_keeping_accession
    from .edlib_alignment_module import edlib_align_sequences_keeping_accession
    from .input_output import fasta_parser, fastq_parser

    def read_all():
        with open(read_file, "r") as fh:
            if params.is_fastq:
                return {acc: seq for (acc, seq, qual) in fastq_parser.readfq(fh)}
            return {acc: seq for (acc, seq) in fasta_parser.read_fasta(fh)}

    S = read_all()
    original_reads = dict(S)          # (the reference reads the file a second time after the loop, :259-261: the same mapping -- S's values are replaced, never changed)
    logfile = getattr(params, "logfile", None)
    step = 1
    exon_filtered = set()
    seq_to_acc = get_unique_seq_accessions(S)
    G_star, graph_partition, M, converged = partitions.partition_strings(S, params)
    partition_alignments = get_partition_alignments(graph_partition, M, G_star, exon_filtered, params)
    _log("nearest_neighbors and partition, step 1 done", logfile)

    two_steps_ago = [2 ** 28, 2 ** 28, 2 ** 28]          # guards against 2-cycles
    previous = [2 ** 28]
    while not converged:
        edit_distances = sorted(t[0] for inner in partition_alignments.values() for t in inner.values())
        if two_steps_ago == edit_distances:              # reads alternating between two equally good centres
            break
        if sum(edit_distances) > sum(previous) and max(edit_distances) > max(previous):
            break                                        # getting worse: corrected and re-corrected reads
        if all(ed == 0 for ed in edit_distances):
            break                                        # nothing left to correct (isolated nodes remain)
        S_prime, _ = correction_module.correct_strings(partition_alignments, seq_to_acc, {}, step, nr_cores=params.nr_cores,
                                                       verbose=params.verbose)
        for acc, s_prime in S_prime.items():
            S[acc] = s_prime
        seq_to_acc = get_unique_seq_accessions(S)
        step += 1
        S_to_align = {acc: seq for acc, seq in S.items() if seq not in exon_filtered}
        G_star, graph_partition, M, converged = partitions.partition_strings(S_to_align, params)
        partition_alignments = get_partition_alignments(graph_partition, M, G_star, exon_filtered, params)
        with open(os.path.join(params.outfolder, "candidates_step_" + str(step) + ".fa"), "w") as out_file:
            for i, m in enumerate(partition_alignments):
                N_t = sum(t[3] for t in partition_alignments[m].values())
                out_file.write(">{0}\n{1}\n".format("read" + str(i) + "_support_" + str(N_t), m))
        two_steps_ago = previous
        previous = edit_distances
        _log("correction, nearest_neighbors and partition, step {0} done".format(step), logfile)

    # candidates = the distinct corrected sequences, supported by the reads that became identical to them
    c_seq_to_read_acc = {}
    for read_acc, seq in S.items():
        c_seq_to_read_acc.setdefault(seq, []).append(read_acc)
    c_acc_to_seq, c_acc_to_support = {}, {}
    for i, m in enumerate(sorted(c_seq_to_read_acc)):
        N_t = partition_alignments[m][m][3] if m in partition_alignments else 1
        c_acc = "transcript_" + str(i) + "_support_" + str(N_t)
        c_acc_to_seq[c_acc] = m
        c_acc_to_support[c_acc] = N_t

    if params.ignore_ends_len > 0:          # candidates that differ only in their ends are merged into the best supported one
        from . import end_invariant_functions
        remaining = end_invariant_functions.collapse_candidates_under_ends_invariant(c_acc_to_seq, c_acc_to_support, params)
        for c_acc in remaining:
            c_seq = c_acc_to_seq[c_acc]
            for removed_c_acc in remaining[c_acc]:
                removed_c_seq = c_acc_to_seq[removed_c_acc]
                c_seq_to_read_acc[c_seq].extend(c_seq_to_read_acc[removed_c_seq])
                del c_acc_to_seq[removed_c_acc]
                del c_acc_to_support[removed_c_acc]
                del c_seq_to_read_acc[removed_c_seq]

    assert len(S) == len(original_reads)
    for c_acc in list(c_acc_to_seq.keys()):
        if c_acc_to_support[c_acc] < params.min_candidate_support:
            del c_seq_to_read_acc[c_acc_to_seq[c_acc]]
            del c_acc_to_seq[c_acc]
            del c_acc_to_support[c_acc]
    assigned = set(read_acc for c_seq in c_seq_to_read_acc for read_acc in c_seq_to_read_acc[c_seq])
    to_realign = {read_acc: original_reads[read_acc] for read_acc in set(original_reads.keys()) - assigned}

    candidates_file_name = os.path.join(params.outfolder, "candidates_converged.fa")
    with open(candidates_file_name, "w") as fh:
        for c_acc, c_seq in sorted(c_acc_to_seq.items()):
            fh.write(">{0}\n{1}\n".format(c_acc, c_seq))
    open(os.path.join(params.outfolder, "not_converged.fa"), "w").close()
    assert len(to_realign) + len(assigned) == len(original_reads)

    c_to_reads = {}
    for c_acc, c_seq in c_acc_to_seq.items():
        c_to_reads[c_acc] = {read_acc: (c_seq, original_reads[read_acc]) for read_acc in c_seq_to_read_acc[c_seq]}
    c_to_reads_edit_distances = edlib_align_sequences_keeping_accession(c_to_reads, nr_cores=params.nr_cores)
    read_partition = sw_align_sequences_keeping_accession(c_to_reads_edit_distances, nr_cores=params.nr_cores)
    filtered_reads = functions.filter_exon_differences(read_partition, params.min_exon_diff, params.ignore_ends_len)
    for read_acc in filtered_reads:
        to_realign[read_acc] = original_reads[read_acc]
    return candidates_file_name, read_partition, to_realign
