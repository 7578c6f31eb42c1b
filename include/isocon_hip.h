/*
 * isocon_hip.h -- C ABI of libisocon_hip.so: the MI355X (gfx950) implementation of IsoCon's all-pairs alignment +
 * nearest-neighbour-graph hot path.
 *
 * The reference (ksahlin/IsoCon v0.3.3) is pure Python and has NO FFI for this path: its arithmetic is reached
 * through the Python bindings of two third-party libraries (edlib, parasail).  Each entry point below therefore
 * cites the reference *call site(s)* it replaces (paths relative to the reference root); the Python modules in
 * isocon_amd/ bind them with ctypes and re-create the reference's function signatures on top
 * (see INTEGRATION.md).
 *
 * Conventions: plain pointers and sizes only; the caller owns every buffer; every function returns an
 * isocon_status (0 = OK, negative = error) and never throws; all calls block until results are in host memory.
 * Alphabet.  Sequences are upper-case ACGT in all shipped data.  edlib compares whatever characters it is given
 * (modules/edlib_alignment_module.py:111, modules/nearest_neighbor_graph.py:105), so the distance and nearest-neighbour entry points
 * take ANY bytes, with the same results as a textbook edit distance over those bytes ('N' equals 'N', 'a' differs from 'A'):
 *   - a set over at most four distinct symbols (lower case, RNA ...) is packed in two bits per base under its own symbol map and runs
 *     through the same kernels as ACGT (distances, nearest neighbours, infix alignments);
 *   - a set with MORE than four distinct symbols (ACGT + N, mixed case) keeps its bytes on the device beside the 2-bit planes, which then
 *     hold class-merged IMAGES of the sequences (lower case folded onto upper case, other bytes onto one code: d(images) <= d).  The
 *     nearest-neighbour search finds its candidates on the images with the ordinary kernels; a candidate pair in which a sequence holds a
 *     symbol outside ACGT gets its exact distance from a byte-wise kernel (one wavefront per pair, much slower per pair than the bit-vector
 *     kernels -- isocon_nn_stats.pairs_bytes counts them); pairs of two ACGT sequences are not affected.  isocon_ed_pairs and the
 *     nearest-neighbour entry points serve such a set; isocon_hw_pairs and the q-gram bound test entry points return ISOCON_E_ALPHABET.
 * The alignment and consensus entry points need ACGT (the reference builds its parasail matrix on "ACGT",
 * modules/SW_alignment_module.py:65; what parasail does off that alphabet is not pinned): ISOCON_E_ALPHABET on any other set.
 */
#ifndef ISOCON_HIP_H
#define ISOCON_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ISOCON_OK = 0,
    ISOCON_E_ARG = -1,        /* bad argument */
    ISOCON_E_ALPHABET = -2,   /* an entry point that needs ACGT (alignments, consensus) or 2-bit planes (infix, bound tests) on a set that is not such */
    ISOCON_E_HIP = -3,        /* HIP runtime error (isocon_last_error() has the text) */
    ISOCON_E_CAPACITY = -4,   /* caller buffer too small; required size written back */
    ISOCON_E_NODEVICE = -5,   /* no usable GPU */
    ISOCON_E_UNSUPPORTED = -6
} isocon_status;

typedef struct isocon_store isocon_store; /* opaque: packed sequence set resident in HBM */

/* Select the HIP device for subsequent calls of this process (one process per GPU). */
int isocon_init(int device_ordinal);
const char *isocon_strerror(int status);
const char *isocon_last_error(void);
int isocon_device_count(void);
/* Free the process-wide device scratch (trace buffers, hit lists ...) that calls keep for reuse. */
void isocon_release_scratch(void);

/*
 * Pack n sequences (ASCII, concatenated; sequence i = ascii[offsets[i] .. offsets[i+1])) into 2 bit-planes per
 * base, chunk-major ([64-base chunk][sequence][lo,hi] -> coalesced 16-B loads across consecutive sequences) and
 * upload them.  Replaces the per-task pickling of Python strings of the reference's Pool fan-out
 * (modules/nearest_neighbor_graph.py:33-35,65; modules/edlib_alignment_module.py:32; modules/SW_alignment_module.py:144).
 * For the nearest-neighbour entry points the sequences MUST be given in the reference's length-sorted order
 * (modules/nearest_neighbor_graph.py:246 / :208).
 */
int isocon_store_create(const uint8_t *ascii, const uint64_t *offsets, uint32_t n, isocon_store **out);
/* The same store from n separate buffers (seq_ptrs[i], seq_lens[i] bytes each): what a host language with its own string objects
 * has -- the library gathers them into its pinned staging buffer while the previous piece is on its way to the device, so the
 * caller does not have to build one contiguous copy first (125 MB at 50 000 x 2.5 kb). */
int isocon_store_create_ptrs(const uint8_t *const *seq_ptrs, const uint64_t *seq_lens, uint32_t n, isocon_store **out);
/* ... with options.  ISOCON_STORE_PRIVATE_SCRATCH: the store keeps a scratch pool of its own (bound matrix, held candidate edges, counters,
 * released with the store) instead of the process-wide one -- for a process that drives SEVERAL searches side by side, each of which
 * expects its scratch to survive the others' calls (the sharded protocol's phases: tests that run the ranks of isocon_amd/dist.py as
 * threads of one process).  One process per GPU -- the reference's Pool workers (modules/nearest_neighbor_graph.py:30-72) are processes
 * too -- needs no flag. */
#define ISOCON_STORE_PRIVATE_SCRATCH 1u
int isocon_store_create_ptrs_ex(const uint8_t *const *seq_ptrs, const uint64_t *seq_lens, uint32_t n, uint32_t flags, isocon_store **out);
/* Pinned host memory for the caller's large input / output buffers (gapped alignments: 2 x 130 MB at 50 000 pairs): copies between
 * the device and these buffers skip the library's staging.  NULL if the allocation fails.  Ordinary memory works everywhere too. */
void *isocon_host_alloc(uint64_t bytes);
void isocon_host_free(void *p);
void isocon_store_destroy(isocon_store *s);
uint32_t isocon_store_size(const isocon_store *s);
/* 64-bit digest of the whole packed set, order-sensitive (computed on the device from the planes and the lengths): the
 * ranks of a sharded run compare it before they split the work (isocon_amd/dist.py). */
int isocon_store_digest(const isocon_store *s, uint64_t *out);
uint64_t isocon_store_device_bytes(const isocon_store *s);

/*
 * Batched global (NW) edit distance over an explicit pair list.
 *   k[p] >= 0 : out[p] = distance if <= k[p], else -1      == edlib.align(x, y, mode="NW", task="distance", k=k)
 *               (modules/nearest_neighbor_graph.py:104-107)
 *   k[p] <  0 or k == NULL : unbounded distance            == edlib.align(x, y, "NW")
 *               (modules/edlib_alignment_module.py:111)
 * kernel_ms (optional) receives the summed HIP-event time of the kernels.
 */
int isocon_ed_pairs(isocon_store *s, const uint32_t *a, const uint32_t *b, const int32_t *k, uint64_t n_pairs,
                    int32_t *out_ed, float *kernel_ms);

/*
 * Lower bounds of the pairs' edit distances from q-gram count profiles (isocon_amd/csrc/qgram_mm.hpp: 9-grams hashed into 12288
 * presence bins plus 2 levels of 2048 excess bins; isocon_qgram_params): out_bound[p] <= ed(a[p], b[p]) always.  The main pass of the
 * nearest-neighbour search skips a pair whose bound exceeds its threshold -- the pair edlib would have answered with -1
 * (modules/nearest_neighbor_graph.py:156-162).  The reference has no counterpart; exposed so that the bound can be tested by itself.
 */
/* parameters of the stored q-gram profiles: out[0] = gram length q, out[1] = presence bins, out[2] = excess bins, out[3] = levels kept
 * of an excess bin; returns the binary elements of a profile (out[1] + out[2] * out[3]): the K of the bound kernel's contraction */
int isocon_qgram_params(int32_t *out);
int isocon_qgram_bound_pairs(isocon_store *s, const uint32_t *a, const uint32_t *b, uint64_t n_pairs, int32_t *out_bound);

/*
 * The SECOND lower bound of the main pass, for explicit pairs (tests, diagnostics): out_count[i] = the greedy number of pairwise disjoint
 * 8-grams of sequence partner[i] -- probed at the positions 0, s, 2 s, ... (probe_stride s = 4 or 2: the main pass runs 4 on every pair and 2 on
 * what is left) of the whole 16-base words of the sequence all of whose grams lie inside it -- that
 * occur NOWHERE in sequence owner[i].  Every such gram holds an edited position of any alignment of the two, so out_count[i] <=
 * edit distance; the main pass drops a surviving pair of its q-gram bound whose count exceeds the pair's threshold before any alignment
 * kernel sees it (csrc/nn_filter.hpp).  The reference evaluates those pairs with edlib and gets -1 (modules/nearest_neighbor_graph.py:156-162,
 * :171-178).  Same alphabet rule as isocon_qgram_bound_pairs.
 */
int isocon_block_bound_pairs(isocon_store *s, const uint32_t *owner, const uint32_t *partner, uint64_t n_pairs, int32_t probe_stride, int32_t *out_count);

/*
 * The bound matrix the main pass of isocon_nn_graph / isocon_nn_partial consults for the shard (q_begin, q_end, q_stride, q_block) --
 * see isocon_nn_partial -- of a length-sorted store, read back for tests: row r belongs to the shard's r-th entry q, its bytes
 * out_bounds[out_row_ptr[r] .. out_row_ptr[r + 1]) are min(255, bound) of the pairs (q, p), p = q + 1, q + 2, ... while
 * len(p) - len(q) <= 63 and p - q <= depth (the pairs the upward scan of modules/nearest_neighbor_graph.py:136-153 can reach within
 * 63 edits).  out_row_ptr has (number of rows + 1) entries.  ISOCON_E_CAPACITY + *n_bounds_needed when bounds_cap is too small.
 */
int isocon_qgram_bound_matrix(isocon_store *s, uint32_t q_begin, uint32_t q_end, uint32_t q_stride, uint32_t q_block, uint64_t depth,
                              uint64_t *out_row_ptr, uint8_t *out_bounds, uint64_t bounds_cap, uint64_t *n_bounds_needed);

/* statistics block filled by the nearest-neighbour entry points (all counters are for the one call) */
typedef struct {
    uint64_t pairs_evaluated;     /* (shared,lane) pairs the banded kernels actually ran */
    uint64_t cells_columns;       /* text columns processed, summed over evaluated lanes (lanes x wave columns) */
    uint64_t live_columns;        /* ... counting a lane only while its pair was still undecided */
    uint64_t tiles;               /* 64-lane tiles executed */
    uint64_t hits;                /* candidate edges recorded on the device */
    uint64_t fallback_queries;    /* queries that needed a band wider than 64 rows */
    uint64_t full_pairs;          /* pairs sent to the un-banded kernel */
    float kernel_ms;              /* HIP-event time of all kernels of the call */
    float scan_kernel_ms;         /* ... of the dominant kernel launch (64-row band scan, main pass) */
    float seed_kernel_ms;         /* ... of the seed pass that precedes it (1-set only) */
    uint32_t scan_launches;       /* launches summed into scan_kernel_ms */
    uint64_t pairs_prefiltered;   /* pairs of the main pass whose q-gram bound exceeded their threshold (never aligned) */
    float bound_kernel_ms;        /* HIP-event time of the q-gram profile + bound kernels (part of kernel_ms) */
    float list_kernel_ms;         /* ... of the kernel that collects the survivors of the bounds into lists (part of kernel_ms) */
    float lanes_kernel_ms;        /* ... of the one-pair-per-lane launch of the main pass (entries with few pairs; NOT in scan_kernel_ms) */
    float narrow_kernel_ms;       /* of scan_kernel_ms: the table launch over the pairs whose threshold is <= 31 (32-row form of the kernel) */
    uint64_t pairs_lanes;         /* pairs of the main pass aligned one pair per lane (the rest went through tables) */
    uint64_t bound_tiles;         /* 256 x 256 tiles of the bound matrix computed (each: 65 536 pairs x isocon_qgram_params() multiply-adds) */
    uint64_t pairs_wide_to_lanes; /* of pairs_lanes: pairs with a threshold above 31 sent there because of it (ISOCON_DEBUG_VARIANT=nn_narrow=1 only) */
    uint64_t narrow_columns;      /* of cells_columns: columns run by the 32-row form of the table kernel */
    uint64_t pairs_narrow;        /* pairs listed in chunks of the 32-row class */
    uint64_t pairs_bytes;         /* pairs with a sequence that holds symbols outside the 2-bit map, aligned on the bytes (see "Alphabet") */
    uint64_t pairs_block_rejected;/* survivors of the q-gram bound that the block filter rejected (greedy count of disjoint absent 8-grams > threshold; never aligned) */
    float filter_kernel_ms;       /* HIP-event time of the block filter (part of list_kernel_ms) */
    float mm_kernel_ms;           /* ... of the bound matrix contraction k_qgram_mm alone (part of bound_kernel_ms) */
} isocon_nn_stats;

/*
 * Exact nearest-neighbour graph over a length-sorted store.
 *   1-set (is_target == NULL): for every i with is_converged[i] == 0 the arg-min set of POSITIVE edit distances
 *     over all other sequences, restricted to d <= len(i) and to sorted-order offsets <= depth
 *     == get_nearest_neighbors (modules/nearest_neighbor_graph.py:110-198) for every query, i.e. the whole of
 *     get_exact_nearest_neighbor_graph (:19-82) independent of nr_cores.
 *   2-set (is_target != NULL): queries are the entries with is_target[i] == 0, neighbours only entries with
 *     is_target[i] == 1, distance 0 admitted == get_nearest_neighbors_2set (:341-424).
 *     depth must be >= the number of targets (the reference's default 2**32 never binds).
 * Output CSR: out_best[i] = minimal distance (-1 if row empty), out_row_ptr[n+1], out_cols = neighbour indices in
 * the reference's insertion order (ascending offset, lower index first).  If the edges do not fit cols_cap the
 * call returns ISOCON_E_CAPACITY and *n_cols_needed holds the required capacity.
 */
int isocon_nn_graph(isocon_store *s, const uint8_t *is_converged, const uint8_t *is_target, uint64_t depth,
                    int32_t *out_best, uint64_t *out_row_ptr, uint32_t *out_cols, uint64_t cols_cap,
                    uint64_t *n_cols_needed, isocon_nn_stats *stats);

/*
 * Sharded variant for one-process-per-GPU runs (the reference has no distributed path; its Pool chunking is
 * modules/nearest_neighbor_graph.py:33-35).  A rank OWNS the entries x of the length-sorted order with q_begin <= x < q_end and
 * (x - q_begin) mod q_stride < q_block (q_block a power of two, 1 <= q_block <= q_stride):
 *     rank r of N, block-cyclic:  q_begin = 256 r, q_end = n, q_stride = 256 N, q_block = 256 -- blocks of 256 consecutive entries
 *                                 (one tile row of the bound matrix) dealt round-robin: the very uneven windows balance by themselves
 *                                 and a rank's tiles stay as dense as on one GPU (what isocon_amd/dist.py uses);
 *     cyclic:                     q_begin = r, q_stride = N, q_block = 1;       contiguous: q_stride = q_block = 1.
 * best_inout[n] is IN/OUT (0x3fffffff = no neighbour known yet).
 *   phase 0: seed pass -- every owned entry against its 64 nearest longer neighbours (64-row band).  Pass best_inout all
 *            0x3fffffff.  (No-op for the 2-set graph.)
 *   phase 1: 64-row band over every remaining admissible pair whose LOWER index is owned (1-set and 2-set alike;
 *            only the long-read fallback of the 2-set search assigns a pair to the rank that owns its read);
 *            best_inout = element-wise MIN over all ranks' phase-0 results (tight thresholds on every rank).
 *   phase 3: phases 0 and 1 in one call, no exchange between them (single-rank callers; with several ranks the seeds of a rank's
 *            own rows leave its thresholds loose: summed work x 2 to x 4 on C3, which is why isocon_amd/dist.py reduces after phase 0).
 *   phase 2: 128/256/512-row bands over the pairs whose lower index is owned and that involve an entry still
 *            unresolved in best_inout (= MIN over all ranks' phase-1 results), then the un-banded kernel for the owned
 *            queries whose neighbour is further than 511 edits.  wide_queries (phase 2 only; NULL = the entries that are unresolved
 *            at the call): the queries of the WHOLE phase when the caller runs it in sub-steps over sub-shards with a reduction of
 *            best_inout after each (isocon_amd/dist.py does: a query's threshold then has seen part of its pairs on ALL ranks from
 *            the second sub-step on) -- wide_queries[i] != 0 for the entries that were unresolved when the phase began.
 * Each call returns up to hits_cap candidate edges (endpoint, neighbour, distance) as int32 triples.  The caller
 * min-reduces best over ranks after each phase, all-gathers the triples and calls isocon_nn_finalize.
 */
int isocon_nn_partial(isocon_store *s, const uint8_t *is_converged, const uint8_t *is_target, uint64_t depth,
                      uint32_t q_begin, uint32_t q_end, uint32_t q_stride, uint32_t q_block, int32_t phase, int32_t *best_inout,
                      int32_t *out_hits, uint64_t hits_cap, uint64_t *n_hits, isocon_nn_stats *stats, const uint8_t *wide_queries);
int isocon_nn_finalize(uint32_t n, const int32_t *best, const int32_t *hits, uint64_t n_hits,
                       int32_t *out_best, uint64_t *out_row_ptr, uint32_t *out_cols, uint64_t cols_cap,
                       uint64_t *n_cols_needed);

/*
 * The same protocol with the exchanged data resident in device memory (what isocon_amd/dist.py uses on a GPU: RCCL reduces and
 * gathers device buffers, nothing crosses PCIe between the exchange steps but a few status words).  All work is issued on the
 * NULL stream of the current device; pointers named *_dev are device pointers owned by the caller.
 *   isocon_nn_partial_dev  one phase like isocon_nn_partial; best_inout_dev[n] is read and updated in place; the phase's candidate
 *                          edges are appended to a list the library keeps in device memory (keep_hits 0: the list starts empty --
 *                          pass 0 for the first phase of a search); *n_hits_held = edges held after the call.
 *   isocon_nn_hits_dev     writes the held edges that attain best_dev[] of their endpoint to out_hits_dev (int32 triples), the
 *                          rest of the cap_rows rows as (-1, -1, -1): a fixed-size block for one all_gather.  cap_rows must be >=
 *                          the number of edges held (ISOCON_E_CAPACITY otherwise).
 *   isocon_nn_finalize_dev isocon_nn_finalize from device buffers (rows with a negative endpoint are skipped); n = size of the store.
 */
int isocon_nn_partial_dev(isocon_store *s, const uint8_t *is_converged, const uint8_t *is_target, uint64_t depth,
                          uint32_t q_begin, uint32_t q_end, uint32_t q_stride, uint32_t q_block, int32_t phase, int32_t *best_inout_dev,
                          int32_t keep_hits, uint64_t *n_hits_held, isocon_nn_stats *stats, const uint8_t *wide_queries);
int isocon_nn_hits_dev(isocon_store *s, const int32_t *best_dev, int32_t *out_hits_dev, uint64_t cap_rows, uint64_t *n_kept);
int isocon_nn_finalize_dev(isocon_store *s, const int32_t *best_dev, const int32_t *hits_dev, uint64_t n_rows,
                           int32_t *out_best, uint64_t *out_row_ptr, uint32_t *out_cols, uint64_t cols_cap, uint64_t *n_cols_needed);

/*
 * Batched semi-global affine alignment with traceback == parasail.sg_trace_scan_16/32(s1=a, s2=b, open, ext,
 * matrix_create("ACGT", match, mismatch)) + the CIGAR decode and column counting of parasail_alignment
 * (modules/SW_alignment_module.py:64-86).  mismatch is per pair (the reference picks it from the error-rate
 * bucket, :102-109).  tie_policy 0 = parasail's behaviour as restated in oracle/isocon_oracle.c.
 * Outputs: CIGAR ops as (len << 4 | code), code 0 '=', 1 'X', 2 'I' (consumes a), 3 'D' (consumes b);
 * out_ops_ptr[n_pairs+1]; out_res[6*p..] = score, end_query, end_ref, matches, mismatches, indels.
 * ed_upper (may be NULL; entries < 0 = unknown): an upper bound of the pair's edit distance -- the reference's
 * sw_align_sequences receives exactly that in its input dict (modules/SW_alignment_module.py:89-101).  It only
 * narrows the part of the DP matrix that is computed; the result is always the full-matrix result: a banded alignment
 * is accepted only if its score proves that nothing outside the band can reach it, otherwise the pair is redone in full.
 */
int isocon_sg_trace_batch(isocon_store *s, const uint32_t *a, const uint32_t *b, uint64_t n_pairs, int32_t match,
                          const int8_t *mismatch_per_pair, int32_t open, int32_t ext, int32_t tie_policy,
                          uint32_t *out_ops, uint64_t *out_ops_ptr, uint64_t ops_cap, uint64_t *n_ops_needed,
                          int32_t *out_res, float *kernel_ms, const int32_t *ed_upper);

/*
 * Same as isocon_sg_trace_batch plus the two gapped strings per pair (what cigar_to_seq builds from the CIGAR in the
 * reference, modules/SW_alignment_module.py:15-56,78): out_aln_a / out_aln_b hold, for pair p, the bytes
 * [out_aln_ptr[p], out_aln_ptr[p+1]) = the aligned query / reference with '-' for gaps (equal lengths).
 * ISOCON_E_CAPACITY + *n_aln_needed when aln_cap is too small.  Both sequences of every pair must be non-empty.
 */
int isocon_sg_strings_batch(isocon_store *s, const uint32_t *a, const uint32_t *b, uint64_t n_pairs, int32_t match,
                            const int8_t *mismatch_per_pair, int32_t open, int32_t ext, int32_t tie_policy,
                            uint32_t *out_ops, uint64_t *out_ops_ptr, uint64_t ops_cap, uint64_t *n_ops_needed,
                            int32_t *out_res, uint8_t *out_aln_a, uint8_t *out_aln_b, uint64_t *out_aln_ptr,
                            uint64_t aln_cap, uint64_t *n_aln_needed, float *kernel_ms, const int32_t *ed_upper);

/*
 * Where the kernel time of this thread's most recent isocon_sg_trace_batch / isocon_sg_strings_batch call went (HIP events on the
 * kernels' stream; a measurement aid for bench.py's `roofline_sw`, the reference has no counterpart: parasail's time is one number per
 * call, modules/SW_alignment_module.py:66-69).  out[0 .. min(cap, 11)): forward kernels (k_sg_band + k_sg_forward) ms, walk kernels ms,
 * compaction ms, string expansion ms, pairs aligned with the band's diagonals on the lanes, pairs aligned in strips, pairs whose
 * banded result could not be certified and was redone in full, bytes of trace scratch the forward kernels wrote, pairs of the fifth
 * value that ran on 128 diagonals (two per lane instead of four), those of them whose bound asks for more than 128 diagonals (run
 * narrower first: the certificate decides), and those of these that failed it and ran again in the band of their bound.  Returns the
 * number of values written.
 */
int isocon_sg_last_stats(double *out, int32_t cap);

/*
 * Exon-difference filter on CIGAR ops (host-only helper, no GPU): out_flag[p] = 1 iff filter_exon_differences
 * (modules/functions.py:23-50, mask rule :218-236) would drop the pair, i.e. one of the gapped strings has a run of
 * >= min_exon_diff gaps inside the window that excludes min(ignore_ends_len, end-gap) columns at either end.
 * ops / ops_ptr as returned by isocon_sg_trace_batch.
 */
int isocon_exon_filter_from_ops(const uint32_t *ops, const uint64_t *ops_ptr, uint64_t n_pairs, int32_t min_exon_diff,
                                int32_t ignore_ends_len, uint8_t *out_flag);

/* Consensus correction of one partition on its multi-alignment matrix -- replaces the column statistics and the
 * per-read correction of modules/correction_module.py:277-402 (position frequency matrix modules/functions.py:526-536).
 * matrix: n_rows x n_cols bytes, row-major, symbols 'A' 'C' 'G' 'T' '-' (layout of functions.create_multialignment_matrix);
 * degree[r]: multiplicity of row r (rows with degree > 1 are counted degree times and never corrected).
 * Per column: weighted counts, majority symbol (first maximum in the order A, C, G, T, -), unambiguous iff unique.
 * Per row of degree 1: the positions where an unambiguous majority differs from the row are correctable; with
 * frequency = count of the row's symbol in the column / partition total of that error class (insertion, deletion,
 * substitution; over the unambiguous columns; at least 1), every position whose frequency is <= the ceil(n/2)-th
 * smallest is replaced by the majority.  Output: the corrected rows without their '-' symbols, packed
 * (out_offsets[n_rows + 1]), out_n_cand[r] = number of correctable positions (rows with more than the 2048 the kernel
 * keeps in LDS are run again with their list in HBM), out_class_totals[3] = insertion, deletion, substitution totals.
 * ISOCON_E_CAPACITY if packed_cap is too small (out_offsets[n_rows] holds the size needed). */
int isocon_msa_correct(const uint8_t *matrix, uint32_t n_rows, uint32_t n_cols, const int32_t *degree,
                       uint8_t *out_packed, uint64_t packed_cap, uint64_t *out_offsets, int32_t *out_n_cand,
                       int64_t *out_class_totals, float *kernel_ms);

/*
 * The same correction with the multi-alignment matrix built ON THE DEVICE from the alignments' CIGAR ops (f1 + f3 fused with a12-a15:
 * the gapped strings of sw_align_sequences never exist).  Replaces modules/functions.py:543-588 (create_multialignment_matrix),
 * :598-631 (position_query_to_alignment) and the column layout of :679-767 for the partitions of get_partition_alignments
 * (modules/isocon_get_candidates.py:37-81) + correct_strings (modules/correction_module.py:12-75).
 *   isocon_msa_build_ops      rows = the partition: row_ids[0] the centre, row_ids[1 ..] its members (ids of the store); the ops of row r
 *                             (isocon_sg_trace_batch's encoding; centre = QUERY of the alignment) are ops[ops_ptr[r] .. ops_ptr[r + 1]),
 *                             ops_ptr[0] = ops_ptr[1] = 0.  Builds the matrix in device memory (kept for the call below) and returns
 *                             *out_n_cols, out_col_slot[len(centre) + 1] (first column of every insertion slot; optional),
 *                             out_longest[len(centre) + 1] (longest insertion per slot; optional) and the insertions of the WIDE slots
 *                             (longest >= 2) as records of 8 uint32 (row, slot, first position in the member, length, the 2-bit codes
 *                             A C G T = 0 1 2 3 of its first 32 bases in two words, two spare words): where those sit inside
 *                             their slot is decided by the reference's string heuristics (get_best_solution, functions.py:635-676),
 *                             which the Python layer applies and hands back as patches.  ISOCON_E_CAPACITY + *n_wide if wide_cap is
 *                             too small; ISOCON_E_ARG if the ops of a row do not spell both sequences.
 *   isocon_msa_correct_built  patches (patch_bytes[patch_ptr[i] .. patch_ptr[i + 1]) into row patch_row[i] from column patch_col[i]),
 *                             then isocon_msa_correct on the built matrix (same outputs).  One build serves one correction.
 *   isocon_msa_read_built     the built matrix (tests).
 */
int isocon_msa_build_ops(isocon_store *s, uint32_t n_rows, const uint32_t *row_ids, const uint32_t *ops, const uint64_t *ops_ptr,
                         uint32_t *out_n_cols, uint32_t *out_col_slot, uint32_t *out_longest, uint32_t *out_wide, uint64_t wide_cap,
                         uint64_t *n_wide, float *kernel_ms);
int isocon_msa_correct_built(isocon_store *s, uint32_t n_rows, uint32_t n_cols, const uint32_t *patch_row, const uint32_t *patch_col,
                             const uint32_t *patch_ptr, const uint8_t *patch_bytes, uint32_t n_patches, const int32_t *degree,
                             uint8_t *out_packed, uint64_t packed_cap, uint64_t *out_offsets, int32_t *out_n_cand,
                             int64_t *out_class_totals, float *kernel_ms);
int isocon_msa_read_built(isocon_store *s, uint32_t n_rows, uint32_t n_cols, uint8_t *out_matrix);

/*
 * The same two steps for ALL partitions of a correction step at once (correct_strings, modules/correction_module.py:12-75, loops over the
 * partitions -- a Pool task each; later steps of a run have hundreds to thousands of small partitions).  Rows of all partitions are
 * concatenated: partition p = rows first_row[p] .. first_row[p + 1] (its first row is the centre, with no ops), row_ids / ops_ptr / degree
 * index the concatenation, the slot arrays (out_col_slot, out_longest) are concatenated too (len(centre) + 1 entries per partition, in
 * partition order).  Wide records as in isocon_msa_build_ops with the row as an index into the concatenation and the partition in word 6;
 * patches address (row of the concatenation, column of that row's matrix).  out_n_cand[r] = -1 for a row with more correctable positions
 * than the kernel keeps in LDS: its partition has to be corrected through the single-partition entry points (its output rows are not valid).
 */
int isocon_msa_build_ops_batch(isocon_store *s, uint32_t n_parts, const uint32_t *first_row, const uint32_t *row_ids, const uint32_t *ops,
                               const uint64_t *ops_ptr, uint32_t *out_n_cols, uint32_t *out_col_slot, uint32_t *out_longest, uint32_t *out_wide,
                               uint64_t wide_cap, uint64_t *n_wide, float *kernel_ms);
int isocon_msa_correct_built_batch(isocon_store *s, uint32_t n_parts, uint32_t n_rows, const uint32_t *patch_row, const uint32_t *patch_col,
                                   const uint32_t *patch_ptr, const uint8_t *patch_bytes, uint32_t n_patches, const int32_t *degree,
                                   uint8_t *out_packed, uint64_t packed_cap, uint64_t *out_offsets, int32_t *out_n_cand, float *kernel_ms);

/*
 * Batched infix ("HW") edit distance with location and path ends == edlib.align(q, t, mode="HW", task="path", k=k)
 * as consumed by edlib_traceback (modules/end_invariant_functions.py:593-620) inside get_all_NN (:622-681), the
 * candidate-vs-candidate graph of the statistical-test phase: the query is aligned globally inside the target, target
 * prefix and suffix are free.  k[p] >= 0 is required (k = 10 + ignore_ends_len there); the diagonals a path of cost <= k
 * can visit must fit 512 (max(len(t) - len(q), 0) + 2 k + 1 <= 512), otherwise ISOCON_E_UNSUPPORTED.
 * out[5 p ..] = editDistance (-1 if > k[p]; then the rest is -1 / 0), locations[0] start, end (0-based, inclusive: the
 * first optimal end, the smallest start for it), length of the insertion run (query bases without target) the path
 * starts with, length of the one it ends with (0 = the CIGAR does not start / end with 'I').  Path = global alignment of
 * q to t[start..end], traced from the end preferring I, then D, then the diagonal (oracle/isocon_oracle.c section 5).
 */
int isocon_hw_pairs(isocon_store *s, const uint32_t *q, const uint32_t *t, const int32_t *k, uint64_t n_pairs,
                    int32_t *out, float *kernel_ms);

/*
 * Greedy partition of the nearest-neighbour graph into consensus centres and their members, on integer ids: what
 * get_partitions_no_copy (modules/partitions.py:301-413, called by partition_strings :416-593 on nx.reverse(G_star)) and
 * partition_highest_reachable_with_edge_degrees (modules/end_invariant_functions.py:405-533; nbr_tiebreak = 0) compute on networkx
 * graphs keyed by the sequences.  Host-only (no GPU work): the graph is the edge list the NN search returns.
 *   n nodes, degree[i] = multiplicity of sequence i (graphs.py:37-51), edges (edge_a[e] -> edge_b[e]): b is a nearest neighbour of a
 *   (an edge of G*), rank[i] = position of sequence i in the lexicographic order of the sequences (the reference breaks ties with
 *   `m < centre` on the strings), nbr_tiebreak != 0: between start nodes of equal reachable weight prefer more direct in-neighbours.
 * Output, in the reference's extraction order (components by size, largest first): out_centre[p], out_weight[p] = total multiplicity
 * of partition p, its other members out_members[out_member_ptr[p] .. out_member_ptr[p + 1]) (any order: the reference returns sets).
 * Capacities: out_centre / out_weight n entries, out_member_ptr n + 1, out_members n.  Deterministic where the reference depends on
 * PYTHONHASHSEED (ties between different reachable sets of equal weight: SURVEY.md F6).
 */
int isocon_partition_ids(uint32_t n, const int32_t *degree, uint64_t n_edges, const uint32_t *edge_a, const uint32_t *edge_b,
                         const uint32_t *rank, int32_t nbr_tiebreak, uint32_t *out_centre, int64_t *out_weight,
                         uint64_t *out_member_ptr, uint32_t *out_members, uint32_t *n_parts);

#ifdef __cplusplus
}
#endif
#endif
