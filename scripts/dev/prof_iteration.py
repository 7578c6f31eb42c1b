"""One correction iteration of the candidate phase at C3 (partition_strings -> get_partition_alignments -> correct_strings): wall per
function over a few repetitions and a cProfile of the last two."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth, partitions, isocon_get_candidates as IGC, correction_module as CM
class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; min_exon_diff = 20; ignore_ends_len = 15
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
S = dict(zip(accs, seqs))
import collections
for rep in range(3):
    t0 = time.perf_counter(); Gs, part, M, conv = partitions.partition_strings(S, P()); t1 = time.perf_counter()
    pa = IGC.get_partition_alignments(part, M, Gs, set(), P()); t2 = time.perf_counter()
    s2a = IGC.get_unique_seq_accessions(S)
    t2b = time.perf_counter(); out = CM.correct_strings(pa, s2a, {}, 1); t3 = time.perf_counter()
    print("partition_strings %.1f ms, get_partition_alignments %.1f ms, correct_strings %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2b)), flush=True)
pr = cProfile.Profile(); pr.enable()
pa = IGC.get_partition_alignments(part, M, Gs, set(), P())
out = CM.correct_strings(pa, s2a, {}, 1)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(22)
