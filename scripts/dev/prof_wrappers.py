"""cProfile of the two public wrappers on C3: compute_nearest_neighbor_graph and sw_align_sequences (strings in, dicts out)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth, partitions
from isocon_amd import nearest_neighbor_graph as NNG, edlib_alignment_module as EAM, SW_alignment_module as SWM
class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
S = dict(zip(accs, seqs))
for rep in range(3):
    t0 = time.perf_counter(); G, iso = NNG.compute_nearest_neighbor_graph(S, set(), P()); print("compute_nearest_neighbor_graph %.1f ms (kernels %.1f)" % (1e3 * (time.perf_counter() - t0), NNG.LAST_STATS["kernel_ms"]))
pr = cProfile.Profile(); pr.enable(); NNG.compute_nearest_neighbor_graph(S, set(), P()); pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(10)
Gs, part, M, conv = partitions.partition_strings(S, P())
ed = EAM.edlib_align_sequences(part)
for rep in range(3):
    t0 = time.perf_counter(); sw = SWM.sw_align_sequences(ed); print("sw_align_sequences %.1f ms" % (1e3 * (time.perf_counter() - t0))); del sw
pr = cProfile.Profile(); pr.enable(); sw = SWM.sw_align_sequences(ed); pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
from isocon_amd import isocon_get_candidates as IGC, correction_module as COR
class Q(P): min_exon_diff = 20; ignore_ends_len = 15
Gs, part, M, conv = partitions.partition_strings(S, Q())
for rep in range(3):
    t0 = time.perf_counter(); pa = IGC.get_partition_alignments(part, M, Gs, set(), Q()); print("get_partition_alignments %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); pa = IGC.get_partition_alignments(part, M, Gs, set(), Q()); pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
s2a = IGC.get_unique_seq_accessions(S)
for rep in range(2):
    t0 = time.perf_counter(); Sp, _ = COR.correct_strings(pa, s2a, {}, 1); print("correct_strings %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); Sp, _ = COR.correct_strings(pa, s2a, {}, 1); pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
for rep in range(2):
    t0 = time.perf_counter(); Gs, part, M, conv = partitions.partition_strings(S, Q()); print("partition_strings %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); partitions.partition_strings(S, Q()); pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
