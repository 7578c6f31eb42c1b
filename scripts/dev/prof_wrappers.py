"""cProfile of the public wrappers at C3 (host share of compute_nearest_neighbor_graph / edlib_align_sequences / sw_align_sequences)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth, partitions
from isocon_amd import nearest_neighbor_graph as NNG, edlib_alignment_module as EAM, SW_alignment_module as SWM

class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None
accs, seqs, iso = synth.make_reads(50000, 2500, 10, 30001)
S = dict(zip(accs, seqs))
NNG.compute_nearest_neighbor_graph(S, set(), P())
G_star, partition, M, conv = partitions.partition_strings(S, P())
for name, fn in (("nn", lambda: NNG.compute_nearest_neighbor_graph(S, set(), P())),):
    pr = cProfile.Profile(); t = time.time(); pr.enable(); fn(); pr.disable(); print(name, "%.3f s" % (time.time() - t))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
ed = EAM.edlib_align_sequences(partition)
sw = SWM.sw_align_sequences(ed); del sw          # (the first call pins the result buffers)
pr = cProfile.Profile(); t = time.time(); pr.enable(); sw = SWM.sw_align_sequences(ed); pr.disable(); print("sw", "%.3f s" % (time.time() - t))
pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
