"""Wall time of the front half of one IsoCon correction iteration at C3 scale (what isocon_get_candidates.py:127-130
times as 'nearest_neighbors and partition'): partition_strings + get_partition_alignments."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isocon_amd import synth, partitions
from isocon_amd import isocon_get_candidates as IGC

class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; min_exon_diff = 20; ignore_ends_len = 15
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
accs, seqs, _ = synth.make_reads(n, 2500, 10, 30001)
S = dict(zip(accs, seqs))
for rep in range(2):
    t0 = time.time(); G, part, M, conv = partitions.partition_strings(S, P()); t1 = time.time()
    ex = set(); pa = IGC.get_partition_alignments(part, M, G, ex, P()); t2 = time.time()
    print("partition_strings %.2f s (%d centres), get_partition_alignments %.2f s (%d alignments, %d exon-filtered)" %
          (t1 - t0, len(part), t2 - t1, sum(len(v) - 1 for v in pa.values()), len(ex)))
if len(sys.argv) > 2:
    pr = cProfile.Profile(); pr.enable()
    G, part, M, conv = partitions.partition_strings(S, P()); ex = set(); pa = IGC.get_partition_alignments(part, M, G, ex, P())
    pr.disable(); pstats.Stats(pr).sort_stats("cumtime").print_stats(25)
from isocon_amd import correction_module as COR
seq_to_acc = IGC.get_unique_seq_accessions(S)
t0 = time.time(); S_prime, _ = COR.correct_strings(pa, seq_to_acc, {}, 1); t1 = time.time()
changed = sum(1 for a, s in S_prime.items() if S[a] != s)
print("correct_strings %.2f s (%d accessions returned, %d changed)" % (t1 - t0, len(S_prime), changed))
if len(sys.argv) > 3:
    pr = cProfile.Profile(); pr.enable(); COR.correct_strings(pa, seq_to_acc, {}, 1); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
