"""cProfile of find_candidate_transcripts at C3 (where the host time of the candidate-inference phase goes)."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth, isocon_get_candidates as IGC
n, L = 50000, 2500
accs, seqs, isoforms = synth.make_reads(n, L, 10, 30001)
tmp = tempfile.mkdtemp()
rf = os.path.join(tmp, "reads.fa")
with open(rf, "w") as fh:
    for a, s in zip(accs, seqs): fh.write(">%s\n%s\n" % (a, s))
class Out:
    def write(self, x): pass
class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; logfile = Out(); min_exon_diff = 20
P.ignore_ends_len = 15; P.min_candidate_support = 2; P.is_fastq = False; P.ccs = None; P.outfolder = tmp
pr = cProfile.Profile(); t = time.time(); pr.enable()
cand_file, rp, to_realign = IGC.find_candidate_transcripts(rf, P)
pr.disable()
print("find_candidate_transcripts: %.1f s" % (time.time() - t))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
