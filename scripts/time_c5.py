"""Config C5 (mixed-length 1-5 kb ONT-error-profile reads, 50 isoforms in 5 gene families) through the whole
candidate-inference phase on one GPU.  Usage: python scripts/time_c5.py [n_reads]   (C5 proper: 200000)"""
import glob, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isocon_amd import synth
from isocon_amd import isocon_get_candidates as IGC
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t = time.time()
accs, seqs, isoforms = synth.make_reads(n, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
print("generated %d reads in %.0f s" % (n, time.time() - t), flush=True)


class Out:
    def write(self, x):
        sys.stdout.write("[%6.0f s] %s" % (time.time() - t0, x)); sys.stdout.flush()


with tempfile.TemporaryDirectory() as tmp:
    rf = os.path.join(tmp, "reads.fa")
    with open(rf, "w") as fh:
        for a, s in zip(accs, seqs):
            fh.write(">%s\n%s\n" % (a, s))

    class P:
        nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; logfile = Out(); min_exon_diff = 20
        ignore_ends_len = 15; min_candidate_support = 2; is_fastq = False; ccs = None; outfolder = tmp
    t0 = time.time()
    cand_file, rp, to_realign = IGC.find_candidate_transcripts(rf, P)
    dt = time.time() - t0
    cands = [l.strip() for l in open(cand_file) if not l.startswith(">")]
    steps = 1 + len(glob.glob(os.path.join(tmp, "candidates_step_*.fa")))
    print("C5 x %d reads: find_candidate_transcripts %.1f s, %d steps -> %d candidates (%d are true isoforms of %d), %d reads assigned, %d to realign" %
          (n, dt, steps, len(cands), len(set(cands) & set(isoforms)), len(isoforms), sum(len(v) for v in rp.values()), len(to_realign)), flush=True)
