"""The candidate-inference phase on ONT-profile reads (wide-band NN search, long gaps): robustness / timing check."""
import os, sys, tempfile, time, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isocon_amd import synth
from isocon_amd import isocon_get_candidates as IGC
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
accs, seqs, isoforms = synth.make_reads(n, 0, 10, 50001, profile=dict(synth.ONT_PROFILE, rate=float(sys.argv[2]) if len(sys.argv) > 2 else 0.06), families=2, length_range=(1000, 2500))
with tempfile.TemporaryDirectory() as tmp:
    rf = os.path.join(tmp, "reads.fa")
    with open(rf, "w") as fh:
        for a, s in zip(accs, seqs): fh.write(">%s\n%s\n" % (a, s))
    class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; logfile = None; min_exon_diff = 20
    P.ignore_ends_len = 15; P.min_candidate_support = 2; P.is_fastq = False; P.ccs = None; P.outfolder = tmp
    t = time.time(); cand_file, rp, to_realign = IGC.find_candidate_transcripts(rf, P); dt = time.time() - t
    cands = [l.strip() for l in open(cand_file) if not l.startswith(">")]
    steps = 1 + len(glob.glob(os.path.join(tmp, "candidates_step_*.fa")))
    print("ONT profile: %.1f s, %d reads, %d steps -> %d candidates (%d are true isoforms of %d), %d reads assigned, %d to realign" %
          (dt, n, steps, len(cands), len(set(cands) & set(isoforms)), len(isoforms), sum(len(v) for v in rp.values()), len(to_realign)))
