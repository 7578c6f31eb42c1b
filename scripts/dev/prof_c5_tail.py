"""cProfile of find_candidate_transcripts on C5-shaped reads (default 50 000): where the time after the last correction step goes."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd import isocon_get_candidates as IGC
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
accs, seqs, isoforms = synth.make_reads(n, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
with tempfile.TemporaryDirectory() as tmp:
    rf = os.path.join(tmp, "reads.fa")
    with open(rf, "w") as fh:
        for a, s in zip(accs, seqs): fh.write(">%s\n%s\n" % (a, s))
    class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; logfile = None; min_exon_diff = 20
    P.ignore_ends_len = 15; P.min_candidate_support = 2; P.is_fastq = False; P.ccs = None; P.outfolder = tmp
    pr = cProfile.Profile(); pr.enable(); t = time.time(); IGC.find_candidate_transcripts(rf, P); dt = time.time() - t; pr.disable()
    print("find_candidate_transcripts %.1f s" % dt)
    pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
