import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd import isocon_get_candidates as IGC
accs, seqs, isoforms = synth.make_reads(50000, 2500, 10, 30001)
with tempfile.TemporaryDirectory() as tmp:
    rf = os.path.join(tmp, "reads.fa")
    with open(rf, "w") as fh:
        for a, s in zip(accs, seqs): fh.write(">%s\n%s\n" % (a, s))
    class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; logfile = None; min_exon_diff = 20
    P.ignore_ends_len = 15; P.min_candidate_support = 2; P.is_fastq = False; P.ccs = None; P.outfolder = tmp
    IGC.find_candidate_transcripts(rf, P)
    pr = cProfile.Profile(); pr.enable(); IGC.find_candidate_transcripts(rf, P); pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(40)
