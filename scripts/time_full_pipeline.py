"""Wall time of the whole pipeline (find_candidate_transcripts + stat_filter_candidates) on synthetic CCS reads, with a
breakdown of the statistical-test phase.  Usage: python scripts/time_full_pipeline.py [n_reads] [length] [isoforms] [ont|ccs] [fastq]"""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isocon_amd import synth
from isocon_amd import isocon_get_candidates as IGC
from isocon_amd import isocon_statistical_test as IST
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
iso = int(sys.argv[3]) if len(sys.argv) > 3 else 10
if len(sys.argv) > 4 and sys.argv[4] == "ont":      # ONT error profile (6 %), two gene families of mixed length
    accs, seqs, isoforms = synth.make_reads(n, 0, iso, 50001, profile=synth.ONT_PROFILE, families=2, length_range=(1000, L))
else:
    accs, seqs, isoforms = synth.make_reads(n, L, iso, 30001)
with tempfile.TemporaryDirectory() as tmp:
    fastq = "fastq" in sys.argv[4:]           # FASTQ input with random base qualities: the tests then use them
    rf = os.path.join(tmp, "reads.fq" if fastq else "reads.fa")
    with open(rf, "w") as fh:
        if fastq:
            import numpy as np
            rq = np.random.default_rng(1)
            for a, s in zip(accs, seqs): fh.write("@%s\n%s\n+\n%s\n" % (a, s, (rq.integers(5, 60, len(s)) + 33).astype(np.uint8).tobytes().decode()))
        else:
            for a, s in zip(accs, seqs): fh.write(">%s\n%s\n" % (a, s))
    class Out:
        def write(self, x): sys.stdout.write(x); sys.stdout.flush()
    class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; logfile = Out(); min_exon_diff = 20
    P.ignore_ends_len = 15; P.min_candidate_support = 2; P.is_fastq = fastq; P.ccs = None; P.outfolder = tmp
    P.p_value_threshold = 0.01; P.min_test_ratio = 5; P.max_phred_q_trusted = 43
    t = time.time(); cand_file, rp, to_realign = IGC.find_candidate_transcripts(rf, P); t1 = time.time() - t
    ncand = sum(1 for l in open(cand_file) if l.startswith(">"))
    print("find_candidate_transcripts: %.1f s -> %d candidates, %d reads assigned, %d to realign" % (t1, ncand, sum(len(v) for v in rp.values()), len(to_realign)), flush=True)
    import numpy as np
    from isocon_amd import end_invariant_functions as END
    cl = np.sort(np.array([len(l.strip()) for l in open(cand_file) if not l.startswith(">")], dtype=np.int64))
    print("candidate-vs-candidate infix alignments (window 10 + 2 x 15): %d pairs" % len(END._window_pairs(cl, 0, len(cl), 40, 2 ** 32)[0]), flush=True)
    pr = cProfile.Profile(); t = time.time(); pr.enable()
    C = IST.stat_filter_candidates(rf, cand_file, rp, to_realign, P)
    pr.disable(); t2 = time.time() - t
    rounds = len([f for f in os.listdir(tmp) if f.startswith("p_values_")])
    print("stat_filter_candidates: %.1f s, %d rounds -> %d final candidates (%d are true isoforms of %d)" % (t2, rounds, len(C), len(set(C.values()) & set(isoforms)), len(isoforms)), flush=True)
    print("whole pipeline: %.1f s for %d reads x ~%d bp" % (t1 + t2, n, L))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
