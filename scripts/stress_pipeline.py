"""Differential stress of the WHOLE pipeline: random small read sets through find_candidate_transcripts +
stat_filter_candidates once on the HIP kernels and once with the CPU oracle substituted for every kernel (the way the CPU
tests do it); the files written must be identical.  Usage: python scripts/stress_pipeline.py [seed] [n_cases] [hard]"""
import hashlib, os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import isocon_amd.SW_alignment_module as SWM
import isocon_amd.edlib_alignment_module as EAM
from isocon_amd import correction_module as COR
from isocon_amd import end_invariant_functions as END
from isocon_amd import graphs, synth
from isocon_amd import isocon_get_candidates as IGC
from isocon_amd import isocon_statistical_test as IST
from oracle import correction as OC
from oracle import oracle as O
from test_all_nn import hw_row


class OracleStore(object):
    def __init__(self, seqs): self.seqs = list(seqs)
    def hw_pairs(self, q, t, k, **_kw):
        return np.asarray([hw_row(O, self.seqs[a], self.seqs[b], int(kk)) for a, b, kk in zip(q, t, np.broadcast_to(k, np.shape(q)))], dtype=np.int32).reshape(-1, 5)


def oracle_align_pairs(pairs, mismatch, match_score=2, opening_penalty=2, gap_ext=0, ed_upper=None):
    return [O.parasail_alignment(a, b, 0, 0, match_score=match_score, mismatch_penalty=int(mm), opening_penalty=opening_penalty, gap_ext=gap_ext)[2] for (a, b), mm in zip(pairs, mismatch)]


GPU = dict(cor=COR._correct_on_device, nng=graphs.nearest_neighbor_graph, ed=IGC.edlib_align_sequences, sw=IGC.sw_align_sequences,
           eka=EAM.edlib_align_sequences_keeping_accession, ska=SWM.sw_align_sequences_keeping_accession, ap=SWM._align_pairs, st=END.SeqStore)


def use(oracle):
    COR._correct_on_device = OC.correct_rows if oracle else GPU["cor"]
    graphs.nearest_neighbor_graph = O if oracle else GPU["nng"]
    IGC.edlib_align_sequences = O.edlib_align_sequences if oracle else GPU["ed"]
    IGC.sw_align_sequences = O.sw_align_sequences if oracle else GPU["sw"]
    EAM.edlib_align_sequences_keeping_accession = IST.edlib_align_sequences_keeping_accession = O.edlib_align_sequences_keeping_accession if oracle else GPU["eka"]
    SWM.sw_align_sequences_keeping_accession = IST.sw_align_sequences_keeping_accession = O.sw_align_sequences_keeping_accession if oracle else GPU["ska"]
    SWM._align_pairs = oracle_align_pairs if oracle else GPU["ap"]
    END.SeqStore = OracleStore if oracle else GPU["st"]


def run(accs, seqs, ends):
    with tempfile.TemporaryDirectory() as tmp:
        rf = os.path.join(tmp, "reads.fa")
        open(rf, "w").write("".join(">%s\n%s\n" % x for x in zip(accs, seqs)))

        class P:
            nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; logfile = None; min_exon_diff = 20
            ignore_ends_len = ends; min_candidate_support = 2; p_value_threshold = 0.01; min_test_ratio = 5; max_phred_q_trusted = 43
            is_fastq = False; ccs = None; outfolder = tmp
        cand_file, rp, to_realign = IGC.find_candidate_transcripts(rf, P)
        try:
            IST.stat_filter_candidates(rf, cand_file, rp, to_realign, P)
        except SystemExit:
            pass
        h = hashlib.sha1()
        for f in sorted(os.listdir(tmp)):
            if f.endswith((".fa", ".tsv")) and f != "reads.fa":
                h.update(f.encode()); h.update(open(os.path.join(tmp, f), "rb").read())
        return h.hexdigest(), len([l for l in open(os.path.join(tmp, "final_candidates.fa")) if l.startswith(">")])


rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
t0 = time.time()
for case in range(ncases):
    if len(sys.argv) > 3 and sys.argv[3] == "hard":      # noisy, longer reads: wide bands, multi-block infix alignments
        n, L, iso = rng.randint(20, 120), rng.choice([400, 900, 1500]), rng.randint(1, 3)
        prof = dict(synth.ONT_PROFILE if rng.random() < 0.5 else synth.CCS_PROFILE, rate=rng.choice([0.03, 0.06, 0.08]))
    else:
        n, L, iso = rng.randint(20, 260), rng.choice([80, 150, 260, 400, 700]), rng.randint(1, 4)
        prof = dict(synth.CCS_PROFILE, rate=rng.choice([0.005, 0.01, 0.02, 0.04]))
    accs, seqs, _ = synth.make_reads(n, L, iso, seed=rng.randint(0, 10 ** 6), profile=prof)
    ends = rng.choice([0, 5, 15, 15])
    use(False); g = run(accs, seqs, ends)
    use(True); o = run(accs, seqs, ends)
    use(False)
    ok = g == o
    bad += not ok
    print("case %d: n=%d L=%d iso=%d rate=%.3f ends=%d -> %d final candidates %s (%.0f s)" % (case, n, L, iso, prof["rate"], ends, g[1], "ok" if ok else "MISMATCH (oracle: %d)" % o[1], time.time() - t0), flush=True)
print("stress_pipeline: %d cases, %d mismatches" % (ncases, bad))
sys.exit(1 if bad else 0)
