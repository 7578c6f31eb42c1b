import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/isocon_amd") else ".")
from isocon_amd import synth
accs, seqs, _ = synth.make_reads(400, 600, 3, 5)
os.makedirs("/tmp/rs_out", exist_ok=True)
open("/tmp/rs_reads.fa", "w").write("".join(">%s\n%s\n" % x for x in zip(accs, seqs)))
from isocon_amd import isocon_get_candidates, isocon_statistical_test
class params:
    nr_cores = 1; neighbor_search_depth = 2 ** 32; min_exon_diff = 20; min_candidate_support = 2; ignore_ends_len = 15
    p_value_threshold = 0.01; min_test_ratio = 5; max_phred_q_trusted = 43; is_fastq = False; ccs = None
    verbose = False; logfile = None; develop_logfile = None; outfolder = "/tmp/rs_out"
cand_file, read_partition, to_realign = isocon_get_candidates.find_candidate_transcripts("/tmp/rs_reads.fa", params)
final = isocon_statistical_test.stat_filter_candidates("/tmp/rs_reads.fa", cand_file, read_partition, to_realign, params)
print(len(final), sorted(os.listdir("/tmp/rs_out"))[:6])
