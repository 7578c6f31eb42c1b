import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from isocon_amd import synth, partitions
from isocon_amd import isocon_get_candidates as IGC
class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None; min_exon_diff = 20; ignore_ends_len = 15
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
S = dict(zip(accs, seqs))
Gs, part, M, conv = partitions.partition_strings(S, P())
pa = IGC.get_partition_alignments(part, M, Gs, set(), P())
b = pa.batch; st = b.store
ed = st.ed_pairs(b.a, b.b, None)
la, lb = st.lens[b.a].astype(np.int64), st.lens[b.b].astype(np.int64)
rate = ed / np.minimum(la, lb)
mm = np.where(rate <= 0.01, -1, np.where(rate <= 0.09, -2, -4)).astype(np.int8)
st.sg_trace(b.a, b.b, mm, ed_upper=ed)
os.environ["ISOCON_DEBUG"] = "1"
for r in range(2):
    t0 = time.perf_counter(); st.sg_trace(b.a, b.b, mm, ed_upper=ed); print("sg_trace wall %.2f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
