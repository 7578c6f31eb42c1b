import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(1200, 1500, 3, seed=20001)
seqs = sorted(set(seqs), key=len)
st = SeqStore(seqs)
conv = np.zeros(len(seqs), np.uint8)
os.environ["ISOCON_NO_REGROUP"] = "1"
b0, r0, c0, s0 = st.nn_graph(is_converged=conv)
del os.environ["ISOCON_NO_REGROUP"]
b1, r1, c1, s1 = st.nn_graph(is_converged=conv)
bad = np.nonzero(b0 != b1)[0]
print("n", len(seqs), "differing best:", len(bad))
lens = st.lens
for i in bad[:10]:
    print(i, lens[i], "ref", b0[i], c0[r0[i]:r0[i+1]], "got", b1[i], c1[r1[i]:r1[i+1]], "lens nb", lens[c0[r0[i]:r0[i+1]]])
rows_differ = [i for i in range(len(seqs)) if c0[r0[i]:r0[i+1]].tolist() != c1[r1[i]:r1[i+1]].tolist()]
print("rows differ", len(rows_differ), rows_differ[:10])
print(s0); print(s1)
