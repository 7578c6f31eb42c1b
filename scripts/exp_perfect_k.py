import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isocon_amd import synth, _lib
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
best, rp, cols, stats = st.nn_graph()
print("normal   scan %.1f seed %.1f live %.3g" % (stats["scan_kernel_ms"], stats["seed_kernel_ms"], stats["live_columns"]))
b = np.where(best < 0, _lib.NN_INF, best).astype(np.int32)
hits, s2 = st.nn_partial(0, st.n, 0, b.copy())
print("perfect  scan %.1f seed %.1f live %.3g" % (s2["scan_kernel_ms"], s2["seed_kernel_ms"], s2["live_columns"]))
b2 = np.minimum(b + 4, _lib.NN_INF).astype(np.int32)
hits, s3 = st.nn_partial(0, st.n, 0, b2.copy())
print("best+4   scan %.1f seed %.1f live %.3g" % (s3["scan_kernel_ms"], s3["seed_kernel_ms"], s3["live_columns"]))
