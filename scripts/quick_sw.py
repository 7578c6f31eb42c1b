"""Times isocon_ed_pairs + isocon_sg_trace_batch on the (query, first NN) pairs of a synthetic set (kernel experiments)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
n, L, iso, seed = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (50000, 2500, 10, 30001)))
npairs = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
accs, seqs, _ = synth.make_reads(n, L, iso, seed)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
best, rp, cols, stats = st.nn_graph()
has = np.nonzero(rp[1:] > rp[:-1])[0][:npairs]
a = cols[rp[has]].astype(np.uint32)       # centre-like: the neighbour
b = has.astype(np.uint32)
t = time.time(); ed, ms = st.ed_pairs(a, b, None, return_ms=True); dt = time.time() - t
print("ed_pairs %d pairs: wall %.1f ms kernel %.2f ms  mean ed %.1f" % (len(a), dt * 1e3, ms, ed.mean()))
mm = np.where(ed / np.minimum(st.lens[a], st.lens[b]) <= 0.01, -1, np.where(ed / np.minimum(st.lens[a], st.lens[b]) <= 0.09, -2, -4)).astype(np.int8)
for rep in range(2):
    t = time.time(); ops, ptr, res, ms = st.sg_trace(a, b, mm, return_ms=True); dt = time.time() - t
    cells = float((st.lens[a] * st.lens[b]).sum())
    print("sg_trace %d pairs: wall %.1f ms kernel %.1f ms  %.3g cells/s (kernel)  %.1f pairs/ms  mean ops %.1f  trace bytes %.3g" % (
        len(a), dt * 1e3, ms, cells / (ms * 1e-3), len(a) / ms, len(ops) / len(a), cells / 2))
