"""Histogram of the q-gram bound bytes over every window pair of the C3 set (isocon_qgram_bound_matrix): how many pairs would a
threshold-free prefilter (bound <= 63) drop before the survivor scan, and how are the rest spread over 16-byte chunks of a row?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
accs, seqs, _ = synth.make_reads(n, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
row_ptr, vals = st.qgram_bound_matrix()
print("window pairs (with row padding): %d" % len(vals))
h = np.bincount(vals, minlength=256)
c = np.cumsum(h) / len(vals)
for t in (8, 16, 24, 32, 40, 48, 56, 63, 80, 100, 150, 200, 254):
    print("bound <= %3d: %.4f" % (t, c[t]))
# 16-byte chunks with no byte <= 63
m = len(vals) // 16 * 16
any_small = (vals[:m].reshape(-1, 16) <= 63).any(axis=1)
print("16-byte chunks holding a bound <= 63: %.4f" % any_small.mean())
any_small64 = (vals[:m // 64 * 64].reshape(-1, 64) <= 63).any(axis=1)
print("64-byte batches holding a bound <= 63: %.4f" % any_small64.mean())
best = st.nn_graph()[0]
print("median / 90 %% / max final best: %d / %d / %d" % (np.median(best[best >= 0]), np.percentile(best[best >= 0], 90), best.max()))
