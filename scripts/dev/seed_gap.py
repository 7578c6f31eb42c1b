"""How far are the thresholds after the seed stage from the final ones at C3, and how many survivors of the q-gram bound does the gap cost?
best0 = best[] after phase 0 (bounds + seeds), best = the final one.  With the bound matrix: survivors under either set of thresholds
(a pair (q, p) survives if bound <= max(thr(q), thr(p)), thr = min(best, 63); roles: every entry a query and a target)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth, _lib
from isocon_amd.store import SeqStore
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
accs, seqs, _ = synth.make_reads(n, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
n = st.n
final = st.nn_graph()[0].astype(np.int64)
best0 = np.full(n, _lib.NN_INF, dtype=np.int32)
st.nn_partial(0, n, 0, best0)
b0 = np.minimum(best0.astype(np.int64), 63)
bf = np.minimum(np.where(final < 0, 63, final), 63)
print("entries: %d; thresholds after the seeds: mean %.2f, final: mean %.2f; entries whose seed threshold is final: %.3f" % (n, b0.mean(), bf.mean(), (b0 == bf).mean()))
d = b0 - bf
for t in (0, 1, 2, 4, 8, 16):
    print("  seed threshold - final <= %2d: %.4f" % (t, (d <= t).mean()))
row_ptr, vals = st.qgram_bound_matrix()
row_ptr = row_ptr.astype(np.int64)
lens = np.diff(row_ptr)
# the storage of a row starts at column (q + 1) & ~15 and is padded: take the window's own bytes
tot0 = totf = 0
step = 1
for q in range(0, n, step):
    a, b = row_ptr[q], row_ptr[q + 1]
    if b <= a:
        continue
    pad = (q + 1) & 15
    row = vals[a + pad:b].astype(np.int64)
    m = min(len(row), n - q - 1)
    row = row[:m]
    p = np.arange(q + 1, q + 1 + m)
    k0 = np.maximum(b0[q], b0[p]); kf = np.maximum(bf[q], bf[p])
    dl = np.abs(np.asarray(st.lens)[p] - st.lens[q])
    tot0 += int(((row <= k0) & (dl <= k0)).sum()); totf += int(((row <= kf) & (dl <= kf)).sum())
print("survivors with the thresholds after the seeds: %d; with the final thresholds: %d (x %.2f)" % (tot0, totf, tot0 / max(totf, 1)))
