"""How many of the main pass' pairs could run in a 32-row band (threshold <= 31)?  C3, final bounds (the most favourable case:
during the pass the bounds are only looser).  A pair (i, j) is evaluated with k = min(63, max(b_i, b_j)) if |len_i - len_j| <= k."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
best, row_ptr, cols, stats = st.nn_graph()
lens = st.lens
b = np.minimum(best, 63).astype(np.int64)
rng = np.random.default_rng(1)
tot = le31 = 0
hist = np.zeros(64, dtype=np.int64)
for i in rng.choice(len(seqs), 3000, replace=False):
    hi = np.searchsorted(lens, lens[i] + 63, "right")
    j = np.arange(i + 1, hi)
    k = np.maximum(b[i], b[j])
    ok = (lens[j] - lens[i]) <= k
    tot += int(ok.sum()); le31 += int((ok & (k <= 31)).sum())
    hist += np.bincount(k[ok], minlength=64)
print("pairs (sampled rows): %d, with k <= 31: %d = %.1f %%" % (tot, le31, 100.0 * le31 / tot))
c = np.cumsum(hist) / hist.sum()
print("cumulative share of pairs by k:", {k: round(float(c[k]), 3) for k in (23, 27, 31, 35, 39, 43, 47, 55, 63)})
print("NN distance quartiles:", np.percentile(best[best >= 0], [5, 25, 50, 75, 95]))
