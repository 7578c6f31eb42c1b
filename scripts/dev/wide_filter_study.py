"""Would the greedy block bound (csrc/nn_filter.hpp) prune the WIDE-band phase of a C5-shaped set?  Sample window pairs (x < y in the
length order, |len difference| <= k, k = min(511, max of the two ends' final nearest-neighbour distances): the pairs phase B aligns once
the thresholds are final), their exact distances (k = none) and the block counts in both directions and at strides 4 / 2;
report how many of the pairs with d > k each count proves away, by d / L class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_sample = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
accs, seqs, _ = synth.make_reads(n_reads, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
n = st.n
lens = np.asarray(st.lens[:n], dtype=np.int64)
best = st.nn_graph()[0].astype(np.int64)
best = np.where(best < 0, 511, best)
rng = np.random.default_rng(5)
x = rng.integers(0, n - 1, size=4 * n_sample)
# a partner above x inside the length window of 511
hi = np.searchsorted(lens, lens[x] + 511, side="right")
ok = hi > x + 1
x = x[ok]; hi = hi[ok]
y = x + 1 + (rng.random(len(x)) * (hi - x - 1)).astype(np.int64)
k = np.minimum(511, np.maximum(best[x], best[y]))
keep = (lens[y] - lens[x]) <= k
x, y, k = x[keep][:n_sample], y[keep][:n_sample], k[keep][:n_sample]
d = st.ed_pairs(x, y, None).astype(np.int64)
L = np.maximum(lens[x], lens[y])
rej = d > k
print("sampled window pairs: %d, of them beyond their threshold: %d (%.3f); median threshold / L: %.3f" % (len(x), rej.sum(), rej.mean(), np.median(k / L)))
for s in (4, 2):
    up = st.block_bound_pairs(x, y, probe_stride=s).astype(np.int64)       # grams of the longer one missing in the shorter one
    dn = st.block_bound_pairs(y, x, probe_stride=s).astype(np.int64)
    assert (up <= d).all() and (dn <= d).all()
    both = np.maximum(up, dn)
    print("stride %d: bound / distance (median): up %.3f  down %.3f  max %.3f" % (s, np.median(up / np.maximum(d, 1)), np.median(dn / np.maximum(d, 1)), np.median(both / np.maximum(d, 1))))
    for name, b in (("up", up), ("down", dn), ("max", both)):
        print("   %-4s rejects %.3f of the pairs beyond their threshold" % (name, (b[rej] > k[rej]).mean()))
    r = d / L
    for lo_, hi_ in ((0, 0.15), (0.15, 0.25), (0.25, 0.4), (0.4, 2)):
        m = rej & (r >= lo_) & (r < hi_)
        if m.sum():
            print("   d/L in [%.2f, %.2f): %6d pairs beyond threshold, up rejects %.3f, max rejects %.3f" % (lo_, hi_, m.sum(), (up[m] > k[m]).mean(), (both[m] > k[m]).mean()))
