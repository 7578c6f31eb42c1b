"""Scaled-down config C5 (mixed-length ONT-profile reads): exercises the wide-band / un-banded fallbacks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
accs, seqs, _ = synth.make_reads(n, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
seqs = sorted(dict.fromkeys(seqs), key=len)
lens = np.array([len(s) for s in seqs])
print("n", len(seqs), "len range", lens.min(), lens.max())
st = SeqStore(seqs)
t = time.time(); best, rp, cols, stats = st.nn_graph(); dt = time.time() - t
print("nn_graph wall %.2f s" % dt, stats)
print("best: median %.0f  max %d  rows empty %d" % (np.median(best[best >= 0]), best.max(), (best < 0).sum()))
print("checksum best %d edges %d cols-sum %d" % (int(best.astype(np.int64).sum()), len(cols), int(cols.astype(np.int64).sum())))
