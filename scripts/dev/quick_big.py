"""Scaling checks on one GPU: 200 k CCS reads (16 x the pairs of C3) and 50 k ONT-profile reads (quarter-size C5)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
which = sys.argv[1] if len(sys.argv) > 1 else "ccs"
if which == "ccs":
    accs, seqs, _ = synth.make_reads(200000, 2500, 10, 30001)
else:
    accs, seqs, _ = synth.make_reads(50000, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
seqs = sorted(dict.fromkeys(seqs), key=len)
t = time.time(); st = SeqStore(seqs); print("store %.2f s, %d sequences, %.0f MB on device" % (time.time() - t, len(seqs), st.device_bytes() / 1e6))
t = time.time(); best, rp, cols, stats = st.nn_graph(); dt = time.time() - t
lens = st.lens
b = np.where(best < 0, 0, best)
nwin = int((np.searchsorted(lens, lens + b, "right") - np.searchsorted(lens, lens - b, "left") - 1)[best >= 0].sum())
print("%s: nn_graph %.2f s, kernels %.2f s, pairs evaluated %.3g, window pairs %.3g (%.3g /s), median NN distance %.0f, edges %d, rows without neighbour %d" %
      (which, dt, stats["kernel_ms"] / 1e3, stats["pairs_evaluated"], nwin, nwin / dt, np.median(best[best >= 0]), len(cols), int((best < 0).sum())))
print("lane-cols %.3g live %.3g seed %.1f ms main %.1f ms" % (stats["cells_columns"], stats["live_columns"], stats["seed_kernel_ms"], stats["scan_kernel_ms"]))
