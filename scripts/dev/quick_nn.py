import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
for (n, L, iso, seed) in ((5000, 1500, 3, 20001), (50000, 2500, 10, 30001)):
    t = time.time(); accs, seqs, _ = synth.make_reads(n, L, iso, seed); print("gen", time.time() - t)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    t = time.time(); st = SeqStore(seqs); print("store", time.time() - t, st.device_bytes() / 1e6, "MB", len(seqs))
    for rep in range(2):
        t = time.time(); best, rp, cols, stats = st.nn_graph(); dt = time.time() - t
        print("nn", dt, stats)
    lens = np.array([len(s) for s in seqs])
    b = np.where(best < 0, 0, best)
    lo = np.searchsorted(lens, lens - b, 'left'); hi = np.searchsorted(lens, lens + b, 'right')
    nwin = int((hi - lo - 1).sum())
    print("window pairs", nwin, "rate", nwin / dt / 1e6, "M/s", "median best", np.median(best), "edges", len(cols))
