import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth, _lib
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
n = st.n
st.nn_graph()
for world in (8, 4):
    best = np.full(n, _lib.NN_INF, dtype=np.int32)
    for phase in (0, 1):
        bests = []
        for r in range(world):
            b = best.copy()
            hits, s = st.nn_partial(r, n, phase, b, q_stride=world)
            bests.append(b)
            if r < 3:
                print("world %d phase %d rank %d: kernel %.2f main %.2f seed %.2f bounds %.2f pairs %.3e live %.3f" % (world, phase, r, s["kernel_ms"], s["scan_kernel_ms"], s["seed_kernel_ms"], s["bound_kernel_ms"], s["pairs_evaluated"], s["live_columns"] / max(1, s["cells_columns"])), flush=True)
        best = np.minimum.reduce(bests)
