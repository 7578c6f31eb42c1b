import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth, _lib
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
n = st.n
bestf, _, _, s = st.nn_graph()
print("single main %.2f" % s["scan_kernel_ms"])
bf = bestf.copy().astype(np.int32); bf[bf < 0] = _lib.NN_INF
for world in (8, 2, 1):
    for rep in range(4):
        b = bf.copy()
        t0 = time.perf_counter()
        hits, s = st.nn_partial(0, n, 1, b, q_stride=world)
        print("world %d rep %d (final bounds in): wall %.2f kernel %.2f main %.2f bounds %.2f pairs %.3e cols %.3e live %.3f" % (world, rep, (time.perf_counter() - t0) * 1e3, s["kernel_ms"], s["scan_kernel_ms"], s["bound_kernel_ms"], s["pairs_evaluated"], s["cells_columns"] / 64, s["live_columns"] / max(1, s["cells_columns"])), flush=True)
os.environ["ISOCON_NN_ORDER"] = "0"
for world in (8, 1):
    for rep in range(3):
        b = bf.copy()
        hits, s = st.nn_partial(0, n, 1, b, q_stride=world)
        print("NO_ORDER world %d rep %d: kernel %.2f main %.2f bounds %.2f" % (world, rep, s["kernel_ms"], s["scan_kernel_ms"], s["bound_kernel_ms"]), flush=True)
