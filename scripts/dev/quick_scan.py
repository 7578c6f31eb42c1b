"""Prints the main-pass kernel time of the 1-set NN search at C3 (kernel experiments; set ISOCON_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
n, L, iso, seed = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (50000, 2500, 10, 30001)))
accs, seqs, _ = synth.make_reads(n, L, iso, seed)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
ms = []
for rep in range(3):
    t = time.time(); best, rp, cols, stats = st.nn_graph(); dt = time.time() - t
    ms.append((stats["scan_kernel_ms"], stats["seed_kernel_ms"], dt * 1e3))
print(os.environ.get("ISOCON_LIB", "default"), "scan/seed/wall ms:", ["%.1f/%.1f/%.1f" % m for m in ms],
      "lane-cols %.3g live %.3g pairs %.4g edges %d checksum %d" % (stats["cells_columns"], stats["live_columns"], stats["pairs_evaluated"], len(cols), int(best.astype(np.int64).sum())))
