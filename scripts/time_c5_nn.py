"""Exact 1-set NN graph of a C5-shaped set (ONT error profile 6 %, 1-5 kb, 50 isoforms in 5 families, seed 50001) on one GPU:
wall time, kernel statistics and a checksum of the graph (to compare builds).  Usage: python scripts/time_c5_nn.py [n_reads]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
t0 = time.time()
accs, seqs, iso = synth.make_reads(n, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
seqs = sorted(dict.fromkeys(seqs), key=len)
print("generated %d reads (%d unique) in %.0f s" % (n, len(seqs), time.time() - t0), flush=True)
st = SeqStore(seqs)
t0 = time.perf_counter()
best, row_ptr, cols, stats = st.nn_graph()
dt = time.perf_counter() - t0
h = hashlib.blake2b(digest_size=8)
h.update(np.ascontiguousarray(best).tobytes()); h.update(np.ascontiguousarray(row_ptr).tobytes()); h.update(np.ascontiguousarray(cols).tobytes())
print("C5-shape x %d: NN graph %.2f s, kernels %.0f ms, pairs %.3g, wave-columns %.3g, edges %d, median NN distance %.0f, checksum %s"
      % (len(seqs), dt, stats["kernel_ms"], stats["pairs_evaluated"], stats["cells_columns"] / 64, len(cols), np.median(best[best >= 0]), h.hexdigest()), flush=True)
print(stats)
