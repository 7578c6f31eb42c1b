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
os.environ["ISOCON_DEBUG"] = "1"
for rep in range(2):
    best = np.full(n, _lib.NN_INF, dtype=np.int32)
    for phase in (0, 1):
        t0 = time.perf_counter()
        hits, s = st.nn_partial(0, n, phase, best, q_stride=8)
        sys.stderr.write("== rep %d phase %d wall %.2f ms kernels %.2f hits %d\n" % (rep, phase, (time.perf_counter() - t0) * 1e3, s["kernel_ms"], len(hits)))
