"""Where phase 0 (bounds + seeds) of one rank's shard goes, per world size: stats fields of isocon_nn_partial."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth, _lib
from isocon_amd.dist import shard_of
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
n = st.n
st.nn_graph()
for world in (1, 2, 4, 8):
    for rep in range(2):
        b = np.full(n, _lib.NN_INF, dtype=np.int32)
        qb, qe, qs, qk = shard_of(0, world, n)
        hits, s = st.nn_partial(qb, qe, 0, b, q_stride=qs, q_block=qk)
    print("world %d rank 0 phase 0: kernel %.2f ms = bounds %.2f (tiles %d) + seeds %.2f + other %.2f; hits %d" % (
        world, s["kernel_ms"], s["bound_kernel_ms"], s["bound_tiles"], s["seed_kernel_ms"], s["kernel_ms"] - s["bound_kernel_ms"] - s["seed_kernel_ms"], len(hits)))
