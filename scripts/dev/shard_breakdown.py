"""Per-kernel times of ONE rank's phases in an emulated 8-rank search at C3 (bounds reduced over all ranks between the phases)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth, _lib
from isocon_amd.store import SeqStore
from isocon_amd.dist import shard_of
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
n = st.n
keys = ["kernel_ms", "bound_kernel_ms", "seed_kernel_ms", "list_kernel_ms", "scan_kernel_ms", "narrow_kernel_ms", "lanes_kernel_ms", "pairs_evaluated", "pairs_lanes"]
for rep in range(2):
    red = np.full(n, _lib.NN_INF, dtype=np.int32)
    for phase in (0, 1):
        bests, rows = [], []
        for r in range(world):
            b, e, s, k = shard_of(r, world, n)
            best = red.copy()
            h, stats = st.nn_partial(b, e, phase, best, q_stride=s, q_block=k)
            bests.append(best); rows.append(stats)
        red = np.minimum.reduce(bests)
        if rep == 1:
            for r in (0, world - 1):
                print("phase %d rank %d: " % (phase, r) + ", ".join("%s %.2f" % (k2.replace("_kernel_ms", "").replace("_ms", ""), rows[r][k2]) if "ms" in k2 else "%s %d" % (k2, rows[r][k2]) for k2 in keys), flush=True)
