"""C3 main pass with the q-gram bounds: waves per workgroup, and what final thresholds from the start would give."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth, _lib
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
n = st.n
def show(tag, stats, wall):
    print("%-34s wall %6.1f ms  main %6.1f  seed %5.1f  bounds %5.1f | pairs %.3e prefiltered %.3e | wave-cols %.3e live %.3f"
          % (tag, wall, stats["scan_kernel_ms"], stats["seed_kernel_ms"], stats["bound_kernel_ms"], stats["pairs_evaluated"], stats["pairs_prefiltered"],
             stats["cells_columns"] / 64.0, stats["live_columns"] / max(stats["cells_columns"], 1)), flush=True)
for waves in sys.argv[1:] or ("8", "4", "2"):
    os.environ["ISOCON_NN_WAVES"] = waves
    for rep in range(2):
        t0 = time.perf_counter(); best, rp, cols, stats = st.nn_graph(); wall = (time.perf_counter() - t0) * 1e3
    show("waves=%s whole graph" % waves, stats, wall)
    b = best.copy().astype(np.int32); b[b < 0] = _lib.NN_INF
    t0 = time.perf_counter(); hits, stats = st.nn_partial(0, n, 1, b); wall = (time.perf_counter() - t0) * 1e3
    show("waves=%s main pass, final bounds" % waves, stats, wall)
