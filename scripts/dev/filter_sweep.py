"""GPU box: the C3 step under variants of the block filter (ISOCON_DEBUG_VARIANT is read per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
ref = None
variants = sys.argv[1:] or ["", "nn_filter_one_pass", "nn_table_chunks=0", "nn_no_block_filter"]
for v in variants:
    if v: os.environ["ISOCON_DEBUG_VARIANT"] = v
    else: os.environ.pop("ISOCON_DEBUG_VARIANT", None)
    ts = []
    for i in range(6):
        t0 = time.perf_counter(); best, rp, cols, s = st.nn_graph(); ts.append(time.perf_counter() - t0)
    key = (best.tobytes(), rp.tobytes(), cols.tobytes())
    if ref is None: ref = key
    print("%-42s wall %6.2f ms kernels %6.2f | bounds %.2f seeds %.2f lists %.2f (filter %.2f) tables %.2f (narrow %.2f) lanes %.2f | aligned %d (lanes %d) rejected %d same graph %s" % (
        v or "(default)", 1e3 * min(ts), s["kernel_ms"], s["bound_kernel_ms"], s["seed_kernel_ms"], s["list_kernel_ms"], s["filter_kernel_ms"], s["scan_kernel_ms"], s["narrow_kernel_ms"],
        s["lanes_kernel_ms"], s["pairs_evaluated"], s["pairs_lanes"], s["pairs_block_rejected"], key == ref), flush=True)
