"""GPU box: lane statistics of the table kernel at C3 (executed vs live lane-columns, pairs per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
for i in range(2): best, rp, cols, s = st.nn_graph()
print({k: s[k] for k in ("pairs_evaluated", "pairs_lanes", "cells_columns", "live_columns", "tiles", "scan_kernel_ms", "lanes_kernel_ms")})
print("table kernel: executed wave-columns %.3e, live lane fraction %.3f, pairs %.3e, columns per pair %.0f" % (
    s["cells_columns"] / 64, s["live_columns"] / s["cells_columns"], s["pairs_evaluated"] - s["pairs_lanes"], s["live_columns"] / (s["pairs_evaluated"] - s["pairs_lanes"])))
