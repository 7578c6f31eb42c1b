import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
st.nn_graph()
os.environ["ISOCON_NN_NO_QGRAM"] = "1"
for waves in ("8", "4"):
    os.environ["ISOCON_NN_WAVES"] = waves
    for rep in range(2):
        best, rp, cols, s = st.nn_graph()
    wc = s["cells_columns"] / 64.0
    print("no bounds, waves %s: main %.2f ms seed %.2f wave-cols %.4e live %.3f -> %.1f ns per 1e3 wave-cols/SIMD" % (waves, s["scan_kernel_ms"], s["seed_kernel_ms"], wc, s["live_columns"] / s["cells_columns"], s["scan_kernel_ms"] * 1e6 * 1024 / wc), flush=True)
del os.environ["ISOCON_NN_NO_QGRAM"]
for waves in ("8", "4"):
    os.environ["ISOCON_NN_WAVES"] = waves
    for rep in range(2):
        best, rp, cols, s = st.nn_graph()
    wc = s["cells_columns"] / 64.0
    print("bounds, waves %s: main %.2f ms wave-cols %.4e live %.3f -> %.1f ns per 1e3 wave-cols/SIMD" % (waves, s["scan_kernel_ms"], wc, s["live_columns"] / s["cells_columns"], s["scan_kernel_ms"] * 1e6 * 1024 / wc), flush=True)
