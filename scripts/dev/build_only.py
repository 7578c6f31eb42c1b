import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
st.nn_graph()
for waves in ("4", "8"):
    os.environ["ISOCON_NN_WAVES"] = waves
    os.environ["ISOCON_NN_BUILD_ONLY"] = "1"
    for rep in range(2):
        best, rp, cols, stats = st.nn_graph()
    print("waves", waves, "table build only: main %.2f ms" % stats["scan_kernel_ms"], flush=True)
    del os.environ["ISOCON_NN_BUILD_ONLY"]
