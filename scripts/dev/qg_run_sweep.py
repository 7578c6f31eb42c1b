import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
for run in sys.argv[1:]:
    os.environ["ISOCON_QG_RUN"] = run
    for rep in range(2):
        best, rp, cols, stats = st.nn_graph()
    print("run", run, "bounds %.2f ms main %.2f seed %.2f total %.2f" % (stats["bound_kernel_ms"], stats["scan_kernel_ms"], stats["seed_kernel_ms"], stats["kernel_ms"]), flush=True)
