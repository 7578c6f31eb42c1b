import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
for i in range(3):
    best, rp, cols, stats = st.nn_graph()
print(os.environ.get("ISOCON_LIB", "default"), "bounds %.2f ms" % stats["bound_kernel_ms"])
