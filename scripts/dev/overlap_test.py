import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
for tag, env in (("sequential", {}), ("main + second bound kernel concurrently", {"ISOCON_DBG_OVERLAP": "1"})):
    os.environ.update(env)
    for rep in range(3):
        best, rp, cols, s = st.nn_graph()
    print("%-44s bounds %.2f main(+overlapped) %.2f total kernels %.2f" % (tag, s["bound_kernel_ms"], s["scan_kernel_ms"], s["kernel_ms"]), flush=True)
