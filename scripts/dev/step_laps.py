"""Host lap times (ISOCON_DEBUG) of one steady-state isocon_nn_graph call at C3, next to its kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
os.environ.pop("ISOCON_DEBUG", None)
for _ in range(10):
    st.nn_graph()
ts = []
for _ in range(20):
    t0 = time.perf_counter(); out = st.nn_graph(); ts.append(1e3 * (time.perf_counter() - t0))
s = out[3]
print("steady state: wall min %.2f median %.2f ms; kernels %.2f ms (bounds %.2f seeds %.2f lists %.2f tables %.2f lanes %.2f)" % (
    min(ts), sorted(ts)[10], s["kernel_ms"], s["bound_kernel_ms"], s["seed_kernel_ms"], s["list_kernel_ms"], s["scan_kernel_ms"], s["lanes_kernel_ms"]))
os.environ["ISOCON_DEBUG"] = "1"
st.nn_graph()
