"""Step time of the C3 graph under a list of ISOCON_DEBUG_VARIANT settings (same graph asserted)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
ref = None
for v in [""] + sys.argv[1:] + [""]:
    if v: os.environ["ISOCON_DEBUG_VARIANT"] = v
    else: os.environ.pop("ISOCON_DEBUG_VARIANT", None)
    for _ in range(3): st.nn_graph()
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); out = st.nn_graph(); ts.append(1e3 * (time.perf_counter() - t0))
    key = (out[0].tobytes(), out[1].tobytes(), out[2].tobytes())
    ref = ref or key
    s = out[3]
    print("%-28s step min %.2f med %.2f | bounds %.2f seeds %.2f lists %.2f tables %.2f (narrow %.2f) lanes %.2f | pairs %d lanes %d narrow %d | same %s" % (
        v or "(default)", min(ts), sorted(ts)[6], s["bound_kernel_ms"], s["seed_kernel_ms"], s["list_kernel_ms"], s["scan_kernel_ms"], s["narrow_kernel_ms"], s["lanes_kernel_ms"],
        s["pairs_evaluated"], s["pairs_lanes"], s["pairs_narrow"], key == ref), flush=True)
