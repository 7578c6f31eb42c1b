"""GPU box: the 32-row form of the table kernel on reads with few errors (50 000 x 2.5 kb, CCS split at 0.3 % instead of C3's 1 %):
step time with the form chosen on the device against the 64-row form forced (ISOCON_DEBUG_VARIANT=nn_narrow=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
rate = float(sys.argv[1]) if len(sys.argv) > 1 else 0.003
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001, profile=dict(synth.CCS_PROFILE, rate=rate))
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
ref = None
for env in (None, "0"):
    if env is None: os.environ.pop("ISOCON_DEBUG_VARIANT", None)
    else: os.environ["ISOCON_DEBUG_VARIANT"] = "nn_narrow=" + env
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); best, rp, cols, s = st.nn_graph(); ts.append((time.perf_counter() - t0) * 1e3)
    key = (best.tobytes(), rp.tobytes(), cols.tobytes())
    if ref is None: ref = key
    print("rate %.4f n %d NARROW=%s | step %.2f ms | bounds %.2f seeds %.2f lists %.2f tables %.2f lanes %.2f | pairs %d (lanes %d, wide %d) | median NN distance %.0f | same graph %s"
          % (rate, st.n, env, min(ts[1:]), s["bound_kernel_ms"], s["seed_kernel_ms"], s["list_kernel_ms"], s["scan_kernel_ms"], s["lanes_kernel_ms"],
             s["pairs_evaluated"], s["pairs_lanes"], s["pairs_wide_to_lanes"], np.median(best[best >= 0]), key == ref), flush=True)
