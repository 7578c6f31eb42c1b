"""GPU box: C3 step time and phase split under environment switches (A/B runs): python step_sweep.py VAR=v1,v2 ..."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:]]
ref = None
for combo in itertools.product(*[v for _, v in axes]) if axes else [()]:
    for (k, _), v in zip(axes, combo):
        if v == "-": os.environ.pop(k, None)
        else: os.environ[k] = v
    ts = []
    for rep in range(4):
        t0 = time.perf_counter(); best, rp, cols, stats = st.nn_graph(); ts.append((time.perf_counter() - t0) * 1e3)
    key = (best.tobytes(), rp.tobytes(), cols.tobytes())
    if ref is None: ref = key
    print(" ".join("%s=%s" % (k, v) for (k, _), v in zip(axes, combo)), "| step %.2f ms (min of 3) | bounds %.2f seeds %.2f lists %.2f tables %.2f lanes %.2f | pairs %d (lanes %d, wide %d) filtered %d | live %.3f wave-cols %.3e | same graph %s"
          % (min(ts[1:]), stats["bound_kernel_ms"], stats["seed_kernel_ms"], stats["list_kernel_ms"], stats["scan_kernel_ms"], stats["lanes_kernel_ms"], stats["pairs_evaluated"], stats["pairs_lanes"], stats["pairs_wide_to_lanes"], stats["pairs_prefiltered"], stats["live_columns"] / max(1, stats["cells_columns"]), stats["cells_columns"] / 64.0, key == ref), flush=True)
