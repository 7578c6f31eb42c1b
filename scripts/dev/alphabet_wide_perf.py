"""Wide phase with exceptional reads: ONT-profile reads (nearest neighbours hundreds of edits away), a fraction with an 'N'."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
accs, seqs, _ = synth.make_reads(n, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
seqs = sorted(dict.fromkeys(seqs), key=len)
rng = random.Random(1)
for frac in [float(x) for x in sys.argv[2:]] or [0.0, 0.01]:
    ss = list(seqs)
    for i in rng.sample(range(len(ss)), int(frac * len(ss))):
        p = rng.randrange(len(ss[i])); ss[i] = ss[i][:p] + "N" + ss[i][p + 1:]
    ss = sorted(dict.fromkeys(ss), key=len)
    st = SeqStore(ss)
    t2 = time.perf_counter(); out = st.nn_graph(); t3 = time.perf_counter()
    s = out[3]
    print("frac %.3f: graph %.1f ms (kernels %.1f), byte-wise pairs %d, wide queries %d, edges %d" % (frac, 1e3 * (t3 - t2), s["kernel_ms"], s["pairs_bytes"], s["fallback_queries"], len(out[2])), flush=True)
    st.close()
