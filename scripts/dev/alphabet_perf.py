"""Cost of the byte-wise path at C3: a fraction of the reads gets one 'N'."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
accs, seqs, _ = synth.make_reads(n, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
rng = random.Random(1)
for frac in [float(x) for x in sys.argv[2:]] or [0.0, 0.001, 0.01]:
    ss = list(seqs)
    for i in rng.sample(range(len(ss)), int(frac * len(ss))):
        p = rng.randrange(len(ss[i]))
        ss[i] = ss[i][:p] + "N" + ss[i][p + 1:]
    ss = sorted(dict.fromkeys(ss), key=len)
    t0 = time.perf_counter(); st = SeqStore(ss); t1 = time.perf_counter()
    st.nn_graph()
    t2 = time.perf_counter(); out = st.nn_graph(); t3 = time.perf_counter()
    s = out[3]
    print("frac %.4f: store %.1f ms, graph %.1f ms (kernels %.1f), byte-wise pairs %d, pairs evaluated %d, edges %d" % (
        frac, 1e3 * (t1 - t0), 1e3 * (t3 - t2), s["kernel_ms"], s["pairs_bytes"], s["pairs_evaluated"], len(out[2])), flush=True)
    st.close()
if os.environ.get("MIXED_CASE"):
    # soft-masked reads: the same motif in lower case wherever it occurs -- every read exceptional, distances nearly those of the ACGT set
    ss = sorted(dict.fromkeys(s.replace("AACA", "aaca") for s in seqs), key=len)
    st = SeqStore(ss)
    st.nn_graph()
    t2 = time.perf_counter(); out = st.nn_graph(); t3 = time.perf_counter()
    s = out[3]
    print("soft-masked, all %d reads: graph %.1f ms (kernels %.1f), byte-wise pairs %d, edges %d" % (len(ss), 1e3 * (t3 - t2), s["kernel_ms"], s["pairs_bytes"], len(out[2])), flush=True)
