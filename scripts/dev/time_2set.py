"""GPU box: 2-set NN search (reads vs candidates) at the shape of the statistical filter's reassignment rounds: 50 000 C3 reads against
~4 900 near-identical candidates of the 10 isoforms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, iso = synth.make_reads(50000, 2500, 10, 30001)
seqs = list(dict.fromkeys(seqs))
grng = np.random.Generator(np.random.PCG64(77))
cset = set()
for s in iso:
    arr = np.frombuffer(s.encode("ascii"), dtype=np.uint8)
    for _ in range(490):
        v = synth.mutate(grng, arr, dict(rate=0.0012, ins=0.4, dele=0.4, sub=0.2))
        a, b = int(grng.integers(0, 12)), int(grng.integers(0, 12))
        cset.add(v[a:len(v) - b].tobytes().decode())
cands = [c for c in cset if c not in set(seqs)]
merged = sorted([(s, 0) for s in seqs] + [(c, 1) for c in cands], key=lambda x: len(x[0]))
st = SeqStore([s for s, _ in merged])
is_t = np.array([f for _, f in merged], dtype=np.uint8)
ref = None
for env in ({}, {"ISOCON_DEBUG_VARIANT": "nn_no_list"}, {"ISOCON_DEBUG_VARIANT": "nn_no_qgram"}):
    os.environ.update(env)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); b, rp, c, stats = st.nn_graph(is_target=is_t); ts.append((time.perf_counter() - t0) * 1e3)
    for k in env: del os.environ[k]
    key = (b.tobytes(), rp.tobytes(), c.tobytes())
    ref = ref or key
    print(env, "wall %.1f ms kernels %.1f (bounds %.1f seeds %.1f lists %.1f tables %.1f lanes %.1f) pairs %d edges %d same %s" % (
        min(ts), stats["kernel_ms"], stats["bound_kernel_ms"], stats["seed_kernel_ms"], stats["list_kernel_ms"], stats["scan_kernel_ms"], stats["lanes_kernel_ms"],
        stats["pairs_evaluated"], len(c), key == ref), flush=True)
