"""The candidate-vs-candidate infix graph of bench.py's `hw_graph` leg (4 900 near-identical candidates, window 40, k = 25):
wall time of isocon_hw_pairs with the host laps (ISOCON_DEBUG=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth, end_invariant_functions as END
from isocon_amd.store import SeqStore
accs, seqs, true_isoforms = synth.make_reads(2000, 2500, 10, 30001)
grng = np.random.Generator(np.random.PCG64(77))
cset = set()
for iso in true_isoforms:
    arr = np.frombuffer(iso.encode("ascii"), dtype=np.uint8)
    for _ in range(490):
        v = synth.mutate(grng, arr, dict(rate=0.0012, ins=0.4, dele=0.4, sub=0.2))
        a, b = int(grng.integers(0, 12)), int(grng.integers(0, 12))
        cset.add(v[a:len(v) - b].tobytes().decode())
cseqs = sorted(cset, key=len)
clens = np.fromiter((len(x) for x in cseqs), dtype=np.int64, count=len(cseqs))
gq, gt = END._window_pairs(clens, 0, len(cseqs), 40, 2 ** 32)
stg = SeqStore(cseqs)
gk = np.full(len(gq), 25, dtype=np.int32)
stg.hw_pairs(gq[:4096], gt[:4096], gk[:4096])
ref = None
for rep in range(4):
    if rep == 3: os.environ["ISOCON_DEBUG"] = "1"
    t0 = time.perf_counter(); gres, g_ms = stg.hw_pairs(gq, gt, gk, return_ms=True, reuse_buffer=True); w = time.perf_counter() - t0
    import hashlib
    dig = hashlib.sha1(gres.tobytes()).hexdigest()[:16]
    print("hw_graph: %d pairs, %d hits, wall %.1f ms, kernels %.1f ms, digest %s" % (len(gq), int((gres[:, 0] >= 0).sum()), w * 1e3, g_ms, dig), flush=True)
stg.close()
