"""Candidate-vs-candidate infix alignments at the size of the C3 pipeline (about 4 900 candidates of 10 isoforms, every
candidate against its length window 10 + 2 x 15: a few million pairs through isocon_hw_pairs).  Prints wall and kernel time.
Usage: python scripts/time_hw_graph.py [candidates_per_isoform] [length]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isocon_amd import synth, end_invariant_functions as END
from isocon_amd.store import SeqStore

per = int(sys.argv[1]) if len(sys.argv) > 1 else 490
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
rng = np.random.Generator(np.random.PCG64(77))
isoforms = synth.make_isoforms(rng, L, 10)
prof = dict(rate=0.0012, ins=0.4, dele=0.4, sub=0.2)            # ~3 residual errors per candidate
seqs = set()
for iso in isoforms:
    for _ in range(per):
        s = synth.mutate(rng, iso, prof)
        a, b = int(rng.integers(0, 12)), int(rng.integers(0, 12))
        seqs.add(s[a:len(s) - b].tobytes().decode())
seqs = sorted(seqs, key=len)
lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
q, t = END._window_pairs(lens, 0, len(seqs), 40, 2 ** 32)
st = SeqStore(seqs)
k = np.full(len(q), 25, dtype=np.int32)
st.hw_pairs(q[:4096], t[:4096], k[:4096])
for rep in range(3):
    t0 = time.perf_counter()
    res, ms = st.hw_pairs(q, t, k, return_ms=True)
    wall = time.perf_counter() - t0
    print("%d candidates, %d pairs, %d hits: wall %.3f s, kernels %.1f ms (%.3g pairs/s kernel)" % (len(seqs), len(q), int((res[:, 0] >= 0).sum()), wall, ms, len(q) / (ms / 1e3)), flush=True)
