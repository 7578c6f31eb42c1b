"""Full-matrix alignments (no hints: k_sg_forward + k_sg_walk) of 4096 C3 read pairs: kernel ms and a digest, for A/B runs of the strip kernel
(ISOCON_LIB=isocon_amd/lib/libisocon_hip_<variant>.so python scripts/dev/sw_full_ab.py)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore, sg_last_stats
accs, seqs, _ = synth.make_reads(8192, 2500, 10, 30001)
st = SeqStore(seqs)
a = np.arange(0, 8192, 2, dtype=np.uint32); b = a + 1
mm = np.full(len(a), -2, dtype=np.int8)
for rep in range(3):
    ops, ptr, res, ms = st.sg_trace(a, b, mm, return_ms=True)
    s = sg_last_stats()
    print("full matrix, 4096 pairs: kernels %.2f ms (forward %.2f, walk %.2f)  digest %s" % (ms, s["forward_ms"], s["walk_ms"],
          hashlib.sha1(ops.tobytes() + ptr.tobytes() + res.tobytes()).hexdigest()[:12]))
for policy, open_, ext in ((21, 2, 0), (0, 3, 1)):
    ops, ptr, res, ms = st.sg_trace(a[:512], b[:512], mm[:512], open_=open_, ext=ext, tie_policy=policy, return_ms=True)
    print("policy %d open %d ext %d: %.2f ms digest %s" % (policy, open_, ext, ms, hashlib.sha1(ops.tobytes() + res.tobytes()).hexdigest()[:12]))
