import os, sys
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
dev = stg.hw_pairs(gq, gt, gk)
dev2 = stg.hw_pairs(gq, gt, gk)
os.environ["ISOCON_DEBUG_VARIANT"] = "hw_host_tiles=1"
host = stg.hw_pairs(gq, gt, gk)
print("device twice identical:", bool((dev == dev2).all()))
bad = np.nonzero((dev != host).any(axis=1))[0]
print("rows differing:", len(bad), "of", len(gq))
for p in bad[:12]:
    print(p, "q", gq[p], "t", gt[p], "lens", clens[gq[p]], clens[gt[p]], "dev", dev[p], "host", host[p])
print("columns differing:", [(int((dev[:, c] != host[:, c]).sum())) for c in range(5)])
import hashlib, time
print("digest dev", hashlib.sha1(dev.tobytes()).hexdigest()[:16], "host", hashlib.sha1(host.tobytes()).hexdigest()[:16])
t0 = time.perf_counter(); stg.hw_pairs(gq, gt, gk); print("host-tile path wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
del os.environ["ISOCON_DEBUG_VARIANT"]
t0 = time.perf_counter(); stg.hw_pairs(gq, gt, gk); print("device-tile path wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
