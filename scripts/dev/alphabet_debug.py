import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
from oracle import oracle as O
accs, seqs, _ = synth.make_reads(3000, 700, 4, 1003)
seqs = list(dict.fromkeys(seqs))
seqs = [s.replace("AACA", "aaca") for s in seqs]
seqs = sorted(dict.fromkeys(seqs), key=len)
img = [s.upper() for s in seqs]
n = len(seqs)
st = SeqStore(seqs)
best, rp, cols, stats = st.nn_graph()
# oracle rows via pairwise: brute force for a few rows
sti = SeqStore(img) if len(set(img)) == len(img) else None
print("images distinct:", len(set(img)) == len(img))
bi = None
if sti is not None:
    bi, rpi, ci, _ = sti.nn_graph()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import Params
S = {"r%d" % i: s for i, s in enumerate(seqs)}
o = O.compute_nearest_neighbor_graph(S, set(), Params(8))[0]
bad = []
for x in range(n):
    want = o.get("r%d" % x, {})
    got = {"r%d" % y: int(best[x]) for y in cols[rp[x]:rp[x + 1]].tolist()}
    if want != got:
        bad.append(x)
print("bad rows", len(bad), bad[:10])
for x in bad[:4]:
    want = o.get("r%d" % x, {})
    nb = sorted(int(k[1:]) for k in want)
    m = list(want.values())[0]
    got = sorted(cols[rp[x]:rp[x + 1]].tolist())
    dsi = [(O.ed_dp(img[x], img[y]), y) for y in range(n) if y != x and abs(len(seqs[y]) - len(seqs[x])) <= 63]
    mi = min(d for d, _ in dsi)
    print("row", x, "len", len(seqs[x]), "true", m, nb[:6], "gpu", best[x], got[:6], "image min", mi, [y for d, y in dsi if d == mi][:6], "gpu image-store best", None if bi is None else bi[x],
          "d' of true nn:", [O.ed_dp(img[x], img[y]) for y in nb[:4]], "exact d of image nn:", [O.ed_dp(seqs[x], seqs[y]) for d, y in dsi if d == mi][:6],
          "pairs with d' <= gpu best:", sum(1 for d, y in dsi if d <= best[x]))
