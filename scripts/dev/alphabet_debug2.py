import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import Params
from isocon_amd import synth
from isocon_amd.store import SeqStore
from oracle import oracle as O
accs, seqs, _ = synth.make_reads(3000, 700, 4, 1003)
seqs = sorted(dict.fromkeys(s.replace("AACA", "aaca") for s in dict.fromkeys(seqs)), key=len)
n = len(seqs)
S = {"r%d" % i: s for i, s in enumerate(seqs)}
o = O.compute_nearest_neighbor_graph(S, set(), Params(8))[0]
for v in ["", "nn_no_list", "nn_no_qgram", "nn_narrow=0", "nn_old_seed", "no_seed"]:
    if v: os.environ["ISOCON_DEBUG_VARIANT"] = v
    else: os.environ.pop("ISOCON_DEBUG_VARIANT", None)
    st = SeqStore(seqs)
    best, rp, cols, stats = st.nn_graph()
    bad = sum(1 for x in range(n) if o.get("r%d" % x, {}) != {"r%d" % y: int(best[x]) for y in cols[rp[x]:rp[x + 1]].tolist()})
    print("%-14s bad rows %d pairs_bytes %d" % (v or "(default)", bad, stats["pairs_bytes"]), flush=True)
    st.close()
os.environ.pop("ISOCON_DEBUG_VARIANT", None)
st = SeqStore(seqs)
import numpy as np
for (x, nb) in [(7, [86, 88, 76]), (88, [7, 76]), (292, [369, 76, 86])]:
    a = [x] * len(nb)
    print(x, nb, "k=8:", st.ed_pairs(a, nb, [8] * len(nb)).tolist(), "k=63:", st.ed_pairs(a, nb, [63] * len(nb)).tolist(), "none:", st.ed_pairs(a, nb, None).tolist(),
          "dp:", [O.ed_dp(seqs[x], seqs[y]) for y in nb], "exc:", [seqs[y].count("a") for y in [x] + nb])
