import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import _lib, synth
from isocon_amd.store import SeqStore, nn_finalize
accs, seqs, _ = synth.make_reads(1500, 1200, 3, seed=9)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
for depth in (2 ** 32, 400):
    for noq in (False, True):
        if noq: os.environ["ISOCON_NN_NO_QGRAM"] = "1"
        else: os.environ.pop("ISOCON_NN_NO_QGRAM", None)
        ref = st.nn_graph(depth=depth)
        n = st.n
        best = np.full(n, _lib.NN_INF, np.int32)
        hits_all = []
        for phase in (0, 1, 2):
            parts = []
            for r in range(3):
                b = best.copy()
                hits, stats = st.nn_partial(r, n, phase, b, depth=depth, q_stride=3)
                hits_all.append(hits); parts.append(b)
            best = np.minimum.reduce(parts)
        out = nn_finalize(n, best, np.concatenate(hits_all))
        print("depth", depth, "no_qgram", noq, [bool((x == y).all()) if len(x) == len(y) else (len(x), len(y)) for x, y in zip(out[:3], ref[:3])], "edges", len(ref[2]), len(out[2]),
              "best diff", int((out[0] != ref[0]).sum()))
