"""How much of the main pass is due to thresholds that are not yet final?  Main pass (nn_partial phase 1) started from
the seed bounds (normal) vs from the FINAL best[] (an oracle nobody has), at C3 and at 200 k reads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth, _lib
from isocon_amd.store import SeqStore
for n in [int(x) for x in (sys.argv[1:] or ["50000", "200000"])]:
    accs, seqs, _ = synth.make_reads(n, 2500, 10, 30001)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    best, rp, cols, stats = st.nn_graph()
    print("n=%d normal   main %.1f ms seed %.1f ms pairs %.4g lane-cols %.3g" % (n, stats["scan_kernel_ms"], stats["seed_kernel_ms"], stats["pairs_evaluated"], stats["cells_columns"]))
    b = np.where(best < 0, _lib.NN_INF, best).astype(np.int32)
    hits, s2 = st.nn_partial(0, st.n, 1, b.copy())
    print("n=%d perfect  main %.1f ms pairs %.4g lane-cols %.3g" % (n, s2["scan_kernel_ms"], s2["pairs_evaluated"], s2["cells_columns"]))
    for add in (2, 4, 8):
        b2 = np.where(best < 0, _lib.NN_INF, best + add).astype(np.int32)
        hits, s3 = st.nn_partial(0, st.n, 1, b2.copy())
        print("n=%d best+%d   main %.1f ms pairs %.4g lane-cols %.3g" % (n, add, s3["scan_kernel_ms"], s3["pairs_evaluated"], s3["cells_columns"]))
    bs = np.full(st.n, _lib.NN_INF, dtype=np.int32)
    st.nn_partial(0, st.n, 0, bs)
    print("n=%d after the seed pass: median best %.0f (final %.0f), mean excess %.1f, entries still without bound %d" %
          (n, np.median(bs[bs < _lib.NN_INF]), np.median(best), float(np.mean((bs - best)[bs < _lib.NN_INF])), int((bs >= _lib.NN_INF).sum())))
    st.close()
    # (appended experiment, last n only) thresholds a few edits above the final ones
