import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import synth
from isocon_amd.store import SeqStore
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
ref = None
for tag, env in (("segments", {}), ("one segment", {"ISOCON_NN_ONE_SEG": "1"}), ("refill 4 waves", {"ISOCON_NN_NO_SEG": "1"})):
    for k, v in env.items(): os.environ[k] = v
    for rep in range(3):
        t0 = time.perf_counter(); best, rp, cols, s = st.nn_graph(); wall = (time.perf_counter() - t0) * 1e3
    wc = s["cells_columns"] / 64.0
    if ref is None: ref = (best.copy(), rp.copy(), cols.copy())
    same = (best == ref[0]).all() and (rp == ref[1]).all() and (cols == ref[2]).all()
    print("%-16s wall %.1f main %.2f bounds %.2f seed %.2f | pairs %.3e prefiltered %.3e wave-cols %.4e live %.3f -> %.1f ns/1e3 wave-cols/SIMD  same graph %s slow-blocks %d of %d" % (tag, wall, s["scan_kernel_ms"], s["bound_kernel_ms"], s["seed_kernel_ms"], s["pairs_evaluated"], s["pairs_prefiltered"], wc, s["live_columns"] / s["cells_columns"], s["scan_kernel_ms"] * 1e6 * 1024 / wc, same, s["tiles"] & 0xffffffff, wc / 32), flush=True)
    for k in env: del os.environ[k]
