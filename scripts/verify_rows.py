"""Full-size parity evidence beyond what the test suite samples: rows of the exact NN graph of a BASELINE configuration
recomputed with the reference loop (oracle restatement of NNG:110-198) for many random queries on all host cores, then compared
with the GPU graph -- neighbours, their order and the distance.  Usage: python scripts/verify_rows.py c3|c2|c5 [n_rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multiprocessing import Pool
from isocon_amd import synth
from oracle import oracle as O

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
if which == "c3":
    accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
elif which == "c2":
    accs, seqs, _ = synth.make_reads(5000, 1500, 3, 20001)
else:
    accs, seqs, _ = synth.make_reads(20000, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
seqs = sorted(dict.fromkeys(seqs), key=len)
G = {"packed": O.pack(seqs), "conv": np.zeros(len(seqs), np.uint8)}


def row(i):
    rp, c, e, calls = O.nn_1set(seqs, G["conv"], int(i), 1, packed=G["packed"])
    return int(i), c.tolist(), (int(e[0]) if len(e) else -1)


if __name__ == "__main__":
    O.build()
    rows = np.random.default_rng(2026).choice(len(seqs), min(n_rows, len(seqs)), replace=False).tolist()
    cores = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = min(cores, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    t0 = time.time()
    with Pool(cores) as pool:                      # before this process touches the GPU
        ref = pool.map(row, rows, chunksize=4)
    t_cpu = time.time() - t0
    from isocon_amd.store import SeqStore
    st = SeqStore(seqs)
    t0 = time.time(); best, row_ptr, cols, stats = st.nn_graph(); t_gpu = time.time() - t0
    bad = 0
    for i, c, d in ref:
        got = cols[row_ptr[i]:row_ptr[i + 1]].tolist()
        if got != c or (int(best[i]) != d):
            bad += 1
            if bad < 5:
                print("MISMATCH row", i, got, c, int(best[i]), d)
    print("%s: %d sequences, %d rows recomputed with the reference loop on %d cores in %.0f s (GPU graph of all rows: %.2f s): %d mismatches"
          % (which, len(seqs), len(ref), cores, t_cpu, t_gpu, bad))
    sys.exit(1 if bad else 0)
