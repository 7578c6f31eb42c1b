"""g17: the WHOLE exact nearest-neighbour graph of a BASELINE configuration as a fixture -- every row recomputed on the CPU with the
oracle's statement of the reference loop (oracle.nn_1set = NNG:110-198, one query at a time, no GPU anywhere), so that the -m gpu
tests compare all 50 000 (C3) / 5 000 (C2) rows with it each round instead of a sample.

    python tests/golden/make_golden_g17.py c2|c3 [cores]        ->  tests/golden/g17_<cfg>_graph.npz

Arrays: best int32[n] (-1 = no neighbour), row_ptr int64[n+1], cols uint32[]; rows in the reference's insertion order; entries =
sorted(dict.fromkeys(seqs), key=len) of synth.make_reads(<the configuration's arguments>), the order every test and bench.py use.
`inputs_sha1` pins the sequence set the rows belong to.  C3 takes about 6 core-hours / cores."""
import hashlib
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from isocon_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

CONFIGS = {"c2": (5000, 1500, 3, 20001), "c3": (50000, 2500, 10, 30001)}
G = {}


def entries(which):
    accs, seqs, _ = synth.make_reads(*CONFIGS[which])
    return sorted(dict.fromkeys(seqs), key=len)


def inputs_sha1(seqs):
    h = hashlib.sha1()
    for s in seqs:
        h.update(s.encode())
        h.update(b"\n")
    return h.hexdigest()


def rows(lo_hi):
    lo, hi = lo_hi
    out = []
    for i in range(lo, hi):
        rp, c, e, calls = O.nn_1set(G["seqs"], G["conv"], i, 1, packed=G["packed"])
        out.append((c.astype(np.uint32), int(e[0]) if len(e) else -1))
    return lo, out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    cores = int(sys.argv[2]) if len(sys.argv) > 2 else len(os.sched_getaffinity(0))
    O.build()
    seqs = entries(which)
    n = len(seqs)
    G.update(seqs=seqs, conv=np.zeros(n, np.uint8), packed=O.pack(seqs))
    t0 = time.time()
    step = 50
    jobs = [(lo, min(lo + step, n)) for lo in range(0, n, step)]
    res = {}
    with Pool(cores) as pool:          # (fork: the workers share the packed set)
        for k, (lo, out) in enumerate(pool.imap_unordered(rows, jobs)):
            res[lo] = out
            if k % 100 == 0:
                print("%d / %d rows, %.0f s" % (k * step, n, time.time() - t0), flush=True)
    best = np.empty(n, np.int32)
    row_ptr = np.zeros(n + 1, np.int64)
    cols = []
    for lo, _ in jobs:
        for j, (c, d) in enumerate(res[lo]):
            i = lo + j
            best[i] = d
            row_ptr[i + 1] = row_ptr[i] + len(c)
            cols.append(c)
    cols = np.concatenate(cols) if cols else np.zeros(0, np.uint32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g17_%s_graph.npz" % which)
    np.savez_compressed(path, best=best, row_ptr=row_ptr, cols=cols, inputs_sha1=np.array(inputs_sha1(seqs)),
                        config=np.array(repr(CONFIGS[which])))
    print("%s: %d rows, %d edges, %.0f s on %d cores -> %s (%d bytes)" % (which, n, len(cols), time.time() - t0, cores, path, os.path.getsize(path)))
