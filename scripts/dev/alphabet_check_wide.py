"""Parity of the wide phase (nearest neighbours > 63 away) on sets with exceptional reads against the oracle loop."""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import Params, ordered
from isocon_amd import synth, nearest_neighbor_graph as NNG
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
bad = 0
for seed, frac, mode, lr in [(1, 0.05, "N", (500, 900)), (2, 0.3, "lower", (500, 900)), (3, 1.0, "mask", (500, 900)), (4, 0.1, "N", (2500, 4500)), (5, 0.5, "lower", (5000, 9000))]:
    nn = n if lr[1] < 1000 else (n // 3 if lr[1] < 5000 else n // 10)
    accs, seqs, _ = synth.make_reads(nn, 0, 12, 7000 + seed, profile=synth.ONT_PROFILE, families=3, length_range=lr)
    rng = random.Random(seed)
    seqs = list(dict.fromkeys(seqs))
    if mode == "N":
        for i in rng.sample(range(len(seqs)), int(frac * len(seqs))):
            for _ in range(rng.randrange(1, 4)):
                p = rng.randrange(len(seqs[i])); seqs[i] = seqs[i][:p] + "N" + seqs[i][p + 1:]
    elif mode == "mask":
        seqs = [s.replace("AACA", "aaca") for s in seqs]
    else:
        for i in rng.sample(range(len(seqs)), int(frac * len(seqs))):
            a = rng.randrange(len(seqs[i]) - 30); seqs[i] = seqs[i][:a] + seqs[i][a:a + 12].lower() + seqs[i][a + 12:]
    seqs = list(dict.fromkeys(seqs))
    S = {"r%d" % i: s for i, s in enumerate(seqs)}
    conv = set(rng.sample(sorted(S), 5))
    t0 = time.time(); g = NNG.compute_nearest_neighbor_graph(S, conv, Params(1))[0]; t1 = time.time()
    st = dict(NNG.LAST_STATS)
    o = O.compute_nearest_neighbor_graph(S, conv, Params(8))[0]; t2 = time.time()
    same = ordered(g) == ordered(o)
    bad += not same
    import numpy as np
    med = int(np.median([list(v.values())[0] for v in o.values() if v]))
    print("seed %d %s frac %.2f len %s: %d reads, gpu %.2f s, oracle %.1f s, median NN distance %d, wide queries %d, pairs_bytes %d, same %s" % (
        seed, mode, frac, lr, len(seqs), t1 - t0, t2 - t1, med, st.get("fallback_queries", -1), st.get("pairs_bytes", -1), same), flush=True)
    if not same:
        diff = [k for k in o if g.get(k) != o.get(k)] + [k for k in g if k not in o]
        print("   differing rows:", len(diff), diff[:5])
        for k in diff[:3]:
            print("   ", k, "gpu", dict(list(g.get(k, {}).items())[:5]), "oracle", dict(list(o.get(k, {}).items())[:5]))
sys.exit(1 if bad else 0)
