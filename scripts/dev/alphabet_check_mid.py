"""Mid-size parity of sets with exceptional reads against the oracle loop (thousands of reads: the list / table kernels take part)."""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import Params, ordered
from isocon_amd import synth, nearest_neighbor_graph as NNG
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 700
bad = 0
cases = [(1, 0.1, "N"), (2, 0.02, "N"), (3, 1.0, "mask"), (4, 0.3, "lower")]
if len(sys.argv) > 3:
    r0 = random.Random(int(sys.argv[3]))
    cases = [(100 + i, r0.choice([0.01, 0.05, 0.3, 1.0]), r0.choice(["N", "mask", "lower"])) for i in range(int(sys.argv[4]) if len(sys.argv) > 4 else 8)]
for seed, frac, mode in cases:
    accs, seqs, _ = synth.make_reads(n, L, 4, 1000 + seed)
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
    t0 = time.time(); g = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))[0]; t1 = time.time()
    o = O.compute_nearest_neighbor_graph(S, set(), Params(8))[0]; t2 = time.time()
    same = ordered(g) == ordered(o)
    bad += not same
    print("seed %d %s frac %.2f: %d reads, gpu %.2f s, oracle %.1f s, pairs_bytes %d, same %s" % (seed, mode, frac, len(seqs), t1 - t0, t2 - t1, NNG.LAST_STATS.get("pairs_bytes", -1), same), flush=True)
    if not same:
        diff = [k for k in o if g.get(k) != o.get(k)] + [k for k in g if k not in o]
        print("   differing rows:", len(diff), diff[:5])
        for k in diff[:3]:
            print("   ", k, "gpu", g.get(k), "oracle", o.get(k))
sys.exit(1 if bad else 0)
