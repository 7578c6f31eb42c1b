"""2-set search timing: n reads (C3 profile) against the true isoforms plus mutated variants."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd import nearest_neighbor_graph as NNG
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
ncand = int(sys.argv[2]) if len(sys.argv) > 2 else 100
accs, seqs, isoforms = synth.make_reads(n, 2500, 10, 30001)
rng = np.random.default_rng(1)
cands = list(isoforms)
prof = dict(synth.CCS_PROFILE, rate=0.004)
while len(cands) < ncand:
    base = np.frombuffer(isoforms[len(cands) % len(isoforms)].encode(), dtype=np.uint8)
    cands.append(synth.mutate(rng, base, prof).tobytes().decode())
X = dict(zip(accs, seqs))
C = {"cand_%d" % i: c for i, c in enumerate(dict.fromkeys(cands))}
class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False; develop_logfile = None
for rep in range(3):
    t = time.time(); g = NNG.compute_2set_nearest_neighbor_graph(X, C, P()); dt = time.time() - t
    print("2set wall %.1f ms, reads %d, candidates %d, reads with NN %d, stats %s" % (dt * 1e3, len(X), len(C), sum(1 for k in g if g[k]), getattr(NNG, "LAST_STATS", None)))
