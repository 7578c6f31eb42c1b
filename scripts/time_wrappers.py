"""End-to-end wall time of the Python drop-in functions at C3 scale (host overheads included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isocon_amd import synth
from isocon_amd import nearest_neighbor_graph as NNG, edlib_alignment_module as EAM, SW_alignment_module as SWM

class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
accs, seqs, _ = synth.make_reads(n, 2500, 10, 30001)
S = dict(zip(accs, seqs))
t = time.time(); g, iso = NNG.compute_nearest_neighbor_graph(S, set(), P()); t_nn = time.time() - t
print("compute_nearest_neighbor_graph: %.2f s (%d rows) stats %s" % (t_nn, len(g), {k: NNG.LAST_STATS[k] for k in ("kernel_ms", "scan_kernel_ms")}))
# partition-like input: each read's first NN is its "centre"
matches = {}
for acc, nbrs in g.items():
    if nbrs:
        matches.setdefault(S[next(iter(nbrs))], set()).add(S[acc])
npairs = sum(len(v) for v in matches.values())
t = time.time(); ed = EAM.edlib_align_sequences(matches, nr_cores=16); t_ed = time.time() - t
print("edlib_align_sequences: %.2f s (%d pairs)" % (t_ed, npairs))
t = time.time(); al = SWM.sw_align_sequences(ed, nr_cores=16); t_sw = time.time() - t
print("sw_align_sequences: %.2f s (%d pairs)" % (t_sw, sum(len(v) for v in al.values())))
