import os, sys, time, importlib.util
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from isocon_amd import nearest_neighbor_graph as NNG
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("m", os.path.join(root, "tests", "golden", "make_golden_g19.py")); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
X, C = mod.candidates("c3")
class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False
NNG.compute_2set_nearest_neighbor_graph(X, C, P())
t0=time.perf_counter(); g = NNG.compute_2set_nearest_neighbor_graph(X, C, P()); print("wall %.1f ms" % (1e3*(time.perf_counter()-t0)))
print({k: (round(v,3) if isinstance(v,float) else v) for k,v in NNG.LAST_STATS.items()})
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); NNG.compute_2set_nearest_neighbor_graph(X, C, P()); pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(10)
