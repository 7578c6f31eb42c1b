import cProfile, pstats, sys, time, os
sys.path.insert(0, os.getcwd())
import bench
from isocon_amd import nearest_neighbor_graph as NNG
X, C, merged, z, mod = bench.two_set_inputs()
class P: nr_cores = 1; neighbor_search_depth = 2 ** 32; verbose = False
for i in range(3):
    t0 = time.perf_counter(); g = NNG.compute_2set_nearest_neighbor_graph(X, C, P()); print("2set %.1f ms kernels %.1f" % (1e3 * (time.perf_counter() - t0), NNG.LAST_STATS["kernel_ms"])); del g
pr = cProfile.Profile(); pr.enable(); g = NNG.compute_2set_nearest_neighbor_graph(X, C, P()); pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
