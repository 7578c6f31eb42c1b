"""Where a rank's time goes in the sharded search of the C3 set at WORLD emulated ranks (tests/baton_dist.py: one rank at a time on this GPU):
per phase the kernel components of every rank's stats and the host wall time of the phase call (measured while the rank holds the baton).
Usage: python scripts/dev/shard_breakdown.py [WORLD=8] [N_READS=50000]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from baton_dist import run_ranks
from isocon_amd import synth
from isocon_amd.dist import sharded_nn_graph
from isocon_amd.store import SeqStore

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
accs, seqs, _ = synth.make_reads(n_reads, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
torch.cuda.set_device(0)
stores = [SeqStore(seqs, private_pool=True) for _ in range(world)]
KEYS = ("kernel_ms", "bound_kernel_ms", "mm_kernel_ms", "seed_kernel_ms", "list_kernel_ms", "filter_kernel_ms", "scan_kernel_ms", "lanes_kernel_ms", "narrow_kernel_ms")


def rank_main(dist, rank):
    torch.cuda.set_device(0)
    sharded_nn_graph(stores[rank], dist=dist, return_stats=True)
    sharded_nn_graph(stores[rank], dist=dist, return_stats=True)
    laps = {}
    t0 = time.perf_counter()
    out = sharded_nn_graph(stores[rank], dist=dist, return_stats=True, laps=laps)
    return out[3], laps, time.perf_counter() - t0


res, group = run_ranks(world, rank_main)
for k in range(2):
    print("phase %d" % k)
    for key in KEYS:
        v = [float(r[0][k].get(key, 0.0)) for r in res]
        print("   %-18s max %6.3f  mean %6.3f" % (key, max(v), float(np.mean(v))))
    w = [r[1].get("nn_partial_phase%d" % k, 0.0) * 1e3 for r in res]
    print("   %-18s max %6.3f  mean %6.3f" % ("phase call wall", max(w), float(np.mean(w))))
for part in ("fingerprint", "setup", "reduce_min", "gather_edges", "finalize"):
    w = [r[1].get(part, 0.0) * 1e3 for r in res]
    print("%-22s (wall incl. waiting for the other ranks' turns) max %6.3f  min %6.3f" % (part, max(w), min(w)))
for s in stores:
    s.close()
