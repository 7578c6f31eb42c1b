"""Randomised stress of the sharded NN search (isocon_amd/dist.py, the device-resident protocol) against the one-call search of the same
store: odd-sized sets, 2 / 3 / 4 / 8 emulated ranks (tests/baton_dist.py: every rank a thread with its own store and scratch pool, the
production protocol code), with and without role flags (converged entries, a 2-set split), finite depths.  Every rank's
(best, row_ptr, cols) must equal the direct call's.  Usage: python scripts/stress_sharded.py [SEED=1] [CASES=60]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from baton_dist import run_ranks
from isocon_amd import synth
from isocon_amd.dist import sharded_nn_graph
from isocon_amd.store import SeqStore

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
torch.cuda.set_device(0)
bad = 0
for case in range(ncases):
    kind = rng.choice(["ccs", "ccs", "ont", "short", "families"])
    if kind == "ccs":
        n, L, iso = rng.choice([300, 1100, 2500, 6000]), rng.choice([200, 700, 1500, 2500]), rng.randint(1, 8)
        _, seqs, _ = synth.make_reads(n, L, iso, seed=rng.randint(0, 10 ** 6))
    elif kind == "ont":
        n = rng.choice([400, 1500, 3000])
        _, seqs, _ = synth.make_reads(n, 0, rng.randint(2, 12), seed=rng.randint(0, 10 ** 6), profile=synth.ONT_PROFILE, families=rng.randint(1, 3), length_range=(300, 2500))
    elif kind == "short":
        n = rng.randint(2, 700)
        seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 90))) for _ in range(n)]
    else:
        n = rng.choice([800, 2000])
        prof = dict(synth.CCS_PROFILE, rate=rng.choice([0.005, 0.02, 0.05]))
        _, seqs, _ = synth.make_reads(n, 0, rng.randint(3, 20), seed=rng.randint(0, 10 ** 6), profile=prof, families=rng.randint(1, 4), length_range=(500, 3000))
    seqs = sorted(dict.fromkeys(seqs), key=len)
    n = len(seqs)
    roles = rng.choice(["none", "none", "converged", "two_set"])
    conv = targ = None
    if roles == "converged":
        conv = (np.random.RandomState(case).rand(n) < 0.3).astype(np.uint8)
    elif roles == "two_set":
        targ = (np.random.RandomState(case).rand(n) < rng.choice([0.02, 0.2, 0.5])).astype(np.uint8)
    depth = rng.choice([2 ** 32, 2 ** 32, 2 ** 32, 50, 7]) if roles != "two_set" else 2 ** 32
    world = rng.choice([2, 3, 4, 8])
    st = SeqStore(seqs)
    want = st.nn_graph(is_converged=conv, is_target=targ, depth=depth)[:3]
    st.close()
    stores = [SeqStore(seqs, private_pool=True) for _ in range(world)]

    def rank_main(dist, rank):
        torch.cuda.set_device(0)
        return sharded_nn_graph(stores[rank], is_converged=conv, is_target=targ, depth=depth, dist=dist)

    try:
        res, group = run_ranks(world, rank_main)
    finally:
        for s in stores:
            s.close()
    ok = all((np.asarray(r[0]) == want[0]).all() and (np.asarray(r[1]) == want[1]).all() and (np.asarray(r[2]) == want[2]).all() for r in res)
    if not ok:
        bad += 1
    print("case %3d %-8s n=%5d world=%d roles=%-9s depth=%-10d %s" % (case, kind, n, world, roles, depth, "ok" if ok else "MISMATCH"), flush=True)
print("stress_sharded: %d cases, %d mismatches" % (ncases, bad))
sys.exit(1 if bad else 0)
