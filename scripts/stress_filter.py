"""Randomised stress of the block filter (csrc/nn_filter.hpp): the graph with the filter must be the graph without it (and, on the small cases,
the oracle's) -- odd lengths around the word / piece boundaries, low-complexity alphabets, homopolymer runs, error rates from 0.2 to 8 %, sets
in which nearly every pair survives the q-gram bound, 2-set searches."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore
from oracle import oracle as O

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
tot_rej = 0
for case in range(ncases):
    kind = rng.choice(["synth", "synth", "synth", "lowc", "homo", "edges"])
    if kind == "synth":
        n, L, iso = rng.randint(200, 3000), rng.choice([40, 63, 64, 65, 130, 255, 256, 257, 272, 500, 1000, 2500]), rng.randint(1, 6)
        prof = dict(synth.CCS_PROFILE, rate=rng.choice([0.002, 0.005, 0.01, 0.02, 0.04, 0.08]))
        seqs = synth.make_reads(n, L, iso, seed=rng.randint(0, 10 ** 6), profile=prof)[1]
    elif kind == "lowc":
        n, L = rng.randint(200, 1500), rng.randint(60, 900)
        base = "".join(rng.choice("AC") for _ in range(L))
        seqs = []
        for i in range(n):
            s = list(base)
            for _ in range(rng.randint(0, 20)):
                p = rng.randrange(len(s)); op = rng.random()
                if op < 0.4: s[p] = rng.choice("AC")
                elif op < 0.7: s.insert(p, rng.choice("AC"))
                else: del s[p]
            seqs.append("".join(s) or "A")
    elif kind == "homo":
        n = rng.randint(200, 1200)
        runs = [(rng.choice("ACGT"), rng.randint(1, 14)) for _ in range(rng.randint(20, 120))]
        seqs = []
        for i in range(n):
            seqs.append("".join(c * max(1, l + rng.choice([0, 0, 0, 1, -1])) for c, l in runs))
    else:
        n = rng.randint(100, 800)
        L = rng.choice([19, 20, 21, 35, 36, 37, 255, 256, 257, 271, 272, 273])
        base = "".join(rng.choice("ACGT") for _ in range(L))
        seqs = []
        for i in range(n):
            s = list(base)
            for _ in range(rng.randint(0, 4)):
                p = rng.randrange(len(s)); s[p] = rng.choice("ACGT")
            if rng.random() < 0.3: s.insert(rng.randrange(len(s)), rng.choice("ACGT"))
            seqs.append("".join(s))
    seqs = sorted(dict.fromkeys(seqs), key=len)
    if len(seqs) < 2:
        continue
    two_set = rng.random() < 0.25
    is_t = None
    if two_set:
        is_t = np.zeros(len(seqs), np.uint8); is_t[rng.sample(range(len(seqs)), max(1, len(seqs) // rng.choice([3, 10, 40])))] = 1
    st = SeqStore(seqs)
    os.environ.pop("ISOCON_DEBUG_VARIANT", None)
    g1 = st.nn_graph(is_target=is_t)
    os.environ["ISOCON_DEBUG_VARIANT"] = "nn_no_block_filter"
    g0 = st.nn_graph(is_target=is_t)
    os.environ["ISOCON_DEBUG_VARIANT"] = "nn_filter_one_pass,nn_table_chunks=0"
    g2 = st.nn_graph(is_target=is_t)
    os.environ.pop("ISOCON_DEBUG_VARIANT", None)
    ok = all((a == b).all() for a, b in zip(g1[:3], g0[:3])) and all((a == b).all() for a, b in zip(g2[:3], g0[:3]))
    if ok and len(seqs) <= 600 and not two_set:
        rp, c, e, _ = O.nn_1set(seqs, np.zeros(len(seqs), np.uint8), 0, len(seqs))
        ok = (np.asarray(rp) == g1[1]).all() and (np.asarray(c) == g1[2]).all()
    tot_rej += g1[3]["pairs_block_rejected"]
    if not ok:
        bad += 1
        print("MISMATCH case %d kind %s n %d two_set %s" % (case, kind, len(seqs), two_set), flush=True)
    st.close()
    if case % 20 == 19:
        print("  ... %d cases, %d pairs rejected by the filter so far" % (case + 1, tot_rej), flush=True)
print("stress_filter: %d cases, %d mismatches, %d pairs rejected by the filter" % (ncases, bad, tot_rej))
sys.exit(1 if bad else 0)
