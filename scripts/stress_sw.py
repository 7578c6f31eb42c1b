"""Randomised parity stress of the semi-global aligner (full and banded, all tie policies, several gap models) vs the oracle."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isocon_amd import SW_alignment_module as SWM
from isocon_amd.edlib_alignment_module import _intern
from isocon_amd.store import SeqStore
from oracle import oracle as O

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for rd in range(rounds):
    pairs, mism, hints = [], [], []
    for _ in range(40):
        m = rng.choice([5, 40, 120, 300, 513, 900, 1500, 2100])
        s1 = "".join(rng.choice("ACGT") for _ in range(m))
        s2 = list(s1)
        for _ in range(int(m * rng.choice([0.0, 0.01, 0.03, 0.1]))):
            p = rng.randrange(len(s2)) if s2 else 0
            r = rng.random()
            if r < 0.4 and s2: s2[p] = rng.choice("ACGT")
            elif r < 0.7: s2.insert(p, rng.choice("ACGT"))
            elif s2: del s2[p]
        if rng.random() < 0.25 and len(s2) > 60:
            c = rng.randrange(len(s2) - 50); del s2[c:c + rng.randint(20, 50)]
        if rng.random() < 0.2: s2 = s2[rng.randint(0, 30):]
        s2 = "".join(s2) or "A"
        if rng.random() < 0.5: s1, s2 = s2, s1
        pairs.append((s1, s2)); mism.append(rng.choice([-1, -2, -3, -4]))
        e = O.ed_bounded(s1, s2, -1)
        hints.append(rng.choice([e, e, e + rng.randint(0, 20), max(0, e - rng.randint(1, 30)), -1, 0]))
    policy = rng.choice([0, 0, 1, 2, 4, 8, 16, 31])
    match, open_, ext = rng.choice([(2, 2, 0), (2, 2, 0), (2, 3, 0), (2, 3, 1), (1, 2, 1)])
    seqs, a, b = _intern(pairs)
    st = SeqStore(seqs)
    ops, ptr, res = st.sg_trace(a, b, np.asarray(mism, dtype=np.int8), match=match, open_=open_, ext=ext, tie_policy=policy,
                                ed_upper=np.asarray(hints, dtype=np.int32))
    for p, (s1, s2) in enumerate(pairs):
        exp = O.sg_trace(s1, s2, match, int(mism[p]), open_, ext, policy)
        got = dict(cigar=SWM.ops_to_cigar(ops[ptr[p]:ptr[p + 1]].tolist()), score=int(res[p, 0]), end_query=int(res[p, 1]), end_ref=int(res[p, 2]),
                   matches=int(res[p, 3]), mismatches=int(res[p, 4]), indels=int(res[p, 5]))
        if got != exp:
            bad += 1
            print("MISMATCH round", rd, "pair", p, len(s1), len(s2), mism[p], policy, (match, open_, ext), hints[p])
    st.close()
print("sw stress: %d pairs, %d mismatches" % (rounds * 40, bad))
sys.exit(1 if bad else 0)
