"""Staged check of isocon_hw_pairs against the oracle (prints after every stage)."""
import random
import sys
import time

sys.path.insert(0, ".")
import numpy as np
from isocon_amd.store import SeqStore
from oracle import oracle as O


def hw_row(x, y, k):
    ed, start, end = O.hw_locate(x, y, k)
    if ed < 0:
        return [-1, -1, -1, 0, 0]
    _, ops = O.nw_path(x, y[start:end + 1])
    return [ed, start, end, ops[0][0] if ops[0][1] == "I" else 0, ops[-1][0] if ops[-1][1] == "I" else 0]


def say(*a):
    print(*a, flush=True)


say("stage 0: store")
st = SeqStore(["ACGTACGTACGTACGTACGT", "TTACGTACGTACGTACGTACGTGG", "ACGTACGTTCGTACGTACGT"])
say("stage 1: one no-hit pair (query much longer than target + k)")
say(st.hw_pairs([1], [0], [1]))
say("stage 2: one pair, k = 5")
t0 = time.time()
say(st.hw_pairs([0], [1], [5]), hw_row("ACGTACGTACGTACGTACGT", "TTACGTACGTACGTACGTACGTGG", 5), "%.3f s" % (time.time() - t0))
say("stage 3: three pairs")
say(st.hw_pairs([0, 2, 0], [1, 0, 2], [5, 5, 0]))
rng = random.Random(1)
for L, k, npairs in ((100, 10, 50), (300, 25, 200), (2500, 25, 64), (300, 40, 50)):
    seqs = []
    for p in range(npairs):
        b = "".join(rng.choice("ACGT") for _ in range(L))
        x = list(b)
        for _ in range(rng.randint(0, 6)):
            x[rng.randrange(len(x))] = rng.choice("ACGT")
        seqs += ["".join(x)[rng.randint(0, 10):], b[rng.randint(0, 10):L - rng.randint(0, 10)]]
    t0 = time.time()
    got, ms = SeqStore(seqs).hw_pairs(range(0, 2 * npairs, 2), range(1, 2 * npairs, 2), k, return_ms=True)
    bad = sum(list(got[p]) != hw_row(seqs[2 * p], seqs[2 * p + 1], k) for p in range(npairs))
    say("stage L=%d k=%d pairs=%d: mismatches %d, hits %d, kernel %.2f ms, wall %.3f s" % (L, k, npairs, bad, int((got[:, 0] >= 0).sum()), ms, time.time() - t0))
say("done")
