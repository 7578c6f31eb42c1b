"""Randomised parity stress of isocon_hw_pairs against the oracle's full-matrix restatement (all five outputs).
Usage: python scripts/stress_hw.py [n_rounds]"""
import random
import sys
import time

sys.path.insert(0, ".")
from isocon_amd.store import SeqStore
from oracle import oracle as O

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(2026)


def hw_row(x, y, k):
    ed, start, end = O.hw_locate(x, y, k)
    if ed < 0:
        return [-1, -1, -1, 0, 0]
    _, ops = O.nw_path(x, y[start:end + 1])
    return [ed, start, end, ops[0][0] if ops[0][1] == "I" else 0, ops[-1][0] if ops[-1][1] == "I" else 0]


def mut(b, nmut, ends, alphabet):
    v = list(b)
    for _ in range(nmut):
        p = rng.randrange(len(v))
        r = rng.random()
        if r < 0.35:
            v[p] = rng.choice(alphabet)
        elif r < 0.7:
            del v[p]
        else:
            v.insert(p, v[p] if rng.random() < 0.5 else rng.choice(alphabet))
    v = "".join(v)
    v = v[rng.randint(0, ends):len(v) - rng.randint(0, ends)]
    if rng.random() < 0.4:
        v = "".join(rng.choice(alphabet) for _ in range(rng.randint(1, ends + 1))) + v
    if rng.random() < 0.4:
        v += "".join(rng.choice(alphabet) for _ in range(rng.randint(1, ends + 1)))
    return v


total = bad = hits = 0
t0 = time.time()
for r in range(rounds):
    L = rng.choice([40, 90, 200, 500, 1200, 3000])
    k = rng.choice([0, 3, 10, 25, 25, 40, 60])
    ends = rng.choice([0, 5, 15, 30])
    alphabet = rng.choice(["ACGT", "ACGT", "AC", "AAAC"])          # low-complexity alphabets: many equally good paths
    npairs = max(8, 12000 // L)
    seqs = []
    for p in range(npairs):
        b = "".join(rng.choice(alphabet) for _ in range(rng.randint(max(8, L - L // 6), L + L // 6)))
        x = mut(b, rng.choice([0, 1, 2, 5, 12, 30]), ends, alphabet)
        y = mut(b, rng.choice([0, 1, 2, 5]), ends, alphabet)
        if len(x) and len(y):
            seqs += [x, y]
    n = len(seqs) // 2
    try:
        got = SeqStore(seqs).hw_pairs(range(0, 2 * n, 2), range(1, 2 * n, 2), k)
    except RuntimeError as e:
        if "not supported" in str(e):
            continue
        raise
    for p in range(n):
        exp = hw_row(seqs[2 * p], seqs[2 * p + 1], k)
        total += 1
        hits += exp[0] >= 0
        if list(got[p]) != exp:
            bad += 1
            if bad <= 5:
                print("MISMATCH", k, seqs[2 * p], seqs[2 * p + 1], list(got[p]), exp, flush=True)
    print("round %d: L=%d k=%d ends=%d alphabet=%s pairs=%d  (total %d, hits %d, mismatches %d, %.0f s)" % (r, L, k, ends, alphabet, n, total, hits, bad, time.time() - t0), flush=True)
print("stress_hw: %d pairs, %d hits, %d mismatches" % (total, hits, bad))
sys.exit(1 if bad else 0)
