"""Randomised parity of the byte-wise path (sets with more than four distinct symbols): pair distances vs the textbook DP and 1-set /
2-set nearest-neighbour graphs vs the oracle loop, over random mixes of ordinary and exceptional sequences.
usage: python scripts/stress_alphabet.py [seed] [cases]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import Params, ordered          # noqa: E402
from isocon_amd import nearest_neighbor_graph as NNG          # noqa: E402
from isocon_amd.store import SeqStore          # noqa: E402
from oracle import oracle as O          # noqa: E402


def family(rng, n, L, max_edits, extra, rate):
    root = [rng.choice("ACGT") for _ in range(L)]
    out = []
    for _ in range(n):
        s = list(root)
        for _ in range(rng.randrange(0, max_edits + 1)):
            i = rng.randrange(len(s))
            r = rng.random()
            if r < 0.4:
                s[i] = rng.choice("ACGT")
            elif r < 0.7:
                del s[i]
            else:
                s.insert(i, rng.choice("ACGT"))
        if rng.random() < rate:
            for _ in range(rng.randrange(1, 6)):
                s[rng.randrange(len(s))] = rng.choice(extra)
        out.append("".join(s))
    return out


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rng = random.Random(seed)
    bad = 0
    n_pairs = 0
    for case in range(cases):
        extra = rng.choice(["N", "Nn", "acgt", "NRYKM", "-"])
        rate = rng.choice([0.02, 0.1, 0.3, 1.0])
        seqs = []
        for _ in range(rng.randrange(2, 6)):
            L = rng.choice([70, 150, 300, 301, 640, 1300, 2100])
            seqs += family(rng, rng.randrange(3, 25), L, rng.choice([6, 12, 40, 150, 500]), extra, rate)
        seqs = list(dict.fromkeys(seqs))
        st = SeqStore(seqs)
        try:
            m = 250
            a = [rng.randrange(len(seqs)) for _ in range(m)]
            b = [rng.randrange(len(seqs)) for _ in range(m)]
            want = [O.ed_dp(seqs[x], seqs[y]) for x, y in zip(a, b)]
            got = st.ed_pairs(a, b, None).tolist()
            k = [rng.choice([rng.randrange(0, 8), rng.randrange(0, 64), rng.randrange(0, 600), 10 ** 6]) for _ in range(m)]
            gk = st.ed_pairs(a, b, k).tolist()
            n_pairs += 2 * m
            if got != want or gk != [d if d <= kk else -1 for d, kk in zip(want, k)]:
                bad += 1
                print("case %d: pair distances differ (extra %r rate %s)" % (case, extra, rate), flush=True)
        finally:
            st.close()
        S = {"r%d" % i: s for i, s in enumerate(seqs)}
        conv = set(rng.sample(sorted(S), rng.randrange(0, 4)))
        depth = rng.choice([None, None, 3, 11])
        params = Params(1) if depth is None else Params(1, depth)
        if ordered(NNG.compute_nearest_neighbor_graph(S, conv, params)[0]) != ordered(O.compute_nearest_neighbor_graph(S, conv, params)[0]):
            bad += 1
            print("case %d: 1-set graph differs (extra %r rate %s depth %s)" % (case, extra, rate, depth), flush=True)
        keys = sorted(S)
        ck = set(rng.sample(keys, max(1, len(keys) // 6)))
        X = {k2: S[k2] for k2 in keys if k2 not in ck}
        C = {k2: S[k2] for k2 in ck}
        if ordered(NNG.compute_2set_nearest_neighbor_graph(X, C, Params(1))) != ordered(O.compute_2set_nearest_neighbor_graph(X, C, Params(1))):
            bad += 1
            print("case %d: 2-set graph differs (extra %r rate %s)" % (case, extra, rate), flush=True)
        if case % 5 == 4:
            print("  ... %d cases" % (case + 1), flush=True)
    print("stress_alphabet: %d cases, %d pair distances, %d mismatches" % (cases, n_pairs, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
