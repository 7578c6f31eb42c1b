"""g19: the WHOLE 2-set nearest-neighbour graph (reads x candidates, NNG:201-234, 300-424 -- the search the metric is named after) of a
BASELINE configuration's reads against a seeded candidate set, every read's row recomputed on the CPU with the oracle's statement of the
reference loop (oracle.nn_2set = NNG:341-424, no GPU anywhere), so that a -m gpu test compares all 50 000 (C3) / 5 000 (C2) rows.

    python tests/golden/make_golden_g19.py c2|c3 [cores]        ->  tests/golden/g19_<cfg>_graph_2set.npz

Candidates (candidates(which) below, the function the test imports): the configuration's true isoforms, per isoform a number of variants with
1-4 random edits (what converged consensus sequences look like: a few bases from an isoform), and a few reads themselves (distance 0 is
admitted in the 2-set search, NNG:388).  Arrays: over the MERGED list sorted(reads + candidates, key=len) (reads first, stable: NNG:202-208)
is_target uint8[n], best int32[n] (-1 = no admissible candidate; target rows -1), row_ptr int64[n+1], cols uint32[] (positions in the merged
list, in the reference's insertion order).  `inputs_sha1` pins the merged list."""
import hashlib
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from isocon_amd import synth  # noqa: E402

CONFIGS = {"c2": (5000, 1500, 3, 20001), "c3": (50000, 2500, 10, 30001)}
VARIANTS = {"c2": 60, "c3": 100}          # per isoform
G = {}


def candidates(which):
    """(X, C): {read_acc: seq}, {cand_acc: seq} -- deterministic"""
    accs, seqs, isoforms = synth.make_reads(*CONFIGS[which])
    rng = np.random.Generator(np.random.PCG64(CONFIGS[which][3] + 19))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    C, seen = {}, set()
    for i, iso in enumerate(isoforms):
        base = np.frombuffer(iso.encode(), dtype=np.uint8)
        C["iso%d" % i] = iso
        seen.add(iso)
        for v in range(VARIANTS[which]):
            s = base.copy()
            for _ in range(int(rng.integers(1, 5))):
                p = int(rng.integers(0, len(s)))
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    s[p] = acgt[(int(np.searchsorted(acgt, s[p])) + int(rng.integers(1, 4))) % 4]
                elif kind == 1:
                    s = np.concatenate([s[:p], acgt[rng.integers(0, 4, size=1)], s[p:]])
                else:
                    s = np.concatenate([s[:p], s[p + 1:]])
            t = s.tobytes().decode()
            if t not in seen:
                seen.add(t)
                C["iso%d_v%d" % (i, v)] = t
    for r in rng.choice(len(seqs), size=20, replace=False).tolist():          # candidates that ARE reads: distance 0
        if seqs[r] not in seen:
            seen.add(seqs[r])
            C["as_read_%d" % r] = seqs[r]
    return dict(zip(accs, seqs)), C


def merged_list(X, C):
    """NNG:202-208: [(seq, acc)] reads first, then candidates, stable sort by length"""
    return sorted([(s, a) for a, s in X.items()] + [(s, a) for a, s in C.items()], key=lambda t: len(t[0]))


def inputs_sha1(merged):
    h = hashlib.sha1()
    for s, a in merged:
        h.update(s.encode()); h.update(b"\t"); h.update(a.encode()); h.update(b"\n")
    return h.hexdigest()


def rows(lo_hi):
    from oracle import oracle as O
    lo, hi = lo_hi
    rp, c, e, calls = O.nn_2set(G["seqs"], G["is_t"], lo, hi - lo)
    out = []
    for r in range(hi - lo):
        b, t = int(rp[r]), int(rp[r + 1])
        out.append((c[b:t].astype(np.uint32), int(e[b]) if t > b else -1))
    return lo, out, calls


if __name__ == "__main__":
    from oracle import oracle as O
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    cores = int(sys.argv[2]) if len(sys.argv) > 2 else len(os.sched_getaffinity(0))
    O.build()
    X, C = candidates(which)
    merged = merged_list(X, C)
    n = len(merged)
    is_t = np.fromiter((1 if a in C else 0 for _, a in merged), dtype=np.uint8, count=n)
    G.update(seqs=[s for s, _ in merged], is_t=is_t)
    t0 = time.time()
    step = 100
    jobs = [(lo, min(lo + step, n)) for lo in range(0, n, step)]
    res, calls = {}, 0
    with Pool(cores) as pool:
        for k, (lo, out, c) in enumerate(pool.imap_unordered(rows, jobs)):
            res[lo] = out
            calls += c
            if k % 50 == 0:
                print("%d / %d rows, %.0f s" % (k * step, n, time.time() - t0), flush=True)
    best = np.empty(n, np.int32)
    row_ptr = np.zeros(n + 1, np.int64)
    cols = []
    for lo, _ in jobs:
        for j, (c, d) in enumerate(res[lo]):
            i = lo + j
            best[i] = d
            row_ptr[i + 1] = row_ptr[i] + len(c)
            cols.append(c)
    cols = np.concatenate(cols) if cols else np.zeros(0, np.uint32)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g19_%s_graph_2set.npz" % which)
    np.savez_compressed(path, is_target=is_t, best=best, row_ptr=row_ptr, cols=cols, inputs_sha1=np.array(inputs_sha1(merged)),
                        config=np.array(repr((CONFIGS[which], VARIANTS[which]))), edlib_calls=np.array(calls))
    print("%s: %d entries (%d candidates), %d edges, %d alignments of the reference loop, %.0f s on %d cores -> %s (%d bytes)" %
          (which, n, int(is_t.sum()), len(cols), calls, time.time() - t0, cores, path, os.path.getsize(path)))
