"""GPU box: how many of the main pass' pairs (C3, final bounds) would a q-gram count bound reject, and what share of the
DP columns (modelled: a pair at distance d with threshold k runs min(1, (k+1)/d) of its columns)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore

accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
best, row_ptr, cols, stats = st.nn_graph()
lens = np.asarray(st.lens).astype(np.int64)
b = np.minimum(best, 63).astype(np.int64)
code = np.zeros(256, np.int64); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3


def profile(s, q, bins):
    c = code[np.frombuffer(s.encode(), np.uint8)]
    v = np.zeros(len(c) - q + 1, np.int64)
    for i in range(q):
        v = v * 4 + c[i:len(c) - q + 1 + i]
    if 4 ** q > bins:
        v = (v * 2654435761 >> 7) % bins
    return np.minimum(np.bincount(v, minlength=bins), 255).astype(np.int16)


rng = np.random.default_rng(1)
sample = rng.choice(len(seqs), 120, replace=False)
for q, bins in ((8, 4096), (8, 6144), (8, 8192), (8, 12288), (9, 8192), (10, 8192)):
    t0 = time.time()
    cache = {}
    def P(i):
        if i not in cache:
            cache[i] = profile(seqs[i], q, bins)
        return cache[i]
    tot = rej = 0
    cols_all = cols_rej = 0.0
    for i in sample:
        hi = np.searchsorted(lens, lens[i] + 63, "right")
        j = np.arange(i + 1, hi)
        if len(j) > 1500:
            j = np.sort(rng.choice(j, 1500, replace=False))
        k = np.maximum(b[i], b[j])
        ok = (lens[j] - lens[i]) <= k
        j, k = j[ok], k[ok]
        if not len(j):
            continue
        pi = P(i)
        M = np.stack([P(int(x)) for x in j])
        l1 = np.abs(M - pi[None, :]).sum(axis=1)
        ds = np.abs(M.sum(axis=1) - pi.sum())
        lb = -(-(l1 + ds) // (2 * q))
        d = st.ed_pairs(np.full(len(j), i, np.uint32), j.astype(np.uint32), None)
        assert (lb <= d).all()
        colsrun = lens[j] * np.minimum(1.0, (k + 1.0) / np.maximum(d, 1))
        r = lb > k
        tot += len(j); rej += int(r.sum()); cols_all += colsrun.sum(); cols_rej += colsrun[r].sum()
    print("q=%d bins=%d: %d admitted pairs sampled, rejected %.1f %%, modelled DP columns removed %.1f %%  (%.0f s)"
          % (q, bins, tot, 100.0 * rej / tot, 100.0 * cols_rej / cols_all, time.time() - t0), flush=True)
