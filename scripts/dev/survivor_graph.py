"""GPU box: the survivor graph of the C3 main pass (pairs whose q-gram bound does not exceed their threshold, final thresholds) --
degrees, and how many match-mask tables the pass needs when a pair may use EITHER endpoint's table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
accs, seqs, _ = synth.make_reads(n_reads, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
best, row_ptr, cols, stats = st.nn_graph()
n = len(seqs)
lens = np.asarray(st.lens).astype(np.int64)
b = np.minimum(np.where(best < 0, 63, best), 63).astype(np.int64)
t0 = time.time()
rp, lb = st.qgram_bound_matrix()
print("matrix: %d bytes in %.1f s" % (len(lb), time.time() - t0), flush=True)
src = []
dst = []
for q in range(n):
    lo, hi = int(rp[q]), int(rp[q + 1])
    if hi == lo:
        continue
    p = np.arange(q + 1, q + 1 + hi - lo)
    k = np.maximum(b[q], b[p])
    ok = (lb[lo:hi] <= k) & ((lens[p] - lens[q]) <= k)
    pp = p[ok]
    src.append(np.full(len(pp), q, np.int64)); dst.append(pp)
src = np.concatenate(src); dst = np.concatenate(dst)
E = len(src)
print("survivor pairs: %d (%.1f per read as the lower index)" % (E, E / n), flush=True)
deg_lo = np.bincount(src, minlength=n)
deg = deg_lo + np.bincount(dst, minlength=n)
def q(x): return np.percentile(x, [0, 10, 25, 50, 75, 90, 99, 100]).astype(int).tolist()
print("pairs per read as the lower index (today's tables): percentiles 0/10/25/50/75/90/99/100", q(deg_lo), "non-empty tables", int((deg_lo > 0).sum()))
print("total degree:", q(deg))
def shares(c, tag):
    c = c[c > 0]
    tot = c.sum()
    print(tag, "tables %d | share of pairs in lists >= X:" % len(c), "  ".join("%d: %.3f (%d lists)" % (x, c[c >= x].sum() / tot, int((c >= x).sum())) for x in (32, 64, 128, 256, 512, 768, 1024, 2048, 4096)), flush=True)
shares(deg_lo, "lower-index lists:")
# orientation: every pair to the endpoint of larger total degree (ties: lower index)
own = np.where(deg[dst] > deg[src], dst, src)
cnt = np.bincount(own, minlength=n)
print("pair -> endpoint of larger degree: non-empty tables %d, pairs per non-empty table %s, share of pairs in tables >= 512: %.2f, >= 1024: %.2f"
      % (int((cnt > 0).sum()), q(cnt[cnt > 0]), cnt[cnt >= 512].sum() / E, cnt[cnt >= 1024].sum() / E))
shares(cnt, "larger-degree lists:")
# greedy cover in descending degree order: a read takes all its still unassigned pairs
order = np.argsort(-deg, kind="stable")
rank = np.empty(n, np.int64); rank[order] = np.arange(n)
own2 = np.where(rank[dst] < rank[src], dst, src)
cnt2 = np.bincount(own2, minlength=n)
print("pair -> endpoint earlier in the degree order (same thing with ties by order): non-empty %d, %s" % (int((cnt2 > 0).sum()), q(cnt2[cnt2 > 0])))
# how long do pairs run?  sample true distances of survivors
rng = np.random.default_rng(3)
idx = rng.choice(E, 20000, replace=False)
d = st.ed_pairs(src[idx].astype(np.uint32), dst[idx].astype(np.uint32), None)
k = np.maximum(b[src[idx]], b[dst[idx]])
frac = np.minimum(1.0, (k + 1.0) / np.maximum(d, 1))
print("sampled survivors: share with d <= k %.3f, modelled share of columns run %.3f" % (float((d <= k).mean()), float(frac.mean())))
