"""GPU box + host cores: what a SECOND-stage rejection test would buy on the pairs that survive the q-gram bound of the C3 main pass
(VERDICT r5 item 1).  Survivors = window pairs whose stored bound does not exceed max(threshold of either end) (final thresholds); a
random sample gets its exact distance from the GPU and, on the host, every candidate bound:

  exact q     un-hashed multiset q-gram lemma, ceil((max(|A|,|B|) - sum min(A,B)) / q), q in 7, 9, 11, 13, and their maximum
  hash2       the product's hashed presence + capped-excess bound under a second, independent hash; max with the stored bound
  blocks b/T  DISJOINT blocks of length b of one sequence looked up in the set of all b-grams of the other: every edit destroys at
              most one block of a tiling (a substituted / deleted base lies in one block, an insertion between two bases of one
              block), an undestroyed block occurs in the other sequence, so #(blocks that occur nowhere) <= ed; max over both
              directions and over T shifted tilings

and reports, per bound, the share of the sampled survivors it rejects at the pair's own threshold (bound > k), split by hit / non-hit
(a hit must never be rejected: asserted)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from isocon_amd.store import SeqStore

mode = sys.argv[1]            # "gpu": survivors + sample + exact distances -> npz;  "cpu": the bounds on the sample (Pool over the host cores)
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
n_sample = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
length = int(sys.argv[4]) if len(sys.argv) > 4 else 2500
NPZ = os.environ.get("STUDY_NPZ", "gpurun_out/second_stage_sample.npz")
accs, seqs, _ = synth.make_reads(n_reads, length, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
if mode == "cpu":
    z = np.load(NPZ)
    sa, sb, slb, d, k = z["sa"], z["sb"], z["slb"], z["d"], z["k"]
    idx = np.arange(len(sa))
    hit = d <= k
else:
  st = SeqStore(seqs)
  best, row_ptr, cols, stats = st.nn_graph()
  n = len(seqs)
  lens = np.asarray(st.lens).astype(np.int64)
  b = np.minimum(np.where(best < 0, 63, best), 63).astype(np.int64)
  rp, lb = st.qgram_bound_matrix()
  src = []; dst = []; lbs = []
  for q in range(n):
      lo, hi = int(rp[q]), int(rp[q + 1])
      if hi == lo:
          continue
      p = np.arange(q + 1, q + 1 + hi - lo)
      k = np.maximum(b[q], b[p])
      ok = (lb[lo:hi] <= k) & ((lens[p] - lens[q]) <= k)
      src.append(np.full(int(ok.sum()), q, np.int64)); dst.append(p[ok]); lbs.append(lb[lo:hi][ok])
  src = np.concatenate(src); dst = np.concatenate(dst); lbs = np.concatenate(lbs).astype(np.int64)
  E = len(src)
  print("reads %d, window pairs %d, survivors %d (%.1f per read)" % (n, len(lb), E, E / n), flush=True)
  rng = np.random.default_rng(11)
  idx = np.sort(rng.choice(E, min(n_sample, E), replace=False))
  sa, sb, slb = src[idx], dst[idx], lbs[idx]
  d = st.ed_pairs(sa.astype(np.uint32), sb.astype(np.uint32), None).astype(np.int64)
  k = np.maximum(b[sa], b[sb])
  hit = d <= k
  print("sample %d: hits %.4f, stored bound / d: mean %.3f (10/50/90 %%: %s); d / k of the non-hits 10/50/90 %%: %s; k 10/50/90 %%: %s" % (
      len(idx), hit.mean(), (slb / np.maximum(d, 1)).mean(), np.round(np.percentile(slb / np.maximum(d, 1), [10, 50, 90]), 3).tolist(),
      np.round(np.percentile((d / k)[~hit], [10, 50, 90]), 3).tolist(), np.percentile(k, [10, 50, 90]).tolist()), flush=True)

  np.savez(NPZ, sa=sa, sb=sb, slb=slb, d=d, k=k)
  sys.exit(0)

_CODE = np.zeros(256, np.int64); _CODE[ord("C")] = 1; _CODE[ord("G")] = 2; _CODE[ord("T")] = 3
_base = {}
def base(i):
    if i not in _base:
        _base[i] = _CODE[np.frombuffer(seqs[i].encode(), np.uint8)]
    return _base[i]
_codes = {}
def codes(i, q):
    """base-4 code of every q-gram of read i, by position"""
    key = (i, q)
    if key not in _codes:
        c = base(i); ng = len(c) - q + 1
        v = np.zeros(ng, np.int64)
        for j in range(q):
            v = v * 4 + c[j:j + ng]
        _codes[key] = v
    return _codes[key]
_uniq = {}
def uniq(i, q):
    key = (i, q)
    if key not in _uniq:
        _uniq[key] = np.unique(codes(i, q), return_counts=True)
    return _uniq[key]

def exact_bound(x, y, q):
    ux, cx = uniq(x, q); uy, cy = uniq(y, q)
    _, ix, iy = np.intersect1d(ux, uy, assume_unique=True, return_indices=True)
    m = int(np.minimum(cx[ix], cy[iy]).sum())
    return (max(int(cx.sum()), int(cy.sum())) - m + q - 1) // q

def unmatched_blocks(x, y, bl, shift):
    """blocks of y (length bl, tiling shifted by `shift`) that occur nowhere in x"""
    blocks = codes(y, bl)[shift::bl]
    ux = uniq(x, bl)[0]
    pos = np.searchsorted(ux, blocks)
    pos[pos == len(ux)] = 0
    return int((ux[pos] != blocks).sum())

def block_bound(x, y, bl, shifts, both=True):
    u = max(unmatched_blocks(x, y, bl, s) for s in shifts)
    if both:
        u = max(u, max(unmatched_blocks(y, x, bl, s) for s in shifts))
    return u

Q, B0, B1, CAP = 9, 24576, 2048, 2
_prof = {}
def hashed_profile(i, mult):
    key = (i, mult)
    if key not in _prof:
        g = (((codes(i, Q) * mult) & 0xffffffff) >> 7) % B0
        cnt = np.bincount(g, minlength=B0)
        ex = np.zeros(B1, np.int64); np.add.at(ex, np.arange(B0) % B1, np.maximum(cnt - 1, 0))
        _prof[key] = np.concatenate([(cnt > 0).astype(np.int64), np.minimum(ex, CAP)])
    return _prof[key]
def hashed_bound(x, y, mult):
    pa, pb = hashed_profile(x, mult), hashed_profile(y, mult)
    return (max(int(pa.sum()), int(pb.sum())) - int(np.minimum(pa, pb).sum()) + Q - 1) // Q

tests = {}
def add(name, fn):
    tests[name] = fn
for q in (7, 9, 11, 13):
    add("exact q=%d" % q, lambda x, y, q=q: exact_bound(x, y, q))
add("hash2 (9-grams, second hash)", lambda x, y: hashed_bound(x, y, 0x85EBCA6B))
for bl in (8, 9, 10, 12):
    add("blocks b=%d one tiling, one direction" % bl, lambda x, y, bl=bl: block_bound(x, y, bl, (0,), False))
    add("blocks b=%d one tiling, both directions" % bl, lambda x, y, bl=bl: block_bound(x, y, bl, (0,), True))
    add("blocks b=%d 3 tilings, both directions" % bl, lambda x, y, bl=bl: block_bound(x, y, bl, (0, bl // 3, (2 * bl) // 3), True))
    add("blocks b=%d all tilings, both directions" % bl, lambda x, y, bl=bl: block_bound(x, y, bl, tuple(range(bl)), True))

def work(part):
    out = np.zeros((len(part), len(tests)), np.int64)
    for r, j in enumerate(part):
        x, y = int(sa[j]), int(sb[j])
        for c, fn in enumerate(tests.values()):
            out[r, c] = fn(x, y)
    return out

t0 = time.time()
import multiprocessing as mp
cores = int(os.environ.get("STUDY_CORES", "14"))
parts = np.array_split(np.arange(len(idx)), cores * 8)          # (sorted by the lower index: a part reuses its reads' tables)
with mp.Pool(cores) as pool:
    outs = pool.map(work, parts)
allr = np.concatenate(outs)
res = {name: allr[:, c] for c, name in enumerate(tests)}
print("host bounds: %.0f s on %d cores" % (time.time() - t0, cores), flush=True)
res["max over exact q in 7, 9, 11, 13"] = np.max([res["exact q=%d" % q] for q in (7, 9, 11, 13)], axis=0)
res["max(stored, hash2)"] = np.maximum(slb, res["hash2 (9-grams, second hash)"])
res["stored bound (first stage)"] = slb
nh = ~hit
print("\n%-46s %9s %9s %9s %11s" % ("bound", "rejected", "of non-h.", "hits rej.", "bound / d"))
for name, v in res.items():
    assert (v <= d).all(), "%s is not a lower bound" % name
    rej = v > k
    print("%-46s %9.3f %9.3f %9d %11.3f" % (name, rej.mean(), rej[nh].mean(), int(rej[hit].sum()), (v / np.maximum(d, 1)).mean()))
# how the rejected share depends on the threshold class
print("\nby threshold (blocks b=9, 3 tilings): k <= 31: %.3f of %d, k > 31: %.3f of %d" % (
    (res["blocks b=9 3 tilings, both directions"] > k)[k <= 31].mean(), int((k <= 31).sum()),
    (res["blocks b=9 3 tilings, both directions"] > k)[k > 31].mean(), int((k > 31).sum())))
