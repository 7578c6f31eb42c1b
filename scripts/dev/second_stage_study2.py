"""Host only, on the sample scripts/dev/second_stage_study.py saved: the GREEDY form of the block bound.  found[p] = "the b-gram of y at p
occurs somewhere in x"; every b-gram that is not found holds an edited position, so ed >= the smallest set of positions that hits all
unfound intervals [p, p + b) = (intervals of equal length) the greedy count: leftmost unfound p, count, continue at p + b.  It dominates
the maximum over all shifted tilings.  Variants: one direction (blocks of the PARTNER in the set of the OWNER: all a table kernel's
workgroup has in LDS is its owner's set) / both; the set as a hashed bitmap of 2^bits bits (false positives only weaken the bound);
the cheap search: stride-b probes, a binary search for the leftmost unfound gram behind an unfound probe."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth

NPZ = os.environ.get("STUDY_NPZ", "gpurun_out/second_stage_sample.npz")
n_use = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
z = np.load(NPZ)
sel = np.sort(np.random.default_rng(5).choice(len(z["sa"]), n_use, replace=False))
sa, sb, slb, d, k = (z[t][sel] for t in ("sa", "sb", "slb", "d", "k"))
hit = d <= k
_CODE = np.zeros(256, np.int64); _CODE[ord("C")] = 1; _CODE[ord("G")] = 2; _CODE[ord("T")] = 3
_codes = {}
def codes(i, q):
    key = (i, q)
    if key not in _codes:
        c = _CODE[np.frombuffer(seqs[i].encode(), np.uint8)]; ng = len(c) - q + 1
        v = np.zeros(ng, np.int64)
        for j in range(q):
            v = v * 4 + c[j:j + ng]
        _codes[key] = v
    return _codes[key]
def fold(v, bits):
    if bits is None:
        return v
    return (v ^ (v >> bits) ^ (v >> (2 * bits))) & ((1 << bits) - 1)
_sets = {}
def gset(i, q, bits):
    key = (i, q, bits)
    if key not in _sets:
        _sets[key] = np.unique(fold(codes(i, q), bits))
    return _sets[key]
def found_vec(x, y, b, bits):
    """found[p] for every b-gram of y, looked up in x's set"""
    g = fold(codes(y, b), bits); ux = gset(x, b, bits)
    pos = np.searchsorted(ux, g); pos[pos == len(ux)] = 0
    return ux[pos] == g
def greedy(f, b):
    un = np.flatnonzero(~f)
    cnt = 0; nxt = 0; lookups = 0
    for p in un:
        if p >= nxt:
            cnt += 1; nxt = p + b
    return cnt
def greedy_grid(f, b, s):
    """greedy on the grams at positions 0, s, 2 s, ... only (b a multiple of s): leftmost unfound probed gram, count, continue at p + b"""
    un = np.flatnonzero(~f[::s]) * s
    cnt = 0; nxt = 0
    for p in un:
        if p >= nxt:
            cnt += 1; nxt = p + b
    return cnt
def probe_search(f, b):
    """stride-b probes; behind an unfound probe a binary search (assuming found ... found unfound ... unfound between the last jump and the probe)
    for the leftmost unfound gram; returns (count, lookups).  Valid whatever the search finds: the counted grams are unfound and disjoint."""
    n = len(f); i = 0; start = 0; cnt = 0; look = 0
    while i < n:
        look += 1
        if f[i]:
            i += b; continue
        lo, hi = max(start, i - b + 1), i          # hi unfound
        while lo < hi:
            mid = (lo + hi) // 2; look += 1
            if f[mid]: lo = mid + 1
            else: hi = mid
        cnt += 1; start = hi + b; i = start
    return cnt, look

res = {}; looks = {}
t0 = time.time()
variants = []
for b, st in ((8, 1), (8, 2), (8, 4), (12, 1), (12, 2), (12, 4), (12, 6), (16, 4), (16, 8)):
    variants.append(("greedy b=%d on a grid of stride %d, exact / 2^16-bit set" % (b, st), b, None if b == 8 else 16, ("grid", st)))
for name, b, bits, kind in variants:
    out = np.zeros(len(sa), np.int64); lk = 0
    for j in range(len(sa)):
        x, y = int(sa[j]), int(sb[j])
        # owner = either end in the product (the hub); here: the lower index is the "owner" -- symmetric on average
        f = found_vec(x, y, b, bits)
        if isinstance(kind, tuple):
            out[j] = greedy_grid(f, b, kind[1])
        elif kind == "g1":
            out[j] = greedy(f, b)
        elif kind == "g2":
            out[j] = max(greedy(f, b), greedy(found_vec(y, x, b, bits), b))
        else:
            out[j], l = probe_search(f, b); lk += l
    res[name] = out; looks[name] = lk / len(sa)
    print("  %s: %.0f s" % (name, time.time() - t0), flush=True)
nh = ~hit
print("\nsample %d pairs (hits %d)" % (len(sa), int(hit.sum())))
print("%-56s %9s %9s %9s %10s %9s" % ("bound", "rejected", "of non-h.", "hits rej.", "bound / d", "lookups"))
for name, v in res.items():
    assert (v <= d).all(), "%s is not a lower bound" % name
    rej = v > k
    print("%-56s %9.3f %9.3f %9d %10.3f %9s" % (name, rej.mean(), rej[nh].mean(), int(rej[hit].sum()), (v / np.maximum(d, 1)).mean(), ("%.0f" % looks[name]) if looks[name] else "-"))
