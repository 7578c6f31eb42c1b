"""How tight is the q-gram count bound  ed(a,b) >= L1(profile_a, profile_b) / (2q)  on C3's pairs?  (CPU study: distances from the
oracle, thresholds approximated by the nearest neighbour inside the sampled window.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from isocon_amd import synth
from oracle import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
accs, seqs, _ = synth.make_reads(N, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
lens = np.array([len(s) for s in seqs])
rng = np.random.default_rng(3)
code = np.zeros(256, np.int64); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3


def profile(s, q, bins):
    c = code[np.frombuffer(s.encode(), np.uint8)]
    v = np.zeros(len(c) - q + 1, np.int64)
    for i in range(q):
        v = v * 4 + c[i:len(c) - q + 1 + i]
    if 4 ** q > bins:
        v = (v * 2654435761 >> 7) % bins
    return np.minimum(np.bincount(v, minlength=bins), 255)


qi = rng.choice(len(seqs), 40, replace=False)
rows = []
for i in qi:
    lo = np.searchsorted(lens, lens[i] - 63, "left"); hi = np.searchsorted(lens, lens[i] + 63, "right")
    js = rng.choice(np.arange(lo, hi), min(150, hi - lo), replace=False)
    js = js[js != i]
    d = oracle.ed_pairs(seqs, np.full(len(js), i), js, None)
    for j, dd in zip(js, d):
        rows.append((i, j, int(dd)))
rows = np.array(rows)
print("pairs:", len(rows), "distance percentiles 5/25/50/75/95:", np.percentile(rows[:, 2], [5, 25, 50, 75, 95]))
for q, bins in ((4, 256), (5, 1024), (6, 4096), (8, 4096), (8, 16384), (10, 16384), (12, 16384)):
    prof = {}
    for idx in set(rows[:, 0]) | set(rows[:, 1]):
        prof[idx] = profile(seqs[idx], q, bins)
    lb = np.array([np.abs(prof[i] - prof[j]).sum() for i, j, _ in rows]) / (2.0 * q)
    lb = np.ceil(lb - 1e-9)
    d = rows[:, 2]
    assert (lb <= d).all(), "not a lower bound?!"
    near = d <= 80
    print("q=%2d bins=%5d  LB/d median %.2f (pairs with d<=80: %.2f);  of pairs with 36<d<=80: LB>32: %.2f  LB>36: %.2f  LB>40: %.2f ; of d>80: LB>40: %.2f"
          % (q, bins, np.median(lb / np.maximum(d, 1)), np.median((lb / np.maximum(d, 1))[near]),
             (lb[near & (d > 36)] > 32).mean(), (lb[near & (d > 36)] > 36).mean(), (lb[near & (d > 36)] > 40).mean(), (lb[~near] > 40).mean()))

print("presence bitsets (min(count, 1)):")
for q, bins in ((6, 4096), (7, 8192), (7, 16384), (8, 8192), (8, 16384), (8, 32768), (10, 32768), (9, 16384)):
    prof = {}
    for idx in set(rows[:, 0]) | set(rows[:, 1]):
        prof[idx] = np.minimum(profile(seqs[idx], q, bins), 1)
    lb = np.array([np.abs(prof[i] - prof[j]).sum() for i, j, _ in rows]) / (2.0 * q)
    lb = np.ceil(lb - 1e-9)
    d = rows[:, 2]
    assert (lb <= d).all(), "not a lower bound?!"
    near = d <= 80
    print("q=%2d bits=%5d  LB/d median (d<=80) %.2f;  of pairs with 36<d<=80: LB>32: %.2f  LB>36: %.2f  LB>40: %.2f ; of d>80: LB>40: %.2f  LB>63: %.2f"
          % (q, bins, np.median((lb / np.maximum(d, 1))[near]),
             (lb[near & (d > 36)] > 32).mean(), (lb[near & (d > 36)] > 36).mean(), (lb[near & (d > 36)] > 40).mean(), (lb[~near] > 40).mean(), (lb[~near] > 63).mean()))

print("capped counts (thermometer-coded bit planes, XOR + popcount):")
for cap in (255, 4, 3, 2, 1):
    prof = {}
    for idx in set(rows[:, 0]) | set(rows[:, 1]):
        prof[idx] = np.minimum(profile(seqs[idx], 6, 4096), cap)
    lb = np.array([np.abs(prof[i] - prof[j]).sum() + abs(int(prof[i].sum()) - int(prof[j].sum())) for i, j, _ in rows]) / 12.0
    lb = np.ceil(lb - 1e-9)
    d = rows[:, 2]
    assert (lb <= d).all()
    near = d <= 80
    print("cap=%3d  LB/d median (d<=80) %.3f;  of pairs with 36<d<=80: LB>32: %.2f  LB>36: %.2f  LB>40: %.2f" %
          (cap, np.median((lb / np.maximum(d, 1))[near]), (lb[near & (d > 36)] > 32).mean(), (lb[near & (d > 36)] > 36).mean(), (lb[near & (d > 36)] > 40).mean()))
