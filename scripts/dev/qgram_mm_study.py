"""GPU box: how many of the main pass' pairs (C3, final thresholds) does the q-gram count bound reject as a function of the
gram length, the number of hashed bins and the cap of the stored counts?  (The thermometer-code contraction of
csrc/qgram_mm.hpp costs in proportion to bins x cap.)  Profiles and bounds are computed with torch on the GPU; nothing here is
product code."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
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
dev = torch.device("cuda:0")
L = int(lens.max())
code = np.zeros(256, np.uint8); code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
C = np.zeros((n, L), np.uint8)
for i, s in enumerate(seqs):
    C[i, :len(s)] = code[np.frombuffer(s.encode(), np.uint8)]
Cd = torch.from_numpy(C).to(dev).to(torch.int64)
lens_d = torch.from_numpy(lens).to(dev)
b_d = torch.from_numpy(b).to(dev)

rng = np.random.default_rng(1)
sample = np.sort(rng.choice(n, 160, replace=False))
hi_of = np.searchsorted(lens, lens + 63, "right")


def profiles(q, bins):
    g = torch.zeros((n, L - q + 1), dtype=torch.int64, device=dev)
    # the kernel's gram index: low code bits of the q bases | high code bits << q
    for i in range(q):
        c = Cd[:, i:L - q + 1 + i]
        g |= ((c & 1) << i) | ((c >> 1) << (q + i))
    if 4 ** q > bins:
        g = (((g * 0x9E3779B1) & 0xFFFFFFFF) >> 7) % bins
    valid = torch.arange(L - q + 1, device=dev)[None, :] < (lens_d[:, None] - q + 1)
    P = torch.zeros((n, bins), dtype=torch.int16, device=dev)
    P.scatter_add_(1, g, valid.to(torch.int16))
    return P


print("n = %d unique reads, %d sampled queries" % (n, len(sample)), flush=True)
for q, bins in ((8, 6144), (8, 16384), (9, 16384), (9, 24576)):
    t0 = time.time()
    P = profiles(q, bins)
    res = {}
    for cap in (1, 2, 3, 4, 255):
        Pc = torch.clamp(P, max=cap)
        sums = Pc.sum(1, dtype=torch.int64)
        tot = rej = 0
        for i in sample:
            hi = int(hi_of[i])
            if hi <= i + 1:
                continue
            j = torch.arange(i + 1, hi, device=dev)
            k = torch.maximum(b_d[i], b_d[j])
            ok = (lens_d[j] - lens_d[i]) <= k
            j, k = j[ok], k[ok]
            if j.numel() == 0:
                continue
            M = torch.minimum(Pc[j], Pc[i][None, :]).sum(1, dtype=torch.int64)
            lb = (torch.maximum(sums[j], sums[i]) - M + q - 1) // q
            tot += int(j.numel()); rej += int((lb > k).sum())
        res[cap] = (tot, rej)
    occ = float((P > 0).sum()) / P.numel()
    print("q=%2d bins=%5d occupancy %.3f | " % (q, bins, occ) + "  ".join(
        "cap %s: %.2f %% rejected (%.0f survive / query)" % (c if c < 255 else "inf", 100.0 * r / t, (t - r) / len(sample)) for c, (t, r) in res.items())
        + "  (%.0f s)" % (time.time() - t0), flush=True)
    del P

# Mixed designs: presence bits of B0 fine bins + the EXCESS counts (a - 1)^+ merged into B1 = B0 / r coarser bins, capped at c
# (csrc/qgram_mm.hpp: the split a -> ([a > 0], (a - 1)^+) keeps S+ exactly, merging / capping each part only shrinks it).
print("mixed designs: K = B0 + B1 * cap", flush=True)
for q, B0, B1, cap in ((8, 12288, 3072, 2), (8, 16384, 2048, 2), (8, 16384, 4096, 2), (8, 16384, 4096, 3), (8, 16384, 8192, 2), (9, 16384, 4096, 2), (9, 16384, 2048, 2),
                       (9, 12288, 3072, 2), (8, 24576, 4096, 2), (9, 24576, 4096, 2), (9, 24576, 2048, 2), (8, 32768, 4096, 2), (9, 32768, 4096, 2), (10, 24576, 4096, 2)):
    t0 = time.time()
    P = profiles(q, B0)
    pres = (P > 0).to(torch.int16)
    ex = torch.clamp(P - 1, min=0)
    exc = torch.zeros((n, B1), dtype=torch.int16, device=dev)
    exc.scatter_add_(1, (torch.arange(B0, device=dev) % B1)[None, :].expand(n, B0), ex)
    exc = torch.clamp(exc, max=cap)
    V = torch.cat([pres, exc], 1)
    sums = V.sum(1, dtype=torch.int64)
    tot = rej = 0
    for i in sample:
        hi = int(hi_of[i])
        if hi <= i + 1:
            continue
        j = torch.arange(i + 1, hi, device=dev)
        k = torch.maximum(b_d[i], b_d[j])
        ok = (lens_d[j] - lens_d[i]) <= k
        j, k = j[ok], k[ok]
        if j.numel() == 0:
            continue
        M = torch.minimum(V[j], V[i][None, :]).sum(1, dtype=torch.int64)
        lb = (torch.maximum(sums[j], sums[i]) - M + q - 1) // q
        tot += int(j.numel()); rej += int((lb > k).sum())
    print("q=%2d B0=%5d B1=%5d cap %d  K=%5d: %.2f %% rejected (%.0f survive / query)  (%.0f s)" % (q, B0, B1, cap, B0 + B1 * cap, 100.0 * rej / tot, (tot - rej) / len(sample), time.time() - t0), flush=True)
    del P, pres, ex, exc, V
