"""Seeded synthetic read generator for the parity tests and bench.py.

gene = uniform random ACGT; isoform = gene minus a random subset of >= 2 internal "exons" (40-400 bp each,
> IsoCon's min_exon_diff=20, /root/reference/IsoCon:200) plus ceil(0.2 % L) SNVs; read = isoform + i.i.d. per-base
errors with the reference author's split ins 68.75 % / del 25 % / sub 6.25 %
(/root/reference/scrips/estimate_read_depth.py:177-179).  Abundances are geometric (ratio 0.8).
"""
from __future__ import annotations

import math

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

CCS_PROFILE = dict(rate=0.01, ins=0.6875, dele=0.25, sub=0.0625)
ONT_PROFILE = dict(rate=0.06, ins=0.35, dele=0.35, sub=0.30)


def _rand_seq(rng, n):
    return _ACGT[rng.integers(0, 4, size=n)]


def make_isoforms(rng, length, n_isoforms, n_exons=8):
    """Return a list of uint8 arrays (ASCII) whose mean length is roughly `length`."""
    lo, hi = (40, 400) if length >= 1200 else (max(5, length // 30), max(8, length // 8))
    exon_lens = rng.integers(lo, hi + 1, size=n_exons)
    gene_len = int(length + exon_lens.mean() * 3)
    gene = _rand_seq(rng, gene_len)
    # non-overlapping internal exons: split the interior in n_exons slots
    margin = max(10, gene_len // 20)
    slots = np.linspace(margin, gene_len - margin, n_exons + 1).astype(int)
    exons = []
    for e in range(n_exons):
        span = slots[e + 1] - slots[e]
        ln = int(min(exon_lens[e], max(1, span - 2)))
        st = int(slots[e] + rng.integers(0, max(1, span - ln)))
        exons.append((st, st + ln))
    isoforms, seen = [], set()
    while len(isoforms) < n_isoforms:
        k = int(rng.integers(2, min(4, n_exons) + 1))
        drop = tuple(sorted(rng.choice(n_exons, size=k, replace=False).tolist()))
        if drop in seen and len(seen) < math.comb(n_exons, 2):
            continue
        seen.add(drop)
        keep = np.ones(gene_len, dtype=bool)
        for e in drop:
            keep[exons[e][0]:exons[e][1]] = False
        iso = gene[keep].copy()
        n_snv = int(math.ceil(0.002 * length))
        pos = rng.choice(len(iso), size=n_snv, replace=False)
        iso[pos] = _ACGT[(np.searchsorted(_ACGT, iso[pos]) + rng.integers(1, 4, size=n_snv)) % 4]
        isoforms.append(iso)
    return isoforms


def mutate(rng, seq, profile):
    """Apply i.i.d. per-base errors; returns a new uint8 array."""
    n = len(seq)
    u = rng.random(n)
    rate = profile["rate"]
    is_del = u < rate * profile["dele"]
    is_sub = (~is_del) & (u < rate * (profile["dele"] + profile["sub"]))
    is_ins = (~is_del) & (~is_sub) & (u < rate)
    out = seq.copy()
    ns = int(is_sub.sum())
    if ns:
        out[is_sub] = _ACGT[(np.searchsorted(_ACGT, seq[is_sub]) + rng.integers(1, 4, size=ns)) % 4]
    reps = np.ones(n, dtype=np.int64)
    reps[is_del] = 0
    reps[is_ins] = 2
    res = np.repeat(out, reps)
    ni = int(is_ins.sum())
    if ni:
        # the first copy of every duplicated base becomes a random inserted base
        idx = np.cumsum(reps) - reps  # start offset of each source base in `res`
        res[idx[is_ins]] = _rand_seq(rng, ni)
    return res


def make_reads(n_reads, length, n_isoforms, seed, profile=None, families=1, length_range=None):
    """Return (accessions, sequences) -- python str lists.  families > 1 draws one gene per family with its own
    length ~ U(length_range) (config C5)."""
    profile = profile or CCS_PROFILE
    rng = np.random.Generator(np.random.PCG64(seed))
    isoforms = []
    per_family = max(1, n_isoforms // families)
    for _f in range(families):
        L = int(rng.integers(length_range[0], length_range[1] + 1)) if length_range else length
        isoforms.extend(make_isoforms(rng, L, per_family))
    w = 0.8 ** np.arange(len(isoforms))
    w /= w.sum()
    which = rng.choice(len(isoforms), size=n_reads, p=w)
    accs, seqs = [], []
    for r in range(n_reads):
        s = mutate(rng, isoforms[which[r]], profile)
        accs.append("read_%d_iso_%d" % (r, which[r]))
        seqs.append(s.tobytes().decode("ascii"))
    return accs, seqs, [i.tobytes().decode("ascii") for i in isoforms]
