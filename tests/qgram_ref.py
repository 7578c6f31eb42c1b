"""numpy restatement of the q-gram bound of isocon_amd/csrc/qgram_mm.hpp (test infrastructure): the stored vector of a sequence is
the presence bits of its q-grams hashed into B0 bins followed by the excess counts (a - 1)^+ merged into B1 bins (bin mod B1)
and capped at CAP; bound = ceil((max(|A|, |B|) - sum min(A, B)) / q)."""
import numpy as np



def _params():
    """the library's own constants (it loads without a GPU); the values of this round as a fallback"""
    try:
        import ctypes
        from isocon_amd import _lib
        out = (ctypes.c_int32 * 4)()
        _lib.load().isocon_qgram_params(out)
        return tuple(int(v) for v in out)
    except Exception:
        return 9, 12288, 2048, 2


Q, B0, B1, CAP = _params()

_CODE = np.zeros(256, np.int64)
_CODE[ord("C")] = 1
_CODE[ord("G")] = 2
_CODE[ord("T")] = 3


def gram_bins(s, q=Q, b0=B0):
    """bin of every q-gram, indexed like the kernel: low code bits of the q bases | high code bits << q, then hashed"""
    c = _CODE[np.frombuffer(s.encode(), np.uint8)]
    ng = len(c) - q + 1
    if ng <= 0:
        return np.zeros(0, np.int64)
    idx = np.zeros(ng, np.int64)
    for i in range(q):
        idx |= (c[i:i + ng] & 1) << i
        idx |= (c[i:i + ng] >> 1) << (q + i)
    if b0 != 4 ** q:
        idx = (((idx * 0x9E3779B1) & 0xffffffff) >> 7) % b0
    return idx


def profile(s, q=Q, b0=B0, b1=B1, cap=CAP):
    cnt = np.bincount(gram_bins(s, q, b0), minlength=b0)
    pres = (cnt > 0).astype(np.int64)
    ex = np.zeros(b1, np.int64)
    np.add.at(ex, np.arange(b0) % b1, np.maximum(cnt - 1, 0))
    return np.concatenate([pres, np.minimum(ex, cap)])


def bound(pa, pb, q=Q):
    return int((max(int(pa.sum()), int(pb.sum())) - int(np.minimum(pa, pb).sum()) + q - 1) // q)


def thermometer(p, b0=B0, cap=CAP):
    """binary form of a stored vector: presence bits, then the levels [excess > t], t < cap (min(a, b) = sum_t [a > t][b > t])"""
    return np.concatenate([p[:b0]] + [(p[b0:] > t).astype(np.int64) for t in range(cap)]).astype(np.float32)


def bound_matrix(seqs, q=Q):
    """all-pairs bounds (n x n, int64) through one float32 product of the thermometer codes -- exact, the sums stay below 2^24"""
    T = np.stack([thermometer(profile(s)) for s in seqs])
    sums = T.sum(axis=1).astype(np.int64)
    M = (T @ T.T).astype(np.int64)
    return (np.maximum(sums[:, None], sums[None, :]) - M + q - 1) // q


# ---- the block bound behind it (isocon_amd/csrc/nn_filter.hpp) -----------------------------------------------------------------------

def block_count(owner, partner, b=8, s=4):
    """greedy number of pairwise disjoint b-grams of `partner`, probed at the positions 0, s, 2 s, ... of the whole 16-base words
    all of whose grams lie inside the sequence (word j counts when 16 j + 16 - s + b <= len; the kernel: b = 8), which occur nowhere in `owner`: a counted gram skips the probes that overlap it"""
    grams = set(owner[i:i + b] for i in range(len(owner) - b + 1))
    n = len(partner)
    nd = (n - (16 - s + b)) // 16 + 1 if n >= 16 - s + b else 0
    cnt, nxt = 0, 0
    for p in range(0, 16 * nd, s):
        if p >= nxt and partner[p:p + b] not in grams:
            cnt += 1
            nxt = p + b
    return cnt
