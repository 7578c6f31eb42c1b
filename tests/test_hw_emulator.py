"""CPU: the lane-level math of the bit-parallel infix kernels (isocon_amd/csrc/hw_core.hpp, shared host/device header) run
through a tile emulator (tests/emul/hw_emul.cpp, g++) and compared with the oracle's full matrices (hw_locate + nw_path):
distance, start, end, leading and trailing insertion run of one query against up to 64 targets per tile -- zero top row with
"virtual" rows, reversed streams of the START pass, stored VP / HP vectors and the walk of the TRACE pass, early abandon."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "emul", "_hw_emul.so")
SRC = os.path.join(HERE, "emul", "hw_emul.cpp")
CORES = [os.path.join(os.path.dirname(HERE), "isocon_amd", "csrc", f) for f in ("band_core.hpp", "hw_core.hpp")]


# Every case runs twice: on the plain build and on a -fsanitize=undefined build of the same sources (the lane-level headers
# are full of shifts by computed amounts; an out-of-range shift is undefined in C++ and MASKED on the GPU, so it has to be
# absent, not merely harmless here).  The sanitizer aborts the process at the first finding.
@pytest.fixture(scope="module", params=["plain", "ubsan"])
def emul(request):
    so = SO if request.param == "plain" else SO.replace(".so", "_ubsan.so")
    flags = ["-O2"] if request.param == "plain" else ["-O1", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=all", "-static-libubsan"]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [SRC] + CORES):
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, SRC])
    return ctypes.CDLL(so)


def run_tile(L, W, q, targets, ks):
    n = len(targets)
    arr = (ctypes.c_char_p * n)(*[t.encode() for t in targets])
    tl = (ctypes.c_int * n)(*[len(t) for t in targets])
    kk = (ctypes.c_int * n)(*ks)
    out = (ctypes.c_int32 * (5 * n))()
    L.emul_hw_tile(W, q.encode(), len(q), n, arr, tl, kk, out)
    return [list(out[5 * i:5 * i + 5]) for i in range(n)]


def expect(x, y, k):
    ed, start, end = O.hw_locate(x, y, k)
    if ed < 0:
        return [-1, -1, -1, 0, 0]
    _, ops = O.nw_path(x, y[start:end + 1])
    return [ed, start, end, ops[0][0] if ops[0][1] == "I" else 0, ops[-1][0] if ops[-1][1] == "I" else 0]


def _mut(rng, b, nmut, ends, alphabet="ACGT"):
    v = list(b)
    for _ in range(nmut):
        p = rng.randrange(len(v))
        r = rng.random()
        if r < 0.4:
            v[p] = rng.choice(alphabet)
        elif r < 0.7:
            del v[p]
        else:
            v.insert(p, rng.choice(alphabet))
    v = "".join(v)
    a, b2 = rng.randint(0, ends), rng.randint(0, ends)
    v = v[a:len(v) - b2]
    if rng.random() < 0.4:
        v = "".join(rng.choice(alphabet) for _ in range(rng.randint(1, ends + 1))) + v
    if rng.random() < 0.4:
        v += "".join(rng.choice(alphabet) for _ in range(rng.randint(1, ends + 1)))
    return v


@pytest.mark.parametrize("L,k,ends,W,alphabet", [(60, 5, 3, 1, "ACGT"), (150, 25, 6, 1, "ACGT"), (150, 25, 20, 2, "ACGT"), (400, 25, 20, 2, "AC"),
                                                 (700, 25, 20, 2, "ACGT"), (300, 40, 45, 4, "ACGT"), (400, 70, 60, 4, "AC"),
                                                 (500, 120, 50, 8, "ACGT"), (200, 0, 0, 1, "ACGT"), (90, 10, 30, 2, "A")],
                         ids=["tiny", "w1", "w2", "w2_low_complexity", "w2_long", "w4", "w4_low_complexity", "w8", "k0", "homopolymer"])
def test_tiles_equal_oracle(emul, L, k, ends, W, alphabet):
    rng = random.Random(L * 1000 + k + W)
    hits = total = 0
    for tile in range(12):
        b = "".join(rng.choice(alphabet) for _ in range(rng.randint(L - L // 8, L + L // 8)))
        q = _mut(rng, b, rng.choice([0, 1, 3, 8, 20]), ends, alphabet)
        nl = rng.choice([1, 2, 7, 33, 64])
        targets, ks = [], []
        for _ in range(nl):
            r = rng.random()
            if k == 0 and r < 0.5:
                t = "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 9))) + q + "".join(rng.choice(alphabet) for _ in range(rng.randint(0, 9)))
            elif r < 0.8:
                t = _mut(rng, b, rng.choice([0, 1, 3, 8, 15]), ends, alphabet)
            elif r < 0.9:
                t = "".join(rng.choice(alphabet) for _ in range(max(1, len(q) + rng.randint(-5, 5))))
            else:
                t = q[rng.randint(0, 3):]                      # shorter than the query
            targets.append(t or "A")
            ks.append(k if rng.random() < 0.8 else rng.randint(0, k))
        # keep what the tile's window can hold (the host classes pairs the same way)
        keep = [i for i in range(nl) if max(len(targets[i]) - len(q), 0) + 2 * k + 1 <= 64 * W]
        targets = [targets[i] for i in keep]; ks = [ks[i] for i in keep]
        if not targets:
            continue
        got = run_tile(emul, W, q, targets, ks)
        for t, kk, g in zip(targets, ks, got):
            e = expect(q, t, kk)
            assert g == e, (q, t, kk)
            hits += e[0] >= 0
            total += 1
    assert total > 20 and hits > total // 10


def test_exact_infix_and_self(emul):
    rng = random.Random(5)
    for _ in range(30):
        t = "".join(rng.choice("ACGT") for _ in range(rng.randint(40, 300)))
        a, b = rng.randint(0, 10), rng.randint(0, 10)
        q = t[a:len(t) - b]
        assert run_tile(emul, 1, q, [t, q], [10, 0]) == [expect(q, t, 10), [0, 0, len(q) - 1, 0, 0]]
