"""CPU: the lane-level math of the banded GPU kernel (isocon_amd/csrc/band_core.hpp, shared host/device header) is run
through a wave emulator (tests/emul/band_emul.cpp, g++) and compared with the oracle DP.  Catches algorithmic errors
in the band geometry, virtual rows, multi-word carries and early exit without a GPU."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "emul", "_band_emul.so")
SRC = os.path.join(HERE, "emul", "band_emul.cpp")
CORE = os.path.join(os.path.dirname(HERE), "isocon_amd", "csrc", "band_core.hpp")
CORE2 = os.path.join(os.path.dirname(HERE), "isocon_amd", "csrc", "ed_lanes_core.hpp")


# Every case runs twice: on the plain build and on a -fsanitize=undefined build of the same sources (the lane-level headers
# are full of shifts by computed amounts; an out-of-range shift is undefined in C++ and MASKED on the GPU, so it has to be
# absent, not merely harmless here).  The sanitizer aborts the process at the first finding.
@pytest.fixture(scope="module", params=["plain", "ubsan"])
def emul(request):
    so = SO if request.param == "plain" else SO.replace(".so", "_ubsan.so")
    flags = ["-O2"] if request.param == "plain" else ["-O1", "-g", "-fsanitize=undefined", "-fno-sanitize-recover=all", "-static-libubsan"]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [SRC, CORE, CORE2]):
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-o", so, SRC])
    return ctypes.CDLL(so)


def run_tile(L, W, pat, texts, ks):
    n = len(texts)
    arr = (ctypes.c_char_p * n)(*[t.encode() for t in texts])
    tl = (ctypes.c_int * n)(*[len(t) for t in texts])
    kk = (ctypes.c_int * n)(*ks)
    out = (ctypes.c_int32 * n)()
    L.emul_band_tile(W, pat.encode(), len(pat), n, arr, tl, kk, out)
    return list(out)


def _mut(rng, s, rate):
    out = []
    for c in s:
        r = rng.random()
        if r < rate * 0.3:
            continue
        if r < rate * 0.6:
            out.append(rng.choice("ACGT")); out.append(c); continue
        if r < rate:
            out.append(rng.choice("ACGT")); continue
        out.append(c)
    return "".join(out) or "A"


@pytest.mark.parametrize("W", [1, 2, 4, 8])
def test_tiles_against_dp(emul, W):
    rng = random.Random(100 + W)
    undetermined = total = 0
    for it in range(120):
        m = rng.choice([1, 2, 5, 31, 63, 64, 65, 100, 257, 600])
        pat = "".join(rng.choice("ACGT") for _ in range(m))
        texts, ks = [], []
        spread = rng.choice([0, 0, 1, 3, 12, 50])
        for _ in range(rng.choice([1, 7, 64])):
            r = rng.random()
            if r < 0.6:
                t = _mut(rng, pat, rng.choice([0.005, 0.03, 0.08, 0.2]))
            elif r < 0.8:
                t = "".join(rng.choice("ACGT") for _ in range(max(1, m + rng.randint(-spread, spread))))
            else:
                cut = rng.randint(0, m - 1); ln = rng.randint(0, min(2 * spread, m - cut)); t = (pat[:cut] + pat[cut + ln:]) or "A"
            texts.append(t)
            kmax = 64 * W - 1
            ks.append(rng.choice([kmax, kmax, rng.randint(0, kmax), 0, 1, -1]))
        for t, k, r in zip(texts, ks, run_tile(emul, W, pat, texts, ks)):
            total += 1
            if r == -2:          # window could not certify this lane's k: the host re-tiles such pairs
                undetermined += 1
                continue
            d = O.ed_dp(pat, t)
            assert r == (-1 if (k < 0 or d > k) else d), (W, m, len(t), k, d, r)
    assert undetermined < 0.1 * total


def test_single_lane_tiles_are_always_determined(emul):
    rng = random.Random(9)
    for _ in range(300):
        W = rng.choice([1, 2, 4])
        m = rng.randint(1, 400)
        pat = "".join(rng.choice("ACGT") for _ in range(m))
        t = _mut(rng, pat, 0.1) if rng.random() < 0.7 else "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 400)))
        k = rng.randint(0, 64 * W - 1)
        r = run_tile(emul, W, pat, [t], [k])[0]
        d = O.ed_dp(pat, t)
        assert r == (d if d <= k else -1)


def test_one_pair_per_lane_routine_against_dp(emul):
    """lane_pair_distance (csrc/ed_lanes_core.hpp, what k_ed_lanes runs per lane): 96-bit plane registers, per-lane band origin,
    masked first blocks and text tails -- every threshold 0..63, lengths around the 32- and 64-column boundaries, both signs of the
    length difference."""
    emul.emul_ed_lane.restype = ctypes.c_int32
    rng = random.Random(29)
    cases = 0
    for L in (1, 2, 5, 31, 32, 33, 63, 64, 65, 95, 96, 97, 128, 200, 700):
        for _ in range(40):
            x = "".join(rng.choice("ACGT") for _ in range(L))
            y = _mut(rng, x, rng.choice([0.0, 0.02, 0.1, 0.3]))
            if rng.random() < 0.3:
                x, y = y, x
            if rng.random() < 0.1:
                y = "".join(rng.choice("ACGT") for _ in range(max(1, len(x) + rng.randint(-5, 5))))
            for k in (0, 1, 2, 7, 31, 32, 33, 62, 63, rng.randint(0, 63)):
                got = emul.emul_ed_lane(x.encode(), len(x), y.encode(), len(y), k)
                assert got == O.ed_bounded(x, y, k), (x, y, k, got)
                cases += 1
    assert cases > 5000
