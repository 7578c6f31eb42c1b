"""The host-side CSR assembly (isocon_amd/csrc/nn_finalize_host.hpp = what isocon_nn_finalize runs, and the fallback of the device
routine) compiled for the CPU with g++, plain and with -fsanitize=address,undefined, against a pure-Python statement of the
reference's insertion order (NNG:155-178: ascending offset, lower neighbour before upper; only edges attaining the final minimum)."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emul", "finalize_host.cpp")
HDR = os.path.join(os.path.dirname(HERE), "isocon_amd", "csrc", "nn_finalize_host.hpp")


def build(kind):
    so = os.path.join(HERE, "emul", "_finalize_host%s.so" % ("" if kind == "plain" else "_asan"))
    flags = ["-O2"] if kind == "plain" else ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [SRC, HDR]):
        subprocess.check_call(["g++"] + flags + ["-std=c++17", "-fPIC", "-shared", "-Wall", "-o", so, SRC])
    return so


def expected(n, best, hits):
    rows = [set() for _ in range(n)]
    for e, o, d in hits:
        if 0 <= e < n and 0 <= o < n and d >= 0 and d == best[e]:
            rows[e].add(int(o))
    out_best, row_ptr, cols = [], [0], []
    for i in range(n):
        r = sorted(rows[i], key=lambda x: (abs(x - i), x))
        cols += r
        row_ptr.append(len(cols))
        out_best.append(int(best[i]) if r else -1)
    return out_best, row_ptr, cols


def run_cases(so):
    import ctypes
    L = ctypes.CDLL(so)
    i32p, u64p, u32p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32)
    L.emul_nn_finalize.restype = ctypes.c_int
    L.emul_nn_finalize.argtypes = [ctypes.c_uint32, i32p, i32p, ctypes.c_uint64, i32p, u64p, u32p, ctypes.c_uint64, u64p]

    def call(n, best, hits, cap):
        best = np.ascontiguousarray(best, dtype=np.int32)
        hits = np.ascontiguousarray(hits, dtype=np.int32).reshape(-1, 3)
        ob = np.full(max(n, 1), 77, dtype=np.int32)
        rp = np.zeros(n + 1, dtype=np.uint64)
        cols = np.zeros(max(cap, 1), dtype=np.uint32)
        need = ctypes.c_uint64(0)
        rc = L.emul_nn_finalize(n, best.ctypes.data_as(i32p), hits.ctypes.data_as(i32p) if len(hits) else None, len(hits), ob.ctypes.data_as(i32p),
                                rp.ctypes.data_as(u64p), cols.ctypes.data_as(u32p) if cap else None, cap, ctypes.byref(need))
        return rc, ob[:n].tolist(), rp.tolist(), cols[:int(rp[n])].tolist(), int(need.value)

    rng = np.random.Generator(np.random.PCG64(11))
    # empty graph, one entry, no hits
    assert call(0, [0], np.zeros((0, 3)), 0)[0] == 0
    rc, ob, rp, cols, need = call(1, [5], np.zeros((0, 3)), 4)
    assert (rc, ob, rp, cols) == (0, [-1], [0, 0], [])
    for n, n_hits, dup in ((5, 40, True), (300, 3000, True), (2000, 9000, False), (50, 4000, True)):          # the last: rows of > 16 neighbours (std::sort)
        best = rng.integers(1, 4, size=n).astype(np.int32)
        hits = np.stack([rng.integers(-1, n + 1, size=n_hits), rng.integers(-1, n + 1, size=n_hits), rng.integers(-1, 5, size=n_hits)], axis=1).astype(np.int32)
        if dup:
            hits = np.concatenate([hits, hits[::3]])          # the same pair reported by more than one phase / rank
        want = expected(n, best, hits.tolist())
        rc, ob, rp, cols, need = call(n, best, hits, len(hits))
        assert rc == 0 and (ob, rp, cols) == want and need == len(want[2])
        if len(want[2]) > 1:          # too little room: the required size comes back, nothing is written past the capacity
            rc, _, _, _, need = call(n, best, hits, len(want[2]) - 1)
            assert rc == -4 and need == len(want[2])
    # argument checks
    assert call(3, [1, 1, 1], np.array([[0, 1, 1]]), 0)[0] in (-4,)          # cols_cap 0 with an edge to deliver: capacity
    return True


@pytest.mark.parametrize("kind", ["plain", "asan_ubsan"])
def test_host_finalize(kind):
    so = build(kind)
    if kind == "plain":
        assert run_cases(so)
        return
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan with this gcc")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    code = "import sys; sys.path.insert(0, %r); import test_finalize_host as T; print('sanitized ok' if T.run_cases(%r) else 'failed')" % (HERE, so)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sanitized ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
