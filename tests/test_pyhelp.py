"""The CPython helper of the wrappers (isocon_amd/cpy/_pyhelp.c: raw addresses from Python ints, CPython objects, pthreads) against
pure-Python equivalents -- no GPU needed.  The same checks run a second time in a child interpreter against a build with
-fsanitize=address,undefined (libasan preloaded), the way the wave emulators are built with -fsanitize=undefined."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _helper():
    sys.path.insert(0, ROOT)
    if os.environ.get("ISOCON_PYHELP_UNDER_TEST"):          # the sanitizer child: import the instrumented build by path
        import importlib.util
        spec = importlib.util.spec_from_file_location("_pyhelp", os.environ["ISOCON_PYHELP_UNDER_TEST"])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    from isocon_amd import _lib
    assert _lib.build_pyhelp() is not None
    H = _lib.pyhelp()
    assert H is not None
    return H


def check_str_pointers(H):
    import ctypes
    for seqs in ([], [""], ["ACGT", "", "A" * 5000, "TTGACA"], ["ACGT"[i % 4] * (i % 17) for i in range(3000)]):
        ptrs = np.zeros(max(len(seqs), 1), dtype=np.uint64)
        lens = np.full(max(len(seqs), 1), 99, dtype=np.uint64)
        total = H.str_pointers(seqs, ptrs.ctypes.data, lens.ctypes.data)
        assert total == sum(len(s) for s in seqs)
        for i, s in enumerate(seqs):
            assert int(lens[i]) == len(s)
            assert ctypes.string_at(int(ptrs[i]), len(s)) == s.encode()
    ptrs = np.zeros(4, dtype=np.uint64)
    lens = np.zeros(4, dtype=np.uint64)
    with pytest.raises(ValueError):          # non-ASCII: what the store reports as a symbol outside ACGT
        H.str_pointers(["ACGT", "ACéT"], ptrs.ctypes.data, lens.ctypes.data)
    with pytest.raises(TypeError):
        H.str_pointers(["ACGT", b"ACGT"], ptrs.ctypes.data, lens.ctypes.data)
    with pytest.raises(TypeError):
        H.str_pointers(("ACGT",), ptrs.ctypes.data, lens.ctypes.data)


def check_split_ascii(H):
    rng = np.random.Generator(np.random.PCG64(5))
    for n, mean in ((0, 0), (1, 0), (7, 30), (4000, 100), (3000, 3200)):          # the last one is > 8 MB: the threaded copy
        lens = rng.integers(0, 2 * mean + 1, size=n)
        if n > 2:
            lens[1] = 0          # zero-length strings inside
            lens[-1] = 0
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=ptr[1:])
        buf = rng.integers(0, 4, size=int(ptr[n]) + 1).astype(np.uint8)
        buf = np.frombuffer(b"ACG-", dtype=np.uint8)[buf].copy()
        out = H.split_ascii(buf.ctypes.data, ptr.ctypes.data, n)
        raw = buf.tobytes()
        assert out == [raw[ptr[i]:ptr[i + 1]].decode() for i in range(n)]
        if n > 2000 and mean > 3000:
            assert int(ptr[n]) >= 8 << 20
    # a window that does not start at 0
    buf = np.frombuffer(b"xxACGTTTGA", dtype=np.uint8).copy()
    ptr = np.array([2, 6, 6, 10], dtype=np.int64)
    assert H.split_ascii(buf.ctypes.data, ptr.ctypes.data, 3) == ["ACGT", "", "TTGA"]
    with pytest.raises(ValueError):
        H.split_ascii(buf.ctypes.data, np.array([4, 2], dtype=np.int64).ctypes.data, 1)


def check_split_ascii_rows(H):
    """selected rows of a packed buffer, equal rows as ONE object (also across different lengths of a common prefix, empty rows, the
    threaded hashing above 8 MB, rows given in any order and more than once)"""
    rng = np.random.Generator(np.random.PCG64(6))
    for n, mean, distinct in ((0, 0, 1), (1, 0, 1), (9, 30, 3), (5000, 100, 40), (3000, 3200, 25)):
        pool = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(0, 2 * mean + 1)))] for _ in range(distinct)]
        if distinct > 2:
            pool[1] = pool[0][:len(pool[0]) // 2]          # a proper prefix of another row
            pool[2] = pool[0][:0]                          # empty
        pick = rng.integers(0, distinct, size=n)
        lens = np.array([len(pool[c]) for c in pick], dtype=np.int64)
        ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=ptr[1:])
        buf = np.concatenate([pool[c] for c in pick] + [np.zeros(1, dtype=np.uint8)]).astype(np.uint8)
        rows = rng.permutation(n).astype(np.int64)
        if n > 2:
            rows = np.concatenate([rows, rows[:3]])
        out = H.split_ascii_rows(buf.ctypes.data, ptr.ctypes.data, len(ptr) - 1, rows.ctypes.data, len(rows))
        raw = buf.tobytes()
        assert out == [raw[ptr[r]:ptr[r + 1]].decode() for r in rows.tolist()]
        first = {}
        for r, s_r in zip(rows.tolist(), out):
            assert first.setdefault(s_r, s_r) is s_r, "equal rows are one object"
        assert len(set(map(id, out))) == len(set(out))
    buf = np.frombuffer(b"ACGT", dtype=np.uint8).copy()
    with pytest.raises(ValueError):
        H.split_ascii_rows(buf.ctypes.data, np.array([4, 2], dtype=np.int64).ctypes.data, 1, np.zeros(1, dtype=np.int64).ctypes.data, 1)
    with pytest.raises(ValueError):
        H.split_ascii_rows(buf.ctypes.data, np.array([0, 2], dtype=np.int64).ctypes.data, 1, np.array([-1], dtype=np.int64).ctypes.data, 1)
    with pytest.raises(ValueError):          # a row behind the offsets
        H.split_ascii_rows(buf.ctypes.data, np.array([0, 2], dtype=np.int64).ctypes.data, 1, np.array([1], dtype=np.int64).ctypes.data, 1)


def check_invariant_partners(H):
    """the C statement of end_invariant_functions._pair_is_invariant against the Python one (itself pinned to the reference by g13): substrings
    whose FIRST occurrence decides (repeats: a later occurrence would pass), suffix-prefix overlaps in both directions, thresholds 0 .. 20,
    empty strings, equal strings, pairs outside the length window the caller uses"""
    import random
    sys.path.insert(0, ROOT)
    from isocon_amd import end_invariant_functions as END
    rng = random.Random(13)
    for thr in (0, 1, 5, 15, 20, 25, 40):
        for rep in range(60):
            L = rng.choice([0, 1, 3, 30, 120, 400])
            alpha = rng.choice(["ACGT", "AC", "A"])          # (short alphabets: repeats, many occurrences)
            base = "".join(rng.choice(alpha) for _ in range(L))
            others = []
            for _ in range(40):
                r = rng.random()
                cut_a, cut_b = rng.randint(0, thr + 3), rng.randint(0, thr + 3)
                if thr > 16 and rng.random() < 0.4:            # lengths more than thr + 16 apart, still inside the caller's window of 2 thr
                    cut_a, cut_b = rng.randint(thr + 17, 2 * thr), 0
                t = base[cut_a:len(base) - cut_b] if r < 0.35 else base
                if 0.35 <= r < 0.55:
                    t = "".join(rng.choice(alpha) for _ in range(rng.randint(0, thr + 2))) + base[rng.choice([rng.randint(0, thr + 2), cut_a]):]
                elif 0.55 <= r < 0.75:
                    t = base[:len(base) - rng.randint(0, thr + 2)] + "".join(rng.choice(alpha) for _ in range(rng.randint(0, thr + 2)))
                elif 0.75 <= r < 0.9 and t:
                    k = rng.randrange(len(t))
                    t = t[:k] + rng.choice(alpha) + t[k + 1:]
                elif r >= 0.97:
                    t = "".join(rng.choice(alpha) for _ in range(rng.randint(0, L + 5)))
                others.append(t)
            got = H.invariant_partners(base, others, thr)
            assert got == [i for i, t in enumerate(others) if END._pair_is_invariant(base, t, thr)], (thr, rep, base)
    assert H.invariant_partners("ACGT", [], 3) == []
    with pytest.raises(TypeError):
        H.invariant_partners("ACGT", ["AC", 5], 3)
    with pytest.raises(TypeError):
        H.invariant_partners("AC\u0394T", ["AC"], 3)


def check_best_solution(H):
    """the C statement of functions.get_best_solution (first occurrence, threading along the optimal alignment with its tie rule, best-offset
    fallback, the all-gap and the too-long cases) against the Python one, which the correction fixtures (g11) pin to the reference"""
    import random
    sys.path.insert(0, ROOT)
    from isocon_amd.functions import get_best_solution
    rng = random.Random(31)
    for it in range(30000):
        L = rng.randint(0, 12)
        core = "".join(rng.choice("ACGT") for _ in range(L))
        mx = "-" + core + "-" if rng.random() < 0.8 else core
        r = rng.random()
        if r < 0.05:
            q = "-"
        elif r < 0.5:
            q = list(core)
            for _ in range(rng.randint(0, 3)):
                if q and rng.random() < 0.5:
                    del q[rng.randrange(len(q))]
                elif q:
                    q[rng.randrange(len(q))] = rng.choice("ACGT")
                else:
                    q.append(rng.choice("ACGT"))
            q = "".join(q)
        else:
            q = "".join(rng.choice("AC" if it % 3 else "ACGT") for _ in range(rng.randint(0, 14)))
        assert H.best_solution(mx, q) == "".join(get_best_solution(mx, q)).encode(), (mx, q)
    long_mx = "-" + "ACGT" * 60 + "-"
    assert H.best_solution(long_mx, "ACGTTACG") == "".join(get_best_solution(long_mx, "ACGTTACG")).encode()
    with pytest.raises(ValueError):
        H.best_solution("A" * 256, "A")
    with pytest.raises(TypeError):
        H.best_solution("A\u0394", "A")


def check_csr_to_dict(H):
    keys = ["k%d" % i for i in range(6)]
    best = np.array([3, -1, 2, 2, 7, -1], dtype=np.int32)
    row_ptr = np.array([0, 2, 2, 3, 5, 6, 6], dtype=np.int64)
    cols = np.array([2, 1, 0, 5, 4, 0], dtype=np.uint32)

    def plain(isq):
        return {keys[i]: {keys[int(c)]: int(best[i]) for c in cols[row_ptr[i]:row_ptr[i + 1]]} for i in range(6) if isq is None or isq[i]}

    got = H.csr_to_dict(keys, 0, best.ctypes.data, row_ptr.ctypes.data, cols.ctypes.data, 6)
    assert [(k, list(v.items())) for k, v in got.items()] == [(k, list(v.items())) for k, v in plain(None).items()]
    isq = np.array([1, 0, 1, 1, 0, 1], dtype=np.uint8)
    got = H.csr_to_dict(keys, isq.ctypes.data, best.ctypes.data, row_ptr.ctypes.data, cols.ctypes.data, 6)
    assert [(k, list(v.items())) for k, v in got.items()] == [(k, list(v.items())) for k, v in plain(isq).items()]
    assert H.csr_to_dict([], 0, best.ctypes.data, np.zeros(1, dtype=np.int64).ctypes.data, 0, 0) == {}
    bad = cols.copy()
    bad[3] = 6          # a column outside the key list
    with pytest.raises(ValueError):
        H.csr_to_dict(keys, 0, best.ctypes.data, row_ptr.ctypes.data, bad.ctypes.data, 6)
    with pytest.raises(TypeError):
        H.csr_to_dict(keys[:3], 0, best.ctypes.data, row_ptr.ctypes.data, cols.ctypes.data, 6)
    # a large one against the loop
    rng = np.random.Generator(np.random.PCG64(6))
    n = 5000
    deg = rng.integers(0, 4, size=n)
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    c = rng.integers(0, n, size=int(rp[n])).astype(np.uint32)
    b = rng.integers(1, 60, size=n).astype(np.int32)
    ks = ["acc_%d" % i for i in range(n)]
    got = H.csr_to_dict(ks, 0, b.ctypes.data, rp.ctypes.data, c.ctypes.data, n)
    want = {ks[i]: {ks[int(x)]: int(b[i]) for x in c[rp[i]:rp[i + 1]]} for i in range(n)}
    assert list(got) == list(want) and all(list(got[k].items()) == list(want[k].items()) for k in want)


def check_pair_ids(H):
    index = {"s%d" % i: i for i in range(100)}
    pairs = [("s%d" % (i % 100), "s%d" % ((7 * i) % 100)) for i in range(1000)]
    a = np.zeros(1000, dtype=np.uint32)
    b = np.zeros(1000, dtype=np.uint32)
    assert H.pair_ids(index, pairs, a.ctypes.data, b.ctypes.data) == 1000
    assert a.tolist() == [index[x] for x, _ in pairs] and b.tolist() == [index[y] for _, y in pairs]
    assert H.pair_ids(index, [], a.ctypes.data, b.ctypes.data) == 0
    assert H.pair_ids(index, pairs[:5] + [("s1", "nope")], a.ctypes.data, b.ctypes.data) == -1 - 5
    assert H.pair_ids(index, [("nope", "s1")], a.ctypes.data, b.ctypes.data) == -1
    with pytest.raises(TypeError):
        H.pair_ids(index, [("s1",)], a.ctypes.data, b.ctypes.data)
    with pytest.raises(TypeError):
        H.pair_ids(index, [(["unhashable"], "s1")], a.ctypes.data, b.ctypes.data)
    with pytest.raises(TypeError):
        H.pair_ids(list(index), pairs, a.ctypes.data, b.ctypes.data)


def check_rank_strings(H):
    import random
    rng = random.Random(3)
    for seqs in ([], ["A"], ["", "A", "", "AC", "A"], ["".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 9))) for _ in range(3000)],
                 ["ACGT" * 50 + "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 5))) for _ in range(500)]):          # long shared prefixes, duplicates
        out = np.full(max(len(seqs), 1), 7, dtype=np.uint32)
        H.rank_strings(seqs, out.ctypes.data)
        want = [0] * len(seqs)
        for r, v in enumerate(sorted(range(len(seqs)), key=seqs.__getitem__)):          # (stable: equal strings in index order)
            want[v] = r
        assert out[:len(seqs)].tolist() == want
    out = np.zeros(4, dtype=np.uint32)
    with pytest.raises(TypeError):
        H.rank_strings(["A", b"C"], out.ctypes.data)
    with pytest.raises(TypeError):
        H.rank_strings(["A", "Cé"], out.ctypes.data)
    with pytest.raises(TypeError):
        H.rank_strings(("A",), out.ctypes.data)


def check_group_keys_by_value(H):
    for d in ({}, {"a": "X"}, {"a": "X", "b": "Y", "c": "X", "d": "Z", "e": "Y"}, {i: "v%d" % (i % 7) for i in range(1000)}):
        want = {}
        for k, v in d.items():
            want.setdefault(v, []).append(k)
        got = H.group_keys_by_value(d)
        assert got == want and list(got) == list(want)
    with pytest.raises(TypeError):
        H.group_keys_by_value([("a", "X")])
    with pytest.raises(TypeError):
        H.group_keys_by_value({"a": ["unhashable"]})
    with pytest.raises(TypeError):          # values must be exact str: their hash / eq cannot run code that changes the dict under the loop
        H.group_keys_by_value({"a": 1})


def check_unique_values_by_length(H):
    rng = np.random.Generator(np.random.PCG64(9))
    for n in (0, 1, 5, 3000):
        vals = ["ACGT"[int(rng.integers(0, 4))] * int(rng.integers(0, 40)) + "G" * int(rng.integers(0, 3)) for _ in range(n)]
        S = {"acc%d" % i: v for i, v in enumerate(vals)}
        inv = {seq: acc for acc, seq in S.items()}                         # NNG:243
        want = sorted(inv.items(), key=lambda x: len(x[0]))                # NNG:246 (stable)
        seqs, accs = H.unique_values_by_length(S)
        assert seqs == [s for s, _ in want] and accs == [a for _, a in want]
    with pytest.raises(TypeError):
        H.unique_values_by_length({"a": 5})
    with pytest.raises(TypeError):
        H.unique_values_by_length([("a", "ACGT")])


def check_flatten_pairs(H):
    m = {"x": {"p": 1, "q": 2}, "y": {}, "z": {"r": 3}}
    assert H.flatten_pairs(m) == ([("x", "p"), ("x", "q"), ("z", "r")], [1, 2, 3])
    sets = {"x": {"p", "q", "r"}, "y": set(), "z": ["k", "k"], "w": ("t",)}
    pairs, vals = H.flatten_pairs(sets)
    assert vals is None and pairs == [(k, v) for k, inner in sets.items() for v in inner]
    assert H.flatten_pairs({}) == ([], None)
    big = {"c%d" % i: {"m%d_%d" % (i, j): i * j for j in range(i % 13)} for i in range(500)}
    assert H.flatten_pairs(big) == ([(a, b) for a, inner in big.items() for b in inner], [v for inner in big.values() for v in inner.values()])
    with pytest.raises(TypeError):
        H.flatten_pairs({"x": {"p": 1}, "y": {"q"}})
    with pytest.raises(TypeError):
        H.flatten_pairs({"x": 5})
    with pytest.raises(TypeError):          # a generator: its iteration runs Python code
        H.flatten_pairs({"x": (c for c in "ab")})
    with pytest.raises(TypeError):
        H.flatten_pairs([("x", "p")])


def check_distance_dict(H):
    pairs = [("x", "p"), ("x", "q"), ("y", "p"), ("x", "r"), ("y", "p")]          # an outer key that comes back, a pair filed twice (the last wins, as in the loop)
    ed = np.array([3, 0, 700, -1, 5], dtype=np.int32)
    d = H.distance_dict(pairs, ed.ctypes.data)
    want = {}
    for (a, b), v in zip(pairs, ed.tolist()):
        want.setdefault(a, {})[b] = v
    assert d == want and list(d) == ["x", "y"] and list(d["x"]) == ["p", "q", "r"] and type(d["x"]["p"]) is int
    assert H.distance_dict([], ed.ctypes.data) == {}
    with pytest.raises(TypeError):
        H.distance_dict([("x",)], ed.ctypes.data)
    with pytest.raises(TypeError):
        H.distance_dict((("x", "p"),), ed.ctypes.data)
    with pytest.raises(TypeError):          # an unhashable key
        H.distance_dict([(["x"], "p")], ed.ctypes.data)


def check_pairs_of(H):
    index = {"x": 0, "y": 1, "p": 2, "q": 3, "r": 4}
    m = {"x": {"p": 1, "q": 2}, "y": set(), "q": ["r", "r", "x"], "p": ("y",)}
    a = np.full(8, 99, np.uint32); b = np.full(8, 99, np.uint32)
    outer, counts, inner = H.pairs_of(m, index, a.ctypes.data, b.ctypes.data, 8)
    assert outer == ["x", "q", "p"] and counts == [2, 3, 1] and inner == ["p", "q", "r", "r", "x", "y"]
    assert a[:6].tolist() == [0, 0, 3, 3, 3, 2] and b[:6].tolist() == [2, 3, 4, 4, 0, 1] and a[6] == 99
    ed = np.array([5, 6, 7, 8, 9, 10], dtype=np.int32)
    d = H.distance_rows(outer, counts, inner, ed.ctypes.data)
    want = {}
    p = 0
    for k1, inn in m.items():
        for k2 in inn:
            want.setdefault(k1, {})[k2] = int(ed[p]); p += 1
    assert d == want and list(d) == ["x", "q", "p"] and list(d["q"]) == ["r", "x"]
    assert H.pairs_of({"x": {"zz"}}, index, a.ctypes.data, b.ctypes.data, 8) is None          # a member the store does not hold
    assert H.pairs_of({"zz": {"x"}}, index, a.ctypes.data, b.ctypes.data, 8) is None
    assert H.pairs_of(m, index, a.ctypes.data, b.ctypes.data, 3) is None                       # capacity
    assert H.pairs_of({}, index, a.ctypes.data, b.ctypes.data, 8) == ([], [], [])
    with pytest.raises(TypeError):
        H.pairs_of({"x": 5}, index, a.ctypes.data, b.ctypes.data, 8)
    with pytest.raises(TypeError):
        H.pairs_of({"x": (c for c in "pq")}, index, a.ctypes.data, b.ctypes.data, 8)
    with pytest.raises(ValueError):
        H.distance_rows(outer, [2, 3, 2], inner, ed.ctypes.data)
    with pytest.raises(TypeError):
        H.distance_rows(outer, counts[:2], inner, ed.ctypes.data)


def check_alignment_dict(H):
    pairs = [("x", "p"), ("x", "q"), ("y", "p")]
    la, lb = ["A-C", "GG", ""], ["AAC", "G-", ""]
    res = np.arange(18, dtype=np.int32).reshape(3, 6)
    out, d = H.alignment_dict(pairs, la, lb, res.ctypes.data, len(res))
    assert out == [("A-C", "AAC", (3, 4, 5)), ("GG", "G-", (9, 10, 11)), ("", "", (15, 16, 17))]
    assert d == {"x": {"p": out[0], "q": out[1]}, "y": {"p": out[2]}} and list(d) == ["x", "y"] and list(d["x"]) == ["p", "q"]
    assert d["x"]["q"] is out[1]                     # the very tuple objects (the ops cache finds an alignment by identity)
    assert H.alignment_dict([], [], [], res.ctypes.data, 0) == ([], {})
    with pytest.raises(ValueError):          # fewer result rows than pairs
        H.alignment_dict(pairs, la, lb, res.ctypes.data, 2)
    with pytest.raises(TypeError):
        H.alignment_dict(pairs, la[:2], lb, res.ctypes.data, len(res))
    with pytest.raises(TypeError):
        H.alignment_dict([("x",)], ["A"], ["A"], res.ctypes.data, len(res))


def check_lazy_rows(H):
    class V(object):
        __slots__ = ("_batch", "_p", "_edit")

        def __init__(self, batch, p, edit):
            self._batch, self._p, self._edit = batch, p, edit

    batch = object()
    pairs = [("c1", "a"), ("c1", "b"), ("c2", "a"), ("c2", "z"), ("c1", "c")]
    keep = np.array([1, 0, 1, 1, 1], dtype=np.uint8)
    edit = np.array([5, 6, 70000, 8, 0], dtype=np.int32)
    out = {"c1": {"c1": 0}, "c2": {"c2": 0}}
    rows_of = {}
    assert H.lazy_rows(V, batch, pairs, keep.ctypes.data, edit.ctypes.data, len(keep), out, rows_of) == 4
    with pytest.raises(ValueError):          # arrays of another length than the pair list
        H.lazy_rows(V, batch, pairs, keep.ctypes.data, edit.ctypes.data, len(keep) - 1, {k: {} for k in out}, {})
    assert rows_of == {"c1": [0, 4], "c2": [2, 3]}
    assert list(out["c1"]) == ["c1", "a", "c"] and list(out["c2"]) == ["c2", "a", "z"]
    v = out["c2"]["a"]
    assert type(v) is V and v._batch is batch and v._p == 2 and v._edit == 70000
    for c in ("c1", "c2"):
        assert H.lazy_rows_intact(out[c], c, V, batch, pairs) is True
    assert H.lazy_rows_intact(out["c1"], "c1", V, object(), pairs) is False          # another batch
    out["c1"]["b"] = out["c1"].pop("a")                                                # re-keyed
    assert H.lazy_rows_intact(out["c1"], "c1", V, batch, pairs) is False
    out["c2"]["z"] = (8, "A", "A", 1)                                                  # replaced by a plain tuple
    assert H.lazy_rows_intact(out["c2"], "c2", V, batch, pairs) is False
    with pytest.raises(KeyError):
        H.lazy_rows(V, batch, [("nope", "a")], keep.ctypes.data, edit.ctypes.data, 1, {}, {})


def test_group_keys_by_value():
    check_group_keys_by_value(_helper())


def test_unique_values_by_length():
    check_unique_values_by_length(_helper())


def test_flatten_pairs():
    check_flatten_pairs(_helper())


def test_distance_dict():
    check_distance_dict(_helper())


def test_pairs_of():
    check_pairs_of(_helper())


def test_alignment_dict():
    check_alignment_dict(_helper())


def test_lazy_rows():
    check_lazy_rows(_helper())


def test_rank_strings():
    check_rank_strings(_helper())


def test_str_pointers():
    check_str_pointers(_helper())


def test_split_ascii():
    check_split_ascii(_helper())


def test_split_ascii_rows():
    check_split_ascii_rows(_helper())


def test_invariant_partners():
    check_invariant_partners(_helper())


def test_best_solution():
    check_best_solution(_helper())


def test_csr_to_dict():
    check_csr_to_dict(_helper())


def test_pair_ids():
    check_pair_ids(_helper())


def test_forced_off_and_abi_named():
    sys.path.insert(0, ROOT)
    import sysconfig
    from isocon_amd import _lib
    assert _lib.PYHELP_SO.endswith(sysconfig.get_config_var("EXT_SUFFIX"))
    os.environ["ISOCON_NO_PYHELP"] = "1"
    try:
        assert _lib.pyhelp() is None
    finally:
        del os.environ["ISOCON_NO_PYHELP"]
    assert _lib.pyhelp() is not None


def test_under_address_and_undefined_sanitizers(tmp_path):
    sys.path.insert(0, ROOT)
    from isocon_amd import _lib
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan with this gcc")
    so = str(tmp_path / "_pyhelp.so")
    assert _lib.build_pyhelp(extra_flags=("-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"), out=so) == so
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               ISOCON_PYHELP_UNDER_TEST=so)
    code = ("import sys; sys.path.insert(0, %r); import test_pyhelp as T; H = T._helper(); "
            "T.check_group_keys_by_value(H); T.check_rank_strings(H); T.check_str_pointers(H); T.check_split_ascii(H); T.check_split_ascii_rows(H); T.check_invariant_partners(H); T.check_best_solution(H); T.check_csr_to_dict(H); T.check_pair_ids(H); "
            "T.check_unique_values_by_length(H); T.check_flatten_pairs(H); T.check_distance_dict(H); T.check_pairs_of(H); T.check_alignment_dict(H); T.check_lazy_rows(H); print('sanitized ok')" % os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "sanitized ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
