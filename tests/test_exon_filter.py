"""filter_exon_differences: the Python mirror and the ops-based C helper against fixtures produced by the
reference's modules/functions.py (tests/golden/make_golden.py), plus a GPU end-to-end check."""
import random
import re

import numpy as np
import pytest

from conftest import golden


def _load():
    g = golden("g6_exon_filter.json")
    aligned = {k1: {k2: (v[0], v[1], tuple(v[2])) for k2, v in inner} for k1, inner in g["alignments"]}
    return g, aligned


def test_string_route_matches_reference_fixture():
    from isocon_amd import functions as F
    g, aligned = _load()
    for case in g["cases"]:
        work = {k1: dict(inner) for k1, inner in aligned.items()}
        filtered = F.filter_exon_differences(work, case["min_exon_diff"], case["ignore_ends_len"])
        assert sorted(filtered) == case["filtered"]
        assert [[k1, list(inner.keys())] for k1, inner in work.items()] == case["remaining"]


def _ops_from_alignment(a1, a2):
    codes = []
    for x, y in zip(a1, a2):
        codes.append(3 if x == "-" else 2 if y == "-" else 0 if x == y else 1)
    ops, i = [], 0
    while i < len(codes):
        j = i
        while j < len(codes) and codes[j] == codes[i]:
            j += 1
        ops.append(((j - i) << 4) | codes[i])
        i = j
    return ops


def test_ops_route_equals_string_route():
    """C helper on CIGAR ops == string scan, on the fixture alignments and on random gapped strings."""
    import ctypes
    from isocon_amd import _lib
    from isocon_amd import functions as F
    L = _lib.load()
    g, aligned = _load()
    pairs = [v[:2] for inner in aligned.values() for v in inner.values()]
    rng = random.Random(4)
    for _ in range(300):     # random alignments with long and short gap runs, also at the ends
        a1, a2 = [], []
        for _seg in range(rng.randint(1, 8)):
            kind = rng.choice("MMMID")
            ln = rng.choice([1, 2, 5, 14, 15, 16, 19, 20, 21, 40])
            s = "".join(rng.choice("ACGT") for _ in range(ln))
            if kind == "M":
                a1.append(s); a2.append(s)
            elif kind == "I":
                a1.append(s); a2.append("-" * ln)
            else:
                a1.append("-" * ln); a2.append(s)
        x, y = "".join(a1), "".join(a2)
        if re.match(r"^-", x) and re.match(r"^-", y):
            continue
        pairs.append((x, y))
    for (mn, ig) in ((20, 15), (20, 0), (5, 3), (16, 15), (1, 0)):
        ops_list = [_ops_from_alignment(a, b) for a, b in pairs]
        ptr = np.zeros(len(ops_list) + 1, dtype=np.uint64)
        np.cumsum([len(o) for o in ops_list], out=ptr[1:])
        ops = np.array([o for ol in ops_list for o in ol], dtype=np.uint32)
        out = np.zeros(len(ops_list), dtype=np.uint8)
        rc = L.isocon_exon_filter_from_ops(ops.ctypes.data_as(_lib.u32p), ptr.ctypes.data_as(_lib.u64p), len(ops_list), mn, ig,
                                           out.ctypes.data_as(_lib.u8p))
        assert rc == 0
        exp = [F._flag_from_strings(a, b, mn, ig) for a, b in pairs]
        assert out.astype(bool).tolist() == exp, (mn, ig)


@pytest.mark.gpu
def test_gpu_alignments_then_filter_uses_cached_ops():
    from isocon_amd import SW_alignment_module as SWM
    from isocon_amd import functions as F
    from oracle import oracle as O
    g, aligned = _load()
    ed_in = {k1: {k2: O.ed_dp(k1, k2) for k2 in inner} for k1, inner in aligned.items()}
    got = SWM.sw_align_sequences(ed_in)
    assert got == aligned
    assert all(id(v) in SWM._OPS_CACHE for inner in got.values() for v in inner.values())
    for case in g["cases"]:
        work = {k1: dict(inner) for k1, inner in got.items()}
        filtered = F.filter_exon_differences(work, case["min_exon_diff"], case["ignore_ends_len"])
        assert sorted(filtered) == case["filtered"]
        assert [[k1, list(inner.keys())] for k1, inner in work.items()] == case["remaining"]
