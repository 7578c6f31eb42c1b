"""CPU: the C-ABI library loads and exports every symbol include/isocon_hip.h declares; host-side logic (finalize,
string conversions, sharding, bench numerator)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from isocon_amd import _lib
    header = open(os.path.join(ROOT, "include", "isocon_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(isocon_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.load()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.isocon_strerror(0) == b"ok"
    assert b"ACGT" in L.isocon_strerror(-2)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from isocon_amd import _lib
    from isocon_amd import edlib_alignment_module as EAM
    with pytest.raises(_lib.IsoconError):
        EAM.edlib_align_sequences({"ACGT": ["ACGA"]})
    # ... and so do the entry points built around the path: infix alignments, consensus correction
    import numpy as np
    from isocon_amd import correction_module as COR
    from isocon_amd import end_invariant_functions as END
    with pytest.raises(_lib.IsoconError):
        END.edlib_traceback("ACGTACGT", "TTACGTTCGTAA", mode="HW", task="path", k=3, end_threshold=1)
    with pytest.raises(_lib.IsoconError):
        COR._correct_on_device(np.frombuffer(b"ACGTACGA", dtype=np.uint8).reshape(2, 4), np.array([1, 1]))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "isocon_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libisocon_oracle" not in text and "#include \"../../oracle" not in text, f


def test_nn_finalize_order_and_filter():
    from isocon_amd.store import nn_finalize
    n = 6
    best = np.array([3, 2, 0x3FFFFFFF, 5, 2, 1], dtype=np.int32)
    hits = np.array([
        [0, 4, 3], [0, 1, 3], [0, 2, 7],      # 0: two minima, offsets 4 and 1 -> order (1, 4); the 7 is dropped
        [1, 0, 2], [1, 2, 2], [1, 2, 2],      # 1: lower neighbour before upper at equal offset; duplicate removed
        [3, 5, 5], [3, 1, 5],                 # 3: offsets 2 (up) and 2 (down) -> down (1) first
        [4, 3, 9],                            # 4: no hit attains best -> empty row, best reported as -1
        [5, 4, 1],
    ], dtype=np.int32)
    out_best, row_ptr, cols = nn_finalize(n, best, hits)
    rows = [cols[row_ptr[i]:row_ptr[i + 1]].tolist() for i in range(n)]
    assert rows == [[1, 4], [0, 2], [], [1, 5], [], [4]]
    assert out_best.tolist() == [3, 2, -1, 5, -1, 1]


def test_cigar_and_penalty_helpers():
    from isocon_amd import SW_alignment_module as SWM
    assert SWM.cigar_to_seq("3=2I1X2D", "ACGTTA", "ACGGCC") == ("ACGTTA--", "ACG--GCC")
    assert SWM._ops_to_alignment([(3 << 4) | 0, (2 << 4) | 2, (1 << 4) | 1, (2 << 4) | 3], "ACGTTA", "ACGGCC") == ("ACGTTA--", "ACG--GCC")
    assert SWM.ops_to_cigar([(3 << 4) | 0, (2 << 4) | 2, (1 << 4) | 1, (2 << 4) | 3]) == "3=2I1X2D"
    s = "A" * 200
    assert SWM._penalty(2, s, s) == -1 and SWM._penalty(3, s, s) == -2        # 0.01 boundary (SWM:104)
    assert SWM._penalty(18, s, s) == -2 and SWM._penalty(19, s, s) == -4      # 0.09 boundary (SWM:106)
    with pytest.raises(SystemExit):
        SWM.cigar_to_seq("3M", "ACG", "ACG")


def test_shard_ranges_cover_and_balance():
    from isocon_amd.dist import shard_ranges
    rng = np.random.default_rng(0)
    lens = np.sort(np.concatenate([rng.integers(1000, 1010, 3000), rng.integers(2000, 2400, 500)]))
    for world in (1, 2, 3, 8):
        r = shard_ranges(lens, world)
        assert r[0][0] == 0 and r[-1][1] == len(lens)
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
    r2 = shard_ranges(lens, 2)
    assert r2[0][1] < 1500          # the dense low-length cluster is cut well before its middle (upward windows)
    tgt = np.zeros(len(lens), bool); tgt[::500] = True
    r = shard_ranges(lens, 4, two_set_targets=tgt)
    assert r[0][0] == 0 and r[-1][1] == len(lens)


def test_bench_window_pairs_numerator():
    import bench
    lens = np.array([10, 10, 11, 15, 30])
    best = np.array([1, 0, 5, -1, 40])
    # q0: |len diff|<=1 -> {1,2} ; q1: <=0 -> {0}; q2: <=5 -> {0,1,3}; q3: none; q4: <=40 -> all 4
    assert bench.window_pairs(lens, best) == 2 + 1 + 3 + 0 + 4


def test_synth_is_deterministic():
    from isocon_amd import synth
    a = synth.make_reads(50, 300, 3, seed=5)
    b = synth.make_reads(50, 300, 3, seed=5)
    assert a == b and len(a[1]) == 50
    assert set("".join(a[1])) <= set("ACGT")
