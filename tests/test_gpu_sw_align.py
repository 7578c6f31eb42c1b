"""GPU parity: semi-global affine alignment + traceback (isocon_sg_trace_batch) against the CPU oracle (all tie
policies) and against fixtures produced by the reference's SWM / GBA modules run under the parasail shim."""
import random

import numpy as np
import pytest

from conftest import Params, golden, list_to_dd, ordered

pytestmark = pytest.mark.gpu


def _rs(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


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


def _check_batch(pairs, mism, policy, match=2, open_=2, ext=0):
    from isocon_amd import SW_alignment_module as SWM
    from isocon_amd.edlib_alignment_module import _intern
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    seqs, a, b = _intern(pairs)
    st = SeqStore(seqs)
    ops, ptr, res = st.sg_trace(a, b, np.asarray(mism, dtype=np.int8), match=match, open_=open_, ext=ext, tie_policy=policy)
    for p, (s1, s2) in enumerate(pairs):
        exp = O.sg_trace(s1, s2, match, int(mism[p]), open_, ext, policy)
        got_cigar = SWM.ops_to_cigar(ops[ptr[p]:ptr[p + 1]].tolist())
        got = dict(cigar=got_cigar, score=int(res[p, 0]), end_query=int(res[p, 1]), end_ref=int(res[p, 2]),
                   matches=int(res[p, 3]), mismatches=int(res[p, 4]), indels=int(res[p, 5]))
        assert got == exp, (p, len(s1), len(s2), int(mism[p]), policy, got, exp)


def test_random_pairs_all_policies():
    rng = random.Random(11)
    pairs, mism = [], []
    for it in range(120):
        m = rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 200, 511, 512, 513, 700, 1100])
        s1 = _rs(rng, m)
        r = rng.random()
        if r < 0.6:
            s2 = _mut(rng, s1, rng.choice([0.01, 0.05, 0.2]))
        elif r < 0.8:
            s2 = _rs(rng, max(1, m + rng.randint(-40, 40)))
        else:
            cut = rng.randint(0, m - 1); ln = rng.randint(0, min(150, m - cut)); s2 = (s1[:cut] + s1[cut + ln:]) or "G"
        if rng.random() < 0.3:
            s1, s2 = s2, s1
        pairs.append((s1, s2)); mism.append(rng.choice([-1, -2, -4, -3]))
    for policy in (0, 1, 2, 3, 4, 8, 12, 16, 17, 31):
        _check_batch(pairs, mism, policy)
    _check_batch(pairs[:40], mism[:40], 0, open_=3, ext=1)     # hypothesis_test_module.py:99 scoring
    _check_batch(pairs[:40], mism[:40], 0, open_=3, ext=0)     # end_invariant_functions.py:22 scoring


def test_read_sized_pairs():
    """2.5 kb reads (config C3 shape): 5 row passes of the forward kernel, exon-sized gaps, homopolymers."""
    from isocon_amd import synth
    rng = random.Random(2)
    accs, seqs, isoforms = synth.make_reads(24, 2500, 4, seed=30001)
    pairs = [(isoforms[i % 4], seqs[i]) for i in range(24)] + [(seqs[0], seqs[1]), (isoforms[0], isoforms[1])]
    mism = [rng.choice([-1, -2, -4]) for _ in pairs]
    _check_batch(pairs, mism, 0)


def test_long_query_many_passes():
    rng = random.Random(8)
    s1 = _rs(rng, 5300)
    s2 = _mut(rng, s1, 0.06)
    _check_batch([(s1, s2), (s2, s1), (s1[:3000], s2)], [-2, -2, -4], 0)


def test_sw_align_sequences_golden():
    from isocon_amd import SW_alignment_module as SWM
    g = golden("g4_sw_align.json")
    assert g["tie_policy"] == SWM.TIE_POLICY
    for name in ("tie_free", "tie_heavy", "buckets"):
        matches = list_to_dd(g[name]["input"])
        for cores in ("1", "2"):
            got = SWM.sw_align_sequences(matches, nr_cores=int(cores))
            exp = {k1: {k2: (v[0], v[1], tuple(v[2])) for k2, v in inner.items()} for k1, inner in list_to_dd(g[name]["expected"][cores]).items()}
            assert ordered(got) == ordered(exp), name


def test_sw_align_sequences_keeping_accession_golden():
    from isocon_amd import SW_alignment_module as SWM
    g = golden("g4_sw_align.json")["keeping_accession"]
    matches = {a1: {a2: tuple(v) for a2, v in inner} for a1, inner in g["input"]}
    got = SWM.sw_align_sequences_keeping_accession(matches, nr_cores=1)
    exp = {k1: {k2: (v[0], v[1], tuple(v[2])) for k2, v in inner.items()} for k1, inner in list_to_dd(g["expected"]["1"]).items()}
    assert ordered(got) == ordered(exp)


def test_parasail_alignment_golden():
    from isocon_amd import SW_alignment_module as SWM
    for c in golden("g4_sw_align.json")["parasail_alignment"]:
        r = SWM.parasail_alignment(c["s1"], c["s2"], 0, 0, mismatch_penalty=c["mismatch_penalty"],
                                   opening_penalty=c["opening_penalty"], gap_ext=c["gap_ext"])
        e = c["expected"]
        assert (r[0], r[1]) == (e[0], e[1])
        assert (r[2][0], r[2][1], tuple(r[2][2])) == (e[2][0], e[2][1], tuple(e[2][2]))
    assert SWM.cigar_to_seq("2=1I1D1X", "ACGT", "ACTA") == ("ACG-T", "AC-TA")


def test_get_best_alignments_golden():
    from isocon_amd import get_best_alignments as GBA
    g = golden("gba_best_matches.json")
    approx = {k: v for k, v in g["approx"]}
    got = GBA.find_best_matches(approx, Params(1))
    exp = {k1: {k2: tuple(v) for k2, v in inner} for k1, inner in g["expected"]}
    assert ordered(got) == ordered(exp)
    paf = {k: [tuple(x) for x in v] for k, v in g["paf"]}
    got2 = GBA.find_best_matches_2set(paf, dict(g["X"]), dict(g["C"]), Params(1))
    exp2 = {k1: {k2: tuple(v) for k2, v in inner} for k1, inner in g["expected_2set"]}
    assert ordered(got2) == ordered(exp2)
    with pytest.raises(ZeroDivisionError):
        GBA.find_best_matches({}, Params(1))


def _check_banded(pairs, mism, policy, hints, match=2, open_=2, ext=0):
    """Results with an edit-distance hint must equal the full-matrix oracle whatever the hint is worth."""
    from isocon_amd import SW_alignment_module as SWM
    from isocon_amd.edlib_alignment_module import _intern
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    seqs, a, b = _intern(pairs)
    st = SeqStore(seqs)
    ops, ptr, res = st.sg_trace(a, b, np.asarray(mism, dtype=np.int8), match=match, open_=open_, ext=ext, tie_policy=policy,
                                ed_upper=np.asarray(hints, dtype=np.int32))
    for p, (s1, s2) in enumerate(pairs):
        exp = O.sg_trace(s1, s2, match, int(mism[p]), open_, ext, policy)
        got = dict(cigar=SWM.ops_to_cigar(ops[ptr[p]:ptr[p + 1]].tolist()), score=int(res[p, 0]), end_query=int(res[p, 1]),
                   end_ref=int(res[p, 2]), matches=int(res[p, 3]), mismatches=int(res[p, 4]), indels=int(res[p, 5]))
        assert got == exp, (p, len(s1), len(s2), int(mism[p]), policy, int(hints[p]), got, exp)


def test_banded_alignment_equals_full_matrix():
    """Exact hints (the band is used), hints that are far too small (certificate fails -> redone in full), absurdly
    large and missing hints; long pairs so that several passes and shifted windows are involved; exon-sized gaps."""
    from oracle import oracle as O
    rng = random.Random(23)
    pairs, mism, exact = [], [], []
    for it in range(60):
        m = rng.choice([300, 700, 1100, 1600, 2300])
        s1 = _rs(rng, m)
        r = rng.random()
        if r < 0.6:
            s2 = _mut(rng, s1, rng.choice([0.005, 0.02, 0.06]))
        elif r < 0.8:
            cut = rng.randint(0, m - 1); ln = rng.randint(20, min(300, m - cut)); s2 = (s1[:cut] + s1[cut + ln:]) or "G"
            s2 = _mut(rng, s2, 0.01)
        else:
            s2 = _mut(rng, s1[rng.randint(0, 40):m - rng.randint(0, 40)], 0.02)       # unequal ends: free end gaps
        if rng.random() < 0.5:
            s1, s2 = s2, s1
        pairs.append((s1, s2)); mism.append(rng.choice([-1, -2, -4])); exact.append(O.ed_bounded(s1, s2, -1))
    for policy in (0, 3, 8, 21):
        _check_banded(pairs, mism, policy, exact)
    _check_banded(pairs, mism, 0, [max(0, e // 6) for e in exact])            # bound violated: must fall back
    _check_banded(pairs, mism, 0, [0] * len(pairs))
    _check_banded(pairs, mism, 0, [e * 50 + 7 if i % 2 else -1 for i, e in enumerate(exact)])
    _check_banded(pairs, mism, 0, exact, match=2, open_=3, ext=1)


def test_wrappers_use_the_hint_and_stay_identical():
    """sw_align_sequences hands the input distances down as band hints: same output as the oracle's full alignment."""
    from isocon_amd import SW_alignment_module as SWM
    from oracle import oracle as O
    rng = random.Random(29)
    centre = _rs(rng, 1500)
    reads = [_mut(rng, centre, 0.01) for _ in range(30)] + [_mut(rng, centre[:700] + centre[900:], 0.01) for _ in range(4)]
    matches = {centre: {r: O.ed_bounded(centre, r, -1) for r in reads}}
    assert SWM.sw_align_sequences(matches) == O.sw_align_sequences(matches)


def test_diagonal_band_kernel_equals_strip_kernel(monkeypatch):
    """Certified bands of at most 256 diagonals run with the diagonals on the lanes (k_sg_band); ISOCON_DEBUG_VARIANT=sw_strips sends them
    through the strip kernel instead.  Read-sized pairs at CCS and at ONT error rates (bands of ~100 .. ~800 diagonals, so
    both kernels are in the default run), every tie policy class, both gap models: identical ops and results."""
    from isocon_amd import synth
    from isocon_amd.store import SeqStore
    rng = np.random.Generator(np.random.PCG64(9))
    seqs, a, b = [], [], []
    for it in range(48):
        L = int(rng.integers(900, 2700))
        base = synth._rand_seq(rng, L)
        rate = [0.004, 0.01, 0.02, 0.05][it % 4]
        x = synth.mutate(rng, base, dict(synth.CCS_PROFILE, rate=rate))
        y = synth.mutate(rng, base, dict(synth.ONT_PROFILE, rate=rate))
        if it % 5 == 0:
            y = y[int(rng.integers(0, 30)):len(y) - int(rng.integers(0, 30))]
        seqs += [x.tobytes().decode(), y.tobytes().decode()]
        a.append(2 * it); b.append(2 * it + 1)
    st = SeqStore(seqs)
    ed = st.ed_pairs(a, b, None)
    mm = np.array([[-1, -2, -4][i % 3] for i in range(len(a))], dtype=np.int8)
    for policy, open_, ext in ((0, 2, 0), (3, 2, 0), (8, 2, 0), (21, 2, 0), (0, 3, 1)):
        monkeypatch.delenv("ISOCON_DEBUG_VARIANT", raising=False)
        o1, p1, r1 = st.sg_trace(a, b, mm, open_=open_, ext=ext, tie_policy=policy, ed_upper=ed)
        monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "sw_strips=1")
        o2, p2, r2 = st.sg_trace(a, b, mm, open_=open_, ext=ext, tie_policy=policy, ed_upper=ed)
        monkeypatch.delenv("ISOCON_DEBUG_VARIANT", raising=False)
        o3, p3, r3 = st.sg_trace(a, b, mm, open_=open_, ext=ext, tie_policy=policy)          # no hints: the full matrix
        assert (r1 == r2).all() and (p1 == p2).all() and (o1 == o2).all(), (policy, open_, ext)
        assert (r1 == r3).all() and (p1 == p3).all() and (o1 == o3).all(), (policy, open_, ext)
    st.close()


def test_two_band_classes_give_the_same_alignments(monkeypatch):
    """Certified bands of at most 128 diagonals run with TWO diagonals per lane (k_sg_band<.., 2>: half the cells per step, half the trace),
    wider ones up to 256 with four; ISOCON_DEBUG_VARIANT=sw_band256 sends every pair through the four-per-lane form.  Pairs on both sides of
    the class boundary (bands of ~40 .. ~250 diagonals, length differences of both signs, ends cut), every tie policy class, both gap
    models: the two runs and the full matrix give identical ops and results, and both classes were in the default run."""
    from isocon_amd import synth
    from isocon_amd.store import SeqStore, sg_last_stats
    rng = np.random.Generator(np.random.PCG64(19))
    seqs, a, b = [], [], []
    for it in range(96):
        L = int(rng.integers(300, 2600))
        base = synth._rand_seq(rng, L)
        rate = [0.002, 0.004, 0.008, 0.012, 0.02, 0.03][it % 6]
        x = synth.mutate(rng, base, dict(synth.CCS_PROFILE, rate=rate))
        y = synth.mutate(rng, base, dict(synth.ONT_PROFILE, rate=rate))
        if it % 4 == 0:
            y = y[int(rng.integers(0, 20)):len(y) - int(rng.integers(0, 20))]
        if it % 7 == 0:
            x = x[int(rng.integers(0, 25)):]
        seqs += [x.tobytes().decode(), y.tobytes().decode()]
        if it % 2:
            a.append(2 * it); b.append(2 * it + 1)
        else:
            a.append(2 * it + 1); b.append(2 * it)
    st = SeqStore(seqs)
    ed = st.ed_pairs(a, b, None)
    mm = np.array([[-1, -2, -4][i % 3] for i in range(len(a))], dtype=np.int8)
    for policy, open_, ext in ((0, 2, 0), (3, 2, 0), (12, 2, 0), (21, 2, 0), (0, 3, 1)):
        monkeypatch.delenv("ISOCON_DEBUG_VARIANT", raising=False)
        o1, p1, r1 = st.sg_trace(a, b, mm, open_=open_, ext=ext, tie_policy=policy, ed_upper=ed)
        stats = sg_last_stats()
        assert 0 < stats["pairs_band_narrow"] < stats["pairs_band"], stats          # both classes
        monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "sw_band256")
        o2, p2, r2 = st.sg_trace(a, b, mm, open_=open_, ext=ext, tie_policy=policy, ed_upper=ed)
        assert sg_last_stats()["pairs_band_narrow"] == 0 and sg_last_stats()["trace_bytes"] > stats["trace_bytes"]
        monkeypatch.delenv("ISOCON_DEBUG_VARIANT", raising=False)
        o3, p3, r3 = st.sg_trace(a, b, mm, open_=open_, ext=ext, tie_policy=policy)
        assert (r1 == r2).all() and (p1 == p2).all() and (o1 == o2).all(), (policy, open_, ext)
        assert (r1 == r3).all() and (p1 == p3).all() and (o1 == o3).all(), (policy, open_, ext)
    st.close()


def test_narrow_tries_certify_themselves_or_run_again(monkeypatch):
    """A pair whose bound asks for 129 .. 256 diagonals runs on 128 first when that keeps >= 85 % of the half-width (two diagonals per lane
    cost half of four); k_sg_recheck evaluates the certificate for the band that was run and sends the failures through the band of their
    bound within the same batch.  Pairs at ~1.2-1.5 % errors with three error mixes -- insertions / deletions only (the narrow band certifies),
    substitutions only (the bound is tight: every try fails) and the CCS mix -- under the default threshold, under a threshold that tries
    every such pair (ISOCON_DEBUG_VARIANT=sw_narrow_try_pct=1), without tries, and without hints (full matrix): identical ops, results and
    gapped strings; the counters show tries that passed and tries that ran again."""
    from isocon_amd import synth
    from isocon_amd.store import SeqStore, sg_last_stats
    rng = np.random.Generator(np.random.PCG64(23))
    seqs, a, b = [], [], []
    mixes = (dict(ins=0.5, dele=0.5, sub=0.0), dict(ins=0.0, dele=0.0, sub=1.0), dict(ins=0.6875, dele=0.25, sub=0.0625))
    for it in range(90):
        L = int(rng.integers(2000, 2800))
        base = synth._rand_seq(rng, L)
        rate = [0.006, 0.007, 0.0075][it % 3]          # per read; the pair sees twice that
        mix = mixes[(it // 3) % 3]
        x = synth.mutate(rng, base, dict(mix, rate=rate))
        y = synth.mutate(rng, base, dict(mix, rate=rate))
        if it % 6 == 0:
            y = y[int(rng.integers(0, 12)):len(y) - int(rng.integers(0, 12))]
        seqs += [x.tobytes().decode(), y.tobytes().decode()]
        a.append(2 * it + (it & 1)); b.append(2 * it + 1 - (it & 1))
    st = SeqStore(seqs)
    ed = st.ed_pairs(a, b, None)
    mm = np.array([[-2, -1, -4][i % 3] for i in range(len(a))], dtype=np.int8)
    seen_pass = seen_again = 0
    for policy, open_, ext in ((0, 2, 0), (12, 2, 0), (21, 2, 0), (0, 3, 1)):
        monkeypatch.delenv("ISOCON_DEBUG_VARIANT", raising=False)
        o0, p0, r0 = st.sg_trace(a, b, mm, open_=open_, ext=ext, tie_policy=policy)          # no hints: the full matrix
        runs = {}
        for name, var in (("default", None), ("all", "sw_narrow_try_pct=1"), ("none", "sw_no_narrow_try")):
            if var is None:
                monkeypatch.delenv("ISOCON_DEBUG_VARIANT", raising=False)
            else:
                monkeypatch.setenv("ISOCON_DEBUG_VARIANT", var)
            o, p, r = st.sg_trace(a, b, mm, open_=open_, ext=ext, tie_policy=policy, ed_upper=ed)
            runs[name] = sg_last_stats()
            assert runs[name]["pairs_redone"] == 0, (name, runs[name])
            assert (r == r0).all() and (p == p0).all() and (o == o0).all(), (name, policy, open_, ext)
            sa, sb, sp, sr = st.sg_strings(a, b, mm, open_=open_, ext=ext, tie_policy=policy, ed_upper=ed)[:4]
            monkeypatch.delenv("ISOCON_DEBUG_VARIANT", raising=False)
            ra, rb, rp, rr = st.sg_strings(a, b, mm, open_=open_, ext=ext, tie_policy=policy)[:4]
            assert (sr == rr).all() and (sp == rp).all() and bytes(sa) == bytes(ra) and bytes(sb) == bytes(rb), (name, policy)
        assert runs["none"]["pairs_tried_narrow"] == 0 and runs["none"]["pairs_retried_wider"] == 0
        assert runs["all"]["pairs_tried_narrow"] >= runs["default"]["pairs_tried_narrow"] > 0, runs
        assert runs["all"]["pairs_retried_wider"] > 0, runs          # the substitution-only pairs
        assert runs["all"]["pairs_retried_wider"] < runs["all"]["pairs_tried_narrow"], runs          # the indel-only pairs
        assert runs["default"]["pairs_band_narrow"] > runs["none"]["pairs_band_narrow"], runs
        seen_pass += runs["default"]["pairs_tried_narrow"] - runs["default"]["pairs_retried_wider"]
        seen_again += runs["all"]["pairs_retried_wider"]
    assert seen_pass > 0 and seen_again > 0
    # hints that are too small: a band that is tried narrow, fails, runs again in the (too small) band of its bound, fails that certificate
    # on the host and is aligned in full -- still the full-matrix result
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "sw_narrow_try_pct=1")
    o0, p0, r0 = st.sg_trace(a, b, mm)
    redone = 0
    for f in (0.55, 0.7, 0.85):
        small = np.maximum((ed * f).astype(np.int32), 1)
        o, p, r = st.sg_trace(a, b, mm, ed_upper=small)
        stats = sg_last_stats()
        assert (r == r0).all() and (p == p0).all() and (o == o0).all(), f
        redone += stats["pairs_redone"]
        assert stats["pairs_retried_wider"] >= 0 and stats["pairs_tried_narrow"] >= 0
    assert redone > 0
    monkeypatch.delenv("ISOCON_DEBUG_VARIANT", raising=False)
    st.close()
