"""GPU: sets with MORE than four distinct symbols (ACGT + N, mixed case, arbitrary bytes).  edlib compares whatever characters it is
given (EAM:111, NNG:105): distances must equal the textbook DP over the bytes (oracle orc_ed_dp) and the nearest-neighbour graphs the
oracle loop, for every mix of ordinary sequences (bit-vector kernels on the 2-bit planes) and sequences that hold other symbols (byte-wise
kernel, csrc/ed_bytes.hpp).  Alignments / consensus / infix entry points refuse such a set."""
import random

import numpy as np
import pytest

from conftest import Params, ordered

pytestmark = pytest.mark.gpu


def _family(rng, alphabet, n, L, max_edits, extra="", extra_rate=0.0):
    """n noisy copies of one random root over `alphabet`; every copy gets symbols of `extra` at rate extra_rate per base."""
    root = [rng.choice(alphabet) for _ in range(L)]
    out = []
    for _ in range(n):
        s = list(root)
        for _ in range(rng.randrange(0, max_edits + 1)):
            i = rng.randrange(len(s))
            r = rng.random()
            if r < 0.4:
                s[i] = rng.choice(alphabet)
            elif r < 0.7:
                del s[i]
            else:
                s.insert(i, rng.choice(alphabet))
        if extra and extra_rate:
            for i in range(len(s)):
                if rng.random() < extra_rate:
                    s[i] = rng.choice(extra)
        out.append("".join(s))
    return out


def test_pair_distances_over_any_bytes():
    from isocon_amd import _lib
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = random.Random(21)
    sets = {
        "rare N": _family(rng, "ACGT", 30, 300, 12) + _family(rng, "ACGT", 6, 300, 12, "N", 0.01) + ["", "N", "NN", "A", "ACGTN" * 30],
        "mixed case": _family(rng, "ACGTacgt", 30, 200, 10) + _family(rng, "ACGT", 10, 200, 10) + [""],
        "iupac and gaps": _family(rng, "ACGTNRYKM-", 25, 130, 30) + _family(rng, "ACGTNRYKM-", 10, 700, 80),
        "every byte": ["".join(chr(rng.randrange(1, 128)) for _ in range(rng.randrange(0, 260))) for _ in range(30)],
    }
    for name, seqs in sets.items():
        st = SeqStore(seqs)
        try:
            a = [rng.randrange(len(seqs)) for _ in range(400)]
            b = [rng.randrange(len(seqs)) for _ in range(400)]
            want = [O.ed_dp(seqs[x], seqs[y]) for x, y in zip(a, b)]
            assert st.ed_pairs(a, b, None).tolist() == want, name
            for kmax in (4, 40, 70, 300, 5000):
                k = [rng.randrange(0, kmax) for _ in range(400)]
                assert st.ed_pairs(a, b, k).tolist() == [d if d <= kk else -1 for d, kk in zip(want, k)], (name, kmax)
            with pytest.raises(_lib.IsoconError) as e:
                st.sg_trace(a[:4], b[:4], -2)
            assert "ACGT" in str(e.value)
            with pytest.raises(_lib.IsoconError) as e:
                st.hw_pairs(a[:4], b[:4], [5] * 4)
            assert "four distinct symbols" in str(e.value)
        finally:
            st.close()


def test_edlib_wrappers_on_reads_with_n():
    from isocon_amd import edlib_alignment_module as EAM
    from oracle import oracle as O
    rng = random.Random(5)
    seqs = _family(rng, "ACGT", 12, 400, 20) + _family(rng, "ACGT", 5, 400, 20, "Nn", 0.02)
    matches = {s: [t for t in rng.sample(seqs, 5) if t != s] for s in seqs}
    got = EAM.edlib_align_sequences(matches)
    want = {s: {t: O.ed_dp(s, t) for t in ts} for s, ts in matches.items() if ts}
    assert got == want


@pytest.mark.parametrize("seed,extra,rate", [(1, "N", 0.004), (2, "Nn", 0.02), (3, "acgt", 0.3), (4, "N", 0.0)])
def test_1set_graph_with_other_symbols(seed, extra, rate):
    """Families of near reads (neighbours within 63: the 64-row phase), families whose members are 64..500 apart and unrelated reads (the wide
    bands and the un-banded stage), some of each with symbols outside ACGT; converged entries; a finite search depth."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from oracle import oracle as O
    rng = random.Random(seed)
    seqs = []
    for L in (180, 300, 301, 420):
        seqs += _family(rng, "ACGT", 14, L, 10)
        seqs += _family(rng, "ACGT", 5, L, 10, extra, rate)
    seqs += _family(rng, "ACGT", 5, 600, 200) + _family(rng, "ACGT", 4, 600, 200, extra, rate)      # members ~100-300 apart
    seqs += ["".join(rng.choice("ACGT") for _ in range(590 + 3 * j)) for j in range(4)]               # unrelated: > 300 apart
    seqs += ["".join(rng.choice("ACGT" + extra) for _ in range(585 + 5 * j)) for j in range(3)]
    seqs += ["ACGTAC", "ACGNAC", "N", "NA"]
    if rate == 0.0:
        seqs += ["ACGTACGTNN"]          # a single exceptional entry in an otherwise ordinary set
    seqs = list(dict.fromkeys(seqs))
    S = {"r%d" % i: s for i, s in enumerate(seqs)}
    conv = {"r3", "r17", "r%d" % (len(seqs) - 9)}
    for depth in (None, 7):
        params = Params(1) if depth is None else Params(1, depth)
        for converged in (set(), conv):
            g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, converged, params)
            g_cpu, _ = O.compute_nearest_neighbor_graph(S, converged, params)
            assert ordered(g_gpu) == ordered(g_cpu), (depth, sorted(converged))


def test_2set_graph_with_other_symbols():
    from isocon_amd import nearest_neighbor_graph as NNG
    from oracle import oracle as O
    rng = random.Random(9)
    roots = _family(rng, "ACGT", 4, 350, 120)
    X, C = {}, {}
    for r, root in enumerate(roots):
        for j in range(12):
            s = list(root)
            for _ in range(rng.randrange(0, 9)):
                s[rng.randrange(len(s))] = rng.choice("ACGT")
            if j % 4 == 0:
                s[rng.randrange(len(s))] = "N"
            X["x%d_%d" % (r, j)] = "".join(s)
        C["c%d" % r] = root if r % 2 else root[:100] + "N" + root[101:]
    C["far"] = "".join(rng.choice("ACGTN") for _ in range(340))
    X["lonely"] = "".join(rng.choice("ACGT") for _ in range(355))
    g_gpu = NNG.compute_2set_nearest_neighbor_graph(X, C, Params(1))
    g_cpu = O.compute_2set_nearest_neighbor_graph(X, C, Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)


def test_sharded_graph_with_other_symbols():
    """isocon_nn_partial over block-cyclic shards + min-reduction + isocon_nn_finalize == the single call, with byte-wise pairs in
    both phases."""
    from isocon_amd import _lib
    from isocon_amd.store import SeqStore, nn_finalize
    rng = random.Random(33)
    seqs = []
    for L in (200, 320, 321):
        seqs += _family(rng, "ACGT", 40, L, 10) + _family(rng, "ACGT", 8, L, 10, "N", 0.01)
    seqs += _family(rng, "ACGT", 6, 500, 160) + _family(rng, "ACGT", 4, 500, 160, "N", 0.01)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    try:
        n = len(seqs)
        best1, rp1, cols1, stats = st.nn_graph()
        assert stats["pairs_bytes"] > 0
        for shards in (((0, n, 3, 1), (1, n, 3, 1), (2, n, 3, 1)), ((0, n, 32, 16), (16, n, 32, 16))):
            for phases in ((0, 1, 2), (3, 2)):
                hits = []
                red = np.full(n, _lib.NN_INF, dtype=np.int32)
                for phase in phases:
                    bests = []
                    for (b, e, stride, block) in shards:
                        best = red.copy()
                        h, _ = st.nn_partial(b, e, phase, best, q_stride=stride, q_block=block)
                        bests.append(best); hits.append(h)
                    red = np.minimum.reduce(bests)
                best2, rp2, cols2 = nn_finalize(n, red, np.concatenate(hits))
                assert best1.tolist() == best2.tolist()
                assert rp1.tolist() == rp2.tolist() and cols1.tolist() == cols2.tolist()
    finally:
        st.close()


def test_device_resident_phases_with_other_symbols():
    """isocon_nn_partial_dev / _hits_dev / _finalize_dev (the path dist.sharded_nn_graph runs on a GPU) over a set with exceptional
    sequences: three ranks emulated on one device as in tests/test_gpu_nn_graph.py, byte-wise pairs in the main and in the wide phase."""
    import torch
    from isocon_amd import _lib, synth
    from isocon_amd.dist import shard_of
    from isocon_amd.store import SeqStore
    rng = random.Random(8)
    accs, seqs, _ = synth.make_reads(2000, 600, 4, seed=17)
    seqs = list(dict.fromkeys(seqs))
    for i in rng.sample(range(len(seqs)), 60):
        p = rng.randrange(len(seqs[i]))
        seqs[i] = seqs[i][:p] + "N" + seqs[i][p + 1:]
    seqs += ["ACGT" * 40 + "TTTTGGGGCCCCAAAA" * 30, "ACGT" * 41 + "NTTTGGGGCCCCAAAA" * 30]          # far from all others: phase 2, one of them exceptional
    seqs = sorted(dict.fromkeys(seqs), key=len)
    conv = np.zeros(len(seqs), np.uint8); conv[::9] = 1
    st = SeqStore(seqs)
    try:
        n, world = st.n, 3
        want = st.nn_graph(is_converged=conv)
        assert want[3]["pairs_bytes"] > 0
        dev = torch.device("cuda", 0)
        reduced = [torch.full((n,), _lib.NN_INF, dtype=torch.int32, device=dev)]
        for phase in (3, 2):
            outs = []
            for r in range(world):
                b = reduced[-1].clone()
                qb, qe, qs, qk = shard_of(r, world, n)
                st.nn_partial_dev(qb, qe, phase, b.data_ptr(), False, is_converged=conv, q_stride=qs, q_block=qk)
                outs.append(b)
            reduced.append(torch.stack(outs).min(dim=0).values)
        final = reduced[-1]
        blocks = []
        for r in range(world):
            held = 0
            for i, phase in enumerate((3, 2)):
                b = reduced[i].clone()
                qb, qe, qs, qk = shard_of(r, world, n)
                held, _ = st.nn_partial_dev(qb, qe, phase, b.data_ptr(), i > 0, is_converged=conv, q_stride=qs, q_block=qk)
            blk = torch.empty((held + 1, 3), dtype=torch.int32, device=dev)
            st.nn_hits_dev(final.data_ptr(), blk.data_ptr(), held + 1)
            blocks.append(blk)
        gathered = torch.cat(blocks)
        got = st.nn_finalize_dev(final.data_ptr(), gathered.data_ptr(), gathered.shape[0])
        assert all((x == y).all() for x, y in zip(got, want[:3]))
        # and the graph is the oracle's
        from isocon_amd import nearest_neighbor_graph as NNG
        from oracle import oracle as O
        S = {"r%d" % i: s for i, s in enumerate(seqs[::3])}
        assert ordered(NNG.compute_nearest_neighbor_graph(S, set(), Params(1))[0]) == ordered(O.compute_nearest_neighbor_graph(S, set(), Params(1))[0])
    finally:
        st.close()


@pytest.mark.parametrize("seed,frac,mode", [(1, 0.1, "N"), (2, 0.02, "N"), (3, 1.0, "mask"), (4, 0.3, "lower")])
def test_thousands_of_reads_with_other_symbols(seed, frac, mode):
    """3 000 reads: the survivor lists, both table kernels and the device-side hit filter take part (small sets never reach them).  Reads with
    a few 'N', a soft-masked copy of the whole set (one motif in lower case wherever it occurs: every read exceptional, image distances mostly
    equal to the true ones) and random lower-case patches (image distances far below the true ones: most queries need the collecting pass)."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    accs, seqs, _ = synth.make_reads(3000, 700, 4, 1000 + seed)
    rng = random.Random(seed)
    seqs = list(dict.fromkeys(seqs))
    if mode == "N":
        for i in rng.sample(range(len(seqs)), int(frac * len(seqs))):
            for _ in range(rng.randrange(1, 4)):
                p = rng.randrange(len(seqs[i]))
                seqs[i] = seqs[i][:p] + "N" + seqs[i][p + 1:]
    elif mode == "mask":
        seqs = [s.replace("AACA", "aaca") for s in seqs]
    else:
        for i in rng.sample(range(len(seqs)), int(frac * len(seqs))):
            a = rng.randrange(len(seqs[i]) - 30)
            seqs[i] = seqs[i][:a] + seqs[i][a:a + 12].lower() + seqs[i][a + 12:]
    S = {"r%d" % i: s for i, s in enumerate(dict.fromkeys(seqs))}
    g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    assert NNG.LAST_STATS["pairs_bytes"] > 0
    g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(8))
    assert ordered(g_gpu) == ordered(g_cpu)


@pytest.mark.parametrize("variant", ["", "nn_tiles=1"])
def test_other_symbols_through_the_fallback_main_pass_kernels(variant, monkeypatch):
    """Reads too long for the lane-refill kernel's LDS layout (16-wave tables, scalar-window scan) and the tile-synchronous kernel
    (ISOCON_DEBUG_VARIANT=nn_tiles: re-run markers resolved on the host): the image passes run through them too."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    if variant:
        monkeypatch.setenv("ISOCON_DEBUG_VARIANT", variant)
    rng = random.Random(3)
    for (n, L, seed) in ((200, 4600, 11), (30, 11000, 12), (500, 900, 21)):
        accs, seqs, _ = synth.make_reads(n, L, 3, seed=seed)
        seqs = list(dict.fromkeys(seqs))
        for i in rng.sample(range(len(seqs)), len(seqs) // 5):
            p = rng.randrange(len(seqs[i]))
            seqs[i] = seqs[i][:p] + rng.choice(["N", "n", seqs[i][p].lower()]) + seqs[i][p + 1:]
        S = {"r%d" % i: s for i, s in enumerate(dict.fromkeys(seqs))}
        g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
        g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
        assert ordered(g_gpu) == ordered(g_cpu), (variant, n, L)


@pytest.mark.parametrize("seed,frac,mode,lr,n", [(1, 0.05, "N", (500, 900), 1500), (2, 0.3, "lower", (500, 900), 1500), (3, 1.0, "mask", (500, 900), 1500),
                                                 (4, 0.1, "N", (2500, 4500), 500), (5, 0.5, "lower", (5000, 9000), 150)])
def test_wide_phase_with_other_symbols(seed, frac, mode, lr, n):
    """ONT-profile reads (nearest neighbours 60 .. 640 edits away: the 128- to 512-row bands and the un-banded stage) with 'N's, random
    lower-case patches or a soft-masked motif, some entries converged: the image passes of the wide phase against the oracle loop."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    accs, seqs, _ = synth.make_reads(n, 0, 12, 7000 + seed, profile=synth.ONT_PROFILE, families=3, length_range=lr)
    rng = random.Random(seed)
    seqs = list(dict.fromkeys(seqs))
    if mode == "N":
        for i in rng.sample(range(len(seqs)), int(frac * len(seqs))):
            for _ in range(rng.randrange(1, 4)):
                p = rng.randrange(len(seqs[i]))
                seqs[i] = seqs[i][:p] + "N" + seqs[i][p + 1:]
    elif mode == "mask":
        seqs = [s.replace("AACA", "aaca") for s in seqs]
    else:
        for i in rng.sample(range(len(seqs)), int(frac * len(seqs))):
            a = rng.randrange(len(seqs[i]) - 30)
            seqs[i] = seqs[i][:a] + seqs[i][a:a + 12].lower() + seqs[i][a + 12:]
    S = {"r%d" % i: s for i, s in enumerate(dict.fromkeys(seqs))}
    conv = set(rng.sample(sorted(S), 5))
    g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, conv, Params(1))
    assert NNG.LAST_STATS["pairs_bytes"] > 0 and NNG.LAST_STATS["fallback_queries"] > 0
    g_cpu, _ = O.compute_nearest_neighbor_graph(S, conv, Params(8))
    assert ordered(g_gpu) == ordered(g_cpu)
