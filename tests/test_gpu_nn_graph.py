"""GPU parity: nearest-neighbour graphs against fixtures produced by the REFERENCE's own NNG module (run under the
edlib shim, tests/golden/make_golden.py) -- dict content AND key order must match."""
import numpy as np
import pytest

from conftest import Params, golden, list_to_dd, ordered

pytestmark = pytest.mark.gpu


def test_1set_golden_cases():
    from isocon_amd import nearest_neighbor_graph as NNG
    for case in golden("g2_nn_graph_1set.json")["cases"]:
        S = dict(case["S"])
        p = Params(case["nr_cores"], case["depth"])
        graph, isolated = NNG.compute_nearest_neighbor_graph(S, set(case["has_converged"]), p)
        assert ordered(graph) == ordered(list_to_dd(case["graph"])), (case["nr_cores"], case["depth"])
        assert sorted(isolated) == case["isolated"]


def test_1set_reference_test_data_n200():
    from isocon_amd import nearest_neighbor_graph as NNG
    case = golden("g2_nn_graph_n200.json")
    graph, isolated = NNG.compute_nearest_neighbor_graph(dict(case["S"]), set(), Params(1))
    assert ordered(graph) == ordered(list_to_dd(case["graph"]))
    assert sorted(isolated) == case["isolated"]


def test_2set_golden_cases():
    from isocon_amd import nearest_neighbor_graph as NNG
    for case in golden("g2_nn_graph_2set.json")["cases"]:
        p = Params(case["nr_cores"], case["depth"])
        graph = NNG.compute_2set_nearest_neighbor_graph(dict(case["X"]), dict(case["C"]), p)
        assert ordered(graph) == ordered(list_to_dd(case["graph"])), (case["nr_cores"], case["depth"])


def test_1set_synthetic_vs_oracle_loop():
    """5 k x 1.5 kb is config C2; here a 1.2 k-read slice of it keeps the CPU oracle within seconds."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    accs, seqs, _ = synth.make_reads(1200, 1500, 3, seed=20001)
    S = dict(zip(accs, seqs))
    conv = set(seqs[:40])
    g_gpu, iso_gpu = NNG.compute_nearest_neighbor_graph(S, conv, Params(1))
    g_cpu, iso_cpu = O.compute_nearest_neighbor_graph(S, conv, Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)
    assert iso_gpu == iso_cpu


def test_1set_far_apart_queries_need_wide_bands():
    """Reads whose nearest neighbour is > 63 / > 511 edits away go through the 128..512-row bands and the un-banded
    kernel; same-length unrelated sequences exercise the early exit."""
    import random
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    rng = random.Random(3)
    nrng = np.random.Generator(np.random.PCG64(4))
    S = {}
    base = "".join(rng.choice("ACGT") for _ in range(1800))
    for i, rate in enumerate([0.0, 0.03, 0.05, 0.08, 0.12, 0.2, 0.3]):
        prof = dict(rate=rate, ins=0.4, dele=0.3, sub=0.3)
        for j in range(3):
            S["r%d_%d" % (i, j)] = synth.mutate(nrng, np.frombuffer(base.encode(), np.uint8), prof).tobytes().decode()
    for j in range(6):
        S["junk%d" % j] = "".join(rng.choice("ACGT") for _ in range(1790 + 4 * j))
    S["short"] = "ACGTAC"
    S["short2"] = "ACGAAC"
    g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)


def test_2set_synthetic_vs_oracle_loop():
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    accs, seqs, isoforms = synth.make_reads(800, 1500, 6, seed=77)
    X = dict(zip(accs, seqs))
    C = {"c%d" % i: s for i, s in enumerate(isoforms[:4])}     # reads of isoforms 4,5 have no close candidate
    C["exact"] = seqs[17]
    g_gpu = NNG.compute_2set_nearest_neighbor_graph(X, C, Params(1))
    g_cpu = O.compute_2set_nearest_neighbor_graph(X, C, Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)
    for depth in (1, 2):
        g_gpu = NNG.compute_2set_nearest_neighbor_graph(X, C, Params(1, depth))
        g_cpu = O.compute_2set_nearest_neighbor_graph(X, C, Params(1, depth))
        assert ordered(g_gpu) == ordered(g_cpu), depth


def test_sharded_partial_matches_single_call():
    """Two shards through isocon_nn_partial + min-reduce + isocon_nn_finalize == one isocon_nn_graph call."""
    from isocon_amd import _lib, synth
    from isocon_amd.store import SeqStore, nn_finalize
    accs, seqs, _ = synth.make_reads(900, 800, 4, seed=5)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    n = len(seqs)
    best1, rp1, cols1, _ = st.nn_graph()
    cut = n // 3
    from isocon_amd.dist import shard_of
    assert shard_of(1, 3, n)[3] > 1
    for shards in (((0, cut, 1, 1), (cut, n, 1, 1)),              # contiguous ranges
                   ((0, n, 3, 1), (1, n, 3, 1), (2, n, 3, 1)),    # entry-cyclic ownership
                   tuple(shard_of(r, 3, n) for r in range(3)),    # block-cyclic ownership (what dist.sharded_nn_graph uses)
                   ((0, n, 48, 16), (16, n, 48, 16), (32, n - 5, 48, 16))):
        for phases in ((0, 1, 2),      # seed, 64-row band, wide bands; min-reduction after each
                       (3, 2)):        # seeds + 64-row band in one call (what dist.sharded_nn_graph runs), wide bands
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


def test_long_reads_take_the_other_main_pass_kernels():
    """> 3.2 kb: 16-wave workgroups (window table > 53 KB); > 10 kb: scalar-window main pass.  Both against the oracle."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    for (n, L, seed) in ((220, 4600, 11), (36, 11000, 12)):
        accs, seqs, _ = synth.make_reads(n, L, 3, seed=seed)
        S = dict(zip(accs, seqs))
        g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
        g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
        assert ordered(g_gpu) == ordered(g_cpu), (n, L)


def test_empty_store_and_single_entry():
    from isocon_amd.store import SeqStore
    st = SeqStore([])
    best, rp, cols, _ = st.nn_graph()
    assert len(best) == 0 and rp.tolist() == [0] and len(cols) == 0
    st1 = SeqStore(["ACGTACGT"])
    best, rp, cols, _ = st1.nn_graph()
    assert best.tolist() == [-1] and rp.tolist() == [0, 0]


def test_tile_synchronous_main_pass_still_matches(monkeypatch):
    """ISOCON_DEBUG_VARIANT=nn_tiles selects the tile-synchronous LDS kernel (+ equal-length regrouping): the fallback for reads too
    long for the lane-refill kernel's LDS layout must stay exact."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "nn_tiles=1")
    accs, seqs, _ = synth.make_reads(700, 900, 4, seed=21)
    S = dict(zip(accs, seqs))
    g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)


@pytest.mark.parametrize("length,rate,lo,hi", [(650, 0.08, 64, 127), (1300, 0.08, 128, 255), (2600, 0.08, 256, 511)])
def test_wide_band_refill_kernels(length, rate, lo, hi):
    """NN distances in (63, 127], (127, 255], (255, 511]: the 128-, 256- and 512-row lane-refill kernels vs the oracle."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    prof = dict(synth.ONT_PROFILE, rate=rate)
    accs, seqs, _ = synth.make_reads(160, length, 2, seed=length, profile=prof)
    S = dict(zip(accs, seqs))
    g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)
    d = sorted(v for nb in g_cpu.values() for v in nb.values())
    assert lo <= d[len(d) // 2] <= hi, d[len(d) // 2]          # the case really sits in the intended band
    assert NNG.LAST_STATS["fallback_queries"] > 0


def test_wide_band_pass_in_bounded_launches(monkeypatch):
    """A wide-band pass is issued as launches of a bounded number of workgroups (8192 in production: <= ~2 s at 200 000 reads); here 7
    per launch on a small set, dense and one-workgroup-per-query forms: the same graph, many launches."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    accs, seqs, _ = synth.make_reads(160, 1300, 2, seed=1300, profile=dict(synth.ONT_PROFILE, rate=0.08))
    S = dict(zip(accs, seqs))
    ref, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    launches = NNG.LAST_STATS["scan_launches"]
    for variant in ("nn_wide_per_launch=7", "nn_wide_per_launch=7,nn_no_sparse"):
        monkeypatch.setenv("ISOCON_DEBUG_VARIANT", variant)
        got, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
        assert ordered(got) == ordered(ref)
        assert NNG.LAST_STATS["scan_launches"] > launches + 10
    monkeypatch.delenv("ISOCON_DEBUG_VARIANT")


def test_hit_list_overflow_reruns_with_the_bounds_kept(monkeypatch):
    """A hit list that is too small makes the phase run again (larger list, bounds kept): same graph."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "hits_cap=64")
    accs, seqs, _ = synth.make_reads(500, 700, 3, seed=33)
    S = dict(zip(accs, seqs))
    g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    assert NNG.LAST_STATS["scan_launches"] >= 2
    g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)
    prof = dict(synth.ONT_PROFILE, rate=0.08)          # the same in the wide-band phase
    accs, seqs, _ = synth.make_reads(120, 1300, 2, seed=34, profile=prof)
    S = dict(zip(accs, seqs))
    g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)


def test_2set_with_many_candidates_takes_the_scan_kernel():
    """More than 512 candidates: the reads-vs-candidates search runs as the generic upward scan with role flags."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    accs, seqs, _ = synth.make_reads(1500, 400, 4, seed=44)
    X = {a: s for a, s in list(zip(accs, seqs))[:900]}
    C = {"c_" + a: s for a, s in list(zip(accs, seqs))[900:]}
    assert len(set(C.values())) > 512
    g_gpu = NNG.compute_2set_nearest_neighbor_graph(X, C, Params(1))
    g_cpu = O.compute_2set_nearest_neighbor_graph(X, C, Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)


@pytest.mark.parametrize("first_w", [3, 5, 6, 7])
def test_every_band_width_in_steps_of_64_rows(first_w, monkeypatch):
    """192-, 320-, 384- and 448-row lane-refill kernels (k_nn_scan_refill<16, 3|5|6|7>) forced as the first stage on reads
    whose nearest neighbours lie 70 .. 500 edits away: queries beyond the forced band go on to 512 rows; same graph as the
    reference loop whatever the first width."""
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import synth
    from oracle import oracle as O
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "nn_first_w=%s" % (first_w,))
    seqs = []
    for length, rate, seed in ((900, 0.07, 1), (1800, 0.07, 2), (2600, 0.08, 3)):
        accs, s, _ = synth.make_reads(90, length, 2, seed=seed, profile=dict(synth.ONT_PROFILE, rate=rate))
        seqs += s
    S = {"r%d" % i: s for i, s in enumerate(seqs)}
    g_gpu, _ = NNG.compute_nearest_neighbor_graph(S, set(), Params(1))
    g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
    assert ordered(g_gpu) == ordered(g_cpu)
    d = sorted(v for nb in g_cpu.values() for v in nb.values())
    assert d[0] > 63 and d[len(d) // 4] <= 64 * first_w - 1 and d[-1] > 191      # the forced band resolves some queries; 192 rows not all


def test_sample_stage_and_one_workgroup_per_query_launches(monkeypatch):
    """Enough unresolved queries (>= 2048) for the sampled width selection, few enough leftovers for the launches with one
    workgroup per listed query (both directions): the graph equals the one of the plain launches, and sampled rows equal
    the reference loop."""
    from isocon_amd import synth
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    accs, seqs, _ = synth.make_reads(2600, 900, 3, seed=77, profile=dict(synth.ONT_PROFILE, rate=0.07))
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    best, row_ptr, cols, stats = st.nn_graph()
    assert stats["fallback_queries"] >= 2048
    monkeypatch.setenv("ISOCON_DEBUG_VARIANT", "nn_no_sparse=1")
    b2, r2, c2, _ = st.nn_graph()
    assert (b2 == best).all() and (r2 == row_ptr).all() and (c2 == cols).all()
    packed = O.pack(seqs)
    conv = np.zeros(len(seqs), np.uint8)
    for i in np.random.default_rng(5).choice(len(seqs), 40, replace=False).tolist():
        rp, c, e, _ = O.nn_1set(seqs, conv, i, 1, packed=packed)
        assert cols[row_ptr[i]:row_ptr[i + 1]].tolist() == c.tolist() and (e == best[i]).all(), i
    st.close()


def test_device_finalize_matches_the_host_routine():
    """The CSR of the graph built on the device (csrc/nn_finalize.hpp) against the host routine on the same hits: a set with many
    equidistant neighbours per read (star-shaped families of single substitutions), and one with a row too long for the device sort."""
    import os
    import random
    from isocon_amd.store import SeqStore
    rng = random.Random(77)
    for family, n_families in ((12, 400), (400, 6)):
        seqs = set()
        for f in range(n_families):
            L = rng.randrange(300, 900)
            root = "".join(rng.choice("ACGT") for _ in range(L))
            seqs.add(root)
            for _ in range(family):
                i = rng.randrange(L)
                seqs.add(root[:i] + rng.choice("ACGT".replace(root[i], "")) + root[i + 1:])
        seqs = sorted(seqs, key=len)
        assert len(seqs) >= 1024
        st = SeqStore(seqs)
        try:
            dev = st.nn_graph()
            os.environ["ISOCON_DEBUG_VARIANT"] = "nn_host_finalize=1"
            try:
                host = st.nn_graph()
            finally:
                del os.environ["ISOCON_DEBUG_VARIANT"]
            assert dev[3]["hits"] >= 4096
            assert all((x == y).all() for x, y in zip(dev[:3], host[:3])), (family, n_families)
            if family == 400:
                assert (np.diff(dev[1]) > 256).any()          # (the fallback was taken: a root with hundreds of neighbours at distance 1)
        finally:
            st.close()


def test_device_resident_phases_equal_the_single_call():
    """isocon_nn_partial_dev / _hits_dev / _finalize_dev (what dist.sharded_nn_graph runs on a GPU).  Three ranks emulated on one device:
    sweep 1 runs every phase on every rank and reduces the bounds with torch.minimum (the all_reduce); the library holds ONE edge list
    per process, so sweep 2 replays each rank's phases from the same reduced bounds with its list kept, filters it against the final
    bounds into a fixed-size block, and the blocks are concatenated like the all_gather."""
    import torch
    from isocon_amd import _lib, synth
    from isocon_amd.dist import shard_of
    from isocon_amd.store import SeqStore
    accs, seqs, _ = synth.make_reads(2500, 700, 4, seed=91)
    seqs = sorted(dict.fromkeys(seqs), key=len) + ["ACGT" * 40 + "TTTTGGGGCCCCAAAA" * 30]          # a read far from all others: phase 2
    seqs = sorted(seqs, key=len)
    conv = np.zeros(len(seqs), np.uint8); conv[::7] = 1
    st = SeqStore(seqs)
    try:
        n, world = st.n, 3
        want = st.nn_graph(is_converged=conv)
        dev = torch.device("cuda", 0)
        reduced = [torch.full((n,), _lib.NN_INF, dtype=torch.int32, device=dev)]          # bounds before phase 0, 1, 2 and the final ones
        for phase in (0, 1, 2):
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
            for phase in (0, 1, 2):
                b = reduced[phase].clone()
                qb, qe, qs, qk = shard_of(r, world, n)
                held, stats = st.nn_partial_dev(qb, qe, phase, b.data_ptr(), phase > 0, is_converged=conv, q_stride=qs, q_block=qk)
            assert held > 0
            with pytest.raises(RuntimeError):
                st.nn_hits_dev(final.data_ptr(), torch.empty((held - 1, 3), dtype=torch.int32, device=dev).data_ptr(), held - 1)
            blk = torch.empty((held + 5, 3), dtype=torch.int32, device=dev)
            st.nn_hits_dev(final.data_ptr(), blk.data_ptr(), held + 5)
            blocks.append(blk)
        gathered = torch.cat(blocks)
        assert int((gathered[:, 0] < 0).sum()) >= 5 * world          # the unused rows of the fixed-size blocks
        got = st.nn_finalize_dev(final.data_ptr(), gathered.data_ptr(), gathered.shape[0])
        assert all((x == y).all() for x, y in zip(got, want[:3]))
    finally:
        st.close()


@pytest.mark.parametrize("tables", ["nn_no_block_filter", "nn_table_chunks=0,nn_list_min=16"])
def test_narrow_mode_on_reads_with_few_errors(monkeypatch, tables):
    """Reads whose nearest neighbour is a few edits away: the list builder files nearly every pair under the 32-row class (thresholds
    <= 31), the few pairs above go to 64-row chunks or one per lane.  Same graph as with every pair on 64 rows (ISOCON_DEBUG_VARIANT=nn_narrow=0) and
    as with the pairs above 31 forced out of the lists (=1), and the rows of some reads against the reference loop.
    The table launches only run on what the block filter (csrc/nn_filter.hpp) leaves, and not at all when that is little: `tables` = without
    the filter / with it and chunks of any size kept for the tables; the graph is also the default path's (filter, survivors one per lane)."""
    from isocon_amd import synth
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    accs, seqs, _ = synth.make_reads(6000, 1100, 4, seed=515, profile=dict(synth.CCS_PROFILE, rate=0.003))
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    try:
        default = st.nn_graph()
        assert default[3]["pairs_block_rejected"] > 0
        monkeypatch.setenv("ISOCON_DEBUG_VARIANT", tables)
        got = st.nn_graph()
        assert all((x == y).all() for x, y in zip(got[:3], default[:3]))
        assert got[3]["pairs_narrow"] > 0 and got[3]["narrow_columns"] > 0 and got[3]["narrow_kernel_ms"] > 0, got[3]
        assert got[3]["narrow_kernel_ms"] <= got[3]["scan_kernel_ms"] and got[3]["narrow_columns"] <= got[3]["cells_columns"]
        monkeypatch.setenv("ISOCON_DEBUG_VARIANT", tables + ",nn_narrow=0")
        wide = st.nn_graph()
        monkeypatch.setenv("ISOCON_DEBUG_VARIANT", tables + ",nn_narrow=1")
        forced = st.nn_graph()
        monkeypatch.delenv("ISOCON_DEBUG_VARIANT")
        assert wide[3]["pairs_narrow"] == 0 and wide[3]["narrow_columns"] == 0 and wide[3]["pairs_wide_to_lanes"] == 0
        assert forced[3]["pairs_narrow"] > 0
        assert all((x == y).all() for x, y in zip(got[:3], wide[:3]))
        assert all((x == y).all() for x, y in zip(got[:3], forced[:3]))
        assert np.median(got[0][got[0] >= 0]) <= 31
        best, row_ptr, cols = got[:3]
        packed = O.pack(seqs)
        conv = np.zeros(st.n, np.uint8)
        for i in list(range(0, st.n, 397)) + [st.n - 1]:
            rp, c, e, _ = O.nn_1set(seqs, conv, i, 1, packed=packed)
            assert list(cols[row_ptr[i]:row_ptr[i + 1]]) == list(c[rp[0]:rp[1]]), i
            assert rp[1] == rp[0] or best[i] == e[rp[0]]
    finally:
        st.close()


def test_device_resident_entry_points_argument_checks_and_long_rows():
    """isocon_nn_finalize_dev on a graph with a row of hundreds of equidistant neighbours (the device sort declines, the host routine
    finishes from the downloaded buffers) equals isocon_nn_finalize on the same hits; null / short buffers are refused."""
    import random
    import torch
    from isocon_amd import _lib
    from isocon_amd.store import SeqStore, nn_finalize
    rng = random.Random(5)
    seqs = set()
    for f in range(5):
        L = rng.randrange(400, 700)
        root = "".join(rng.choice("ACGT") for _ in range(L))
        seqs.add(root)
        for _ in range(420):
            i = rng.randrange(L)
            seqs.add(root[:i] + rng.choice("ACGT".replace(root[i], "")) + root[i + 1:])
    seqs = sorted(seqs, key=len)
    st = SeqStore(seqs)
    try:
        n = st.n
        best = np.full(n, _lib.NN_INF, dtype=np.int32)
        hits = []
        for phase in (0, 1, 2):
            h, _ = st.nn_partial(0, n, phase, best)
            hits.append(h)
        hits = np.concatenate(hits)
        want = nn_finalize(n, best, hits)
        assert (np.diff(want[1]) > 256).any()
        dev = torch.device("cuda", 0)
        b_d = torch.from_numpy(best).to(dev)
        h_d = torch.from_numpy(np.ascontiguousarray(hits)).to(dev)
        got = st.nn_finalize_dev(b_d.data_ptr(), h_d.data_ptr(), len(hits))
        assert all((x == y).all() for x, y in zip(got, want))
        with pytest.raises(RuntimeError):
            st.nn_partial_dev(0, n, 1, 0, False)                       # no bounds buffer
        with pytest.raises(RuntimeError):
            st.nn_partial_dev(0, n, 7, b_d.data_ptr(), False)          # no such phase
        with pytest.raises(RuntimeError):
            st.nn_finalize_dev(b_d.data_ptr(), 0, 5)                   # rows announced, no buffer
    finally:
        st.close()


def test_few_close_pairs_plus_far_reads_keep_both_kinds_of_edge():
    """n >= 1024, the 64-row pass finds fewer than 4096 edges (they stay on the device, in the list the wide-band stages
    write to from its start again), and some reads only have neighbours 64..600 edits away: the edges of BOTH passes must
    reach the graph (NNG:155-178 for every row) -- compared with the host-finalize variant and with the oracle loop."""
    import os
    import random
    from isocon_amd import synth
    from isocon_amd.store import SeqStore
    from oracle import oracle as O
    rng = random.Random(91)
    nrng = np.random.Generator(np.random.PCG64(92))
    seqs = set()
    for f in range(520):          # twins 1-3 substitutions apart, every family at its own length
        L = 100 + 2 * f
        root = "".join(rng.choice("ACGT") for _ in range(L))
        twin = list(root)
        for _ in range(rng.randrange(1, 4)):
            i = rng.randrange(L)
            twin[i] = rng.choice("ACGT".replace(root[i], ""))
        seqs.add(root)
        seqs.add("".join(twin))
    far_base = "".join(rng.choice("ACGT") for _ in range(1900))
    for rate in (0.05, 0.09, 0.15):          # neighbours at ~ 90 / 170 / 280 edits: 128- to 512-row bands
        prof = dict(rate=rate, ins=0.4, dele=0.3, sub=0.3)
        for _ in range(3):
            seqs.add(synth.mutate(nrng, np.frombuffer(far_base.encode(), np.uint8), prof).tobytes().decode())
    for j in range(3):                       # unrelated reads of the same length: beyond 511
        seqs.add("".join(rng.choice("ACGT") for _ in range(1895 + 3 * j)))
    seqs = sorted(seqs, key=len)
    assert len(seqs) >= 1024
    st = SeqStore(seqs)
    try:
        dev = st.nn_graph()
        os.environ["ISOCON_DEBUG_VARIANT"] = "nn_host_finalize=1"
        try:
            host = st.nn_graph()
        finally:
            del os.environ["ISOCON_DEBUG_VARIANT"]
    finally:
        st.close()
    assert dev[3]["fallback_queries"] > 0
    assert all((x == y).all() for x, y in zip(dev[:3], host[:3]))
    best, row_ptr, cols = dev[:3]
    assert (np.diff(row_ptr)[best >= 0] > 0).all()          # a row with a bound has its edges
    S = {"r%d" % i: s for i, s in enumerate(seqs)}
    g_cpu, _ = O.compute_nearest_neighbor_graph(S, set(), Params(1))
    accs = list(S)
    got = {accs[i]: {accs[int(c)]: int(best[i]) for c in cols[row_ptr[i]:row_ptr[i + 1]]} for i in range(len(accs))}
    assert ordered(got) == ordered(g_cpu)


def test_wide_band_phase_in_sub_steps_with_reductions():
    """Phase 2 of the sharded protocol in sub-steps (dist.protocol_steps with phase2_steps = 4; the default is one step, see there): the
    rank's blocks dealt into sub-shards, best[] min-reduced after every sub-step, the phase's query set fixed at its start (wide_queries).  Three emulated ranks on reads whose neighbours are
    64..600 edits away plus a dense part: the same graph as the single call, and as phase 2 in one step."""
    from isocon_amd import _lib, synth
    from isocon_amd.dist import protocol_steps
    from isocon_amd.store import SeqStore, nn_finalize
    prof = dict(synth.ONT_PROFILE, rate=0.07)
    accs, seqs, _ = synth.make_reads(700, 1100, 3, seed=1907, profile=prof)
    accs2, seqs2, _ = synth.make_reads(600, 900, 2, seed=1908)
    seqs = sorted(dict.fromkeys(seqs + seqs2), key=len)
    st = SeqStore(seqs)
    try:
        n, world = st.n, 3
        want = st.nn_graph()
        assert want[3]["fallback_queries"] > 100
        lens = np.asarray(st.lens)[:n]
        best = np.full(n, _lib.NN_INF, dtype=np.int32)
        hits_all, wide, phase2_steps = [], None, 0
        for k in range(len(protocol_steps(0, world, n, 4))):
            phase = protocol_steps(0, world, n, 4)[k][0]
            if phase == 2 and wide is None:
                wide = ((best == _lib.NN_INF) & (lens > 63)).astype(np.uint8)
                assert wide.sum() > 100
            parts = []
            for r in range(world):
                ph, (qb, qe, qs, qk) = protocol_steps(r, world, n, 4)[k]
                b = best.copy()
                hits, stats = st.nn_partial(qb, qe, ph, b, q_stride=qs, q_block=qk, wide_queries=wide if ph == 2 else None)
                hits_all.append(hits); parts.append(b)
            best = np.minimum.reduce(parts)
            phase2_steps += phase == 2
        assert phase2_steps >= 2
        hits = np.concatenate(hits_all)
        got = nn_finalize(n, best, hits[(hits[:, 2] >= 0) & (hits[:, 2] == best[np.clip(hits[:, 0], 0, n - 1)])])
        assert all((x == y).all() for x, y in zip(got, want[:3]))
    finally:
        st.close()
