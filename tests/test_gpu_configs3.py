"""GPU: BASELINE.json configs[3] -- the 50 000-read C3 set sharded over EIGHT ranks -- on the one GPU of the test box: every rank is a thread
with its own store (tests/baton_dist.py: the production protocol code of isocon_amd/dist.py, block-cyclic ownership `shard_of`, the
device-resident phases isocon_nn_partial_dev -> all_reduce(MIN) -> isocon_nn_hits_dev -> all_gather -> isocon_nn_finalize_dev; one rank runs at
a time, the collectives are done on the ranks' own device tensors).  The graph every rank returns must be the reference-loop fixture g17_c3
(digest asserted); the ranks' kernel times per phase, their max / mean and the emulated critical path go to the test log and to
gpurun_out/configs3_*.json.  The same for 20 000 reads of configs[4]'s shape (C5: ONT profile, 1-5 kb), whose queries need phase 2.
What the reference has instead: Pool chunking, modules/nearest_neighbor_graph.py:19-82."""
import json
import os

import numpy as np
import pytest

from baton_dist import run_ranks
from conftest import ROOT, g17

pytestmark = pytest.mark.gpu


def _sharded(seqs, world):
    import torch
    from isocon_amd.dist import sharded_nn_graph
    from isocon_amd.store import SeqStore
    torch.cuda.set_device(0)
    # every emulated rank needs what a real rank's process has for itself: its own scratch pool (bound matrix, held edges, counters)
    stores = [SeqStore(seqs, private_pool=True) for _ in range(world)]

    def rank_main(dist, rank):
        torch.cuda.set_device(0)
        laps = {}
        sharded_nn_graph(stores[rank], dist=dist, return_stats=True)          # warm-up: scratch pools, pinned buffers, the group's choices
        out = sharded_nn_graph(stores[rank], dist=dist, return_stats=True, laps=laps)
        return out, laps

    try:
        res, group = run_ranks(world, rank_main)
    finally:
        for s in stores:
            s.close()
    return res, group


def _report(name, world, res, group, single_kernel_ms):
    phases = max(len(r[0][3]) for r in res)
    per_phase = []
    for k in range(phases):
        kms = [float(r[0][3][k].get("kernel_ms", 0.0)) if k < len(r[0][3]) else 0.0 for r in res]
        per_phase.append({"phase": k, "per_rank_kernel_ms": kms, "max": max(kms), "mean": float(np.mean(kms)),
                          "max_over_mean": max(kms) / max(float(np.mean(kms)), 1e-9)})
    cols = [sum(float(st.get("cells_columns", 0)) for st in r[0][3]) for r in res]
    rep = {"workload": name, "ranks": world, "phases": per_phase, "critical_path_kernel_ms": sum(p["max"] for p in per_phase),
           "single_gpu_kernel_ms": single_kernel_ms, "summed_lane_columns": float(sum(cols)), "collectives_two_searches": group.collectives,
           "collective_bytes_two_searches": group.bytes_moved,
           "note": "ranks run one at a time on ONE GPU (threads, tests/baton_dist.py); per-rank kernel ms are HIP-event times of the rank's own "
                   "launches; the critical path is the sum over phases of the slowest rank -- an emulation, not a multi-GPU measurement"}
    print(json.dumps(rep))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "configs3_%s_%d.json" % (name, world)), "w") as f:
            json.dump(rep, f, indent=1)
    except OSError:
        pass
    return rep


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [8, 4, 2])
def test_c3_in_block_cyclic_shards_equals_the_reference_loop_fixture(world):
    import bench
    from isocon_amd.dist import shard_of
    from isocon_amd.store import SeqStore
    seqs, fbest, frow_ptr, fcols = g17("c3")
    n = len(seqs)
    assert shard_of(world - 1, world, n) == ((world - 1) * 256, n, world * 256, 256)          # blocks of one bound-tile row, dealt round-robin
    st = SeqStore(seqs)
    st.nn_graph()                                   # (warm-up: scratch pool, first-launch costs)
    single = st.nn_graph()
    st.close()
    assert bench.graph_digest(*single[:3]) == bench.EXPECTED_GRAPH_DIGEST_C3
    res, group = _sharded(seqs, world)
    for (best, row_ptr, cols, stats), laps in res:          # EVERY rank holds the whole graph
        assert (best == fbest).all() and (row_ptr == frow_ptr).all() and (cols == fcols).all()
        assert bench.graph_digest(best, row_ptr, cols) == bench.EXPECTED_GRAPH_DIGEST_C3
        assert len(stats) >= 2 and stats[0]["kernel_ms"] > 0 and stats[1]["kernel_ms"] > 0
    rep = _report("c3", world, res, group, float(single[3]["kernel_ms"]))
    assert rep["phases"][1]["max_over_mean"] < 1.5          # the block-cyclic deal balances the uneven windows
    assert rep["critical_path_kernel_ms"] < float(single[3]["kernel_ms"])


@pytest.mark.timeout(900)
def test_c5_shape_sharded_with_phase_2():
    """20 000 reads of configs[4]'s shape over 4 ranks: median NN distance in the hundreds, so the wide-band phase 2 runs on every rank"""
    from isocon_amd import synth
    from isocon_amd.store import SeqStore
    accs, seqs, _ = synth.make_reads(20000, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
    seqs = sorted(dict.fromkeys(seqs), key=len)
    st = SeqStore(seqs)
    want = st.nn_graph()
    st.close()
    assert np.median(want[0][want[0] >= 0]) > 63
    res, group = _sharded(seqs, 4)
    for (best, row_ptr, cols, stats), laps in res:
        assert (best == want[0]).all() and (row_ptr == want[1]).all() and (cols == want[2]).all()
        assert len(stats) == 3 and stats[2]["kernel_ms"] > 0          # phase 2 ran
    _report("c5_shape_20k", 4, res, group, float(want[3]["kernel_ms"]))
