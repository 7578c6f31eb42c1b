"""CPU: the in-process rank harness of the configs[3] GPU test (tests/baton_dist.py) drives isocon_amd.dist.sharded_nn_graph at world sizes
4 and 8 with the oracle-backed stand-in store of test_dist_gloo.py (host-staged protocol: the stand-in has no device entry points): every rank
must return the graph of the reference loop, and a failing rank must fail all of them instead of leaving them in a collective."""
import numpy as np
import pytest
import torch

from baton_dist import run_ranks
from test_dist_gloo import FakeStore


def _seqs():
    from isocon_amd import synth
    accs, seqs, _ = synth.make_reads(90, 260, 3, seed=77)
    return sorted(dict.fromkeys(seqs), key=len) + ["ACGT" * 20 + "TTGGCCAA" * 30]          # one read far from the others: phase 2


@pytest.mark.parametrize("world", [4, 8])
def test_threads_as_ranks_return_the_reference_graph(world):
    from isocon_amd.dist import sharded_nn_graph
    from oracle import oracle as O
    seqs = sorted(_seqs(), key=len)
    n = len(seqs)
    conv = np.zeros(n, np.uint8)
    rp, cols, eds, _ = O.nn_1set(seqs, conv, 0, n)

    def rank_main(dist, rank):
        return sharded_nn_graph(FakeStore(seqs), is_converged=conv, dist=dist, device=torch.device("cpu"))

    res, group = run_ranks(world, rank_main, backend="gloo")
    assert group.collectives > 0
    for best, row_ptr, c in res:
        assert row_ptr.tolist() == rp.tolist() and c.tolist() == cols.tolist()
        assert [int(best[i]) for i in range(n) if rp[i + 1] > rp[i]] == [int(eds[rp[i]]) for i in range(n) if rp[i + 1] > rp[i]]


def test_a_failing_rank_fails_every_rank():
    from isocon_amd.dist import sharded_nn_graph
    seqs = sorted(_seqs(), key=len)

    class Failing(FakeStore):
        def nn_partial(self, *a, **k):
            raise MemoryError("rank 1 is out of memory")

    def rank_main(dist, rank):
        return sharded_nn_graph((Failing if rank == 1 else FakeStore)(seqs), dist=dist, device=torch.device("cpu"))

    with pytest.raises(RuntimeError, match="failed on"):
        run_ranks(3, rank_main, backend="gloo")
