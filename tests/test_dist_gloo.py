"""CPU, world_size 2, gloo: the sharded nearest-neighbour protocol (isocon_amd/dist.py) -- shard by lower index,
all_reduce(MIN) of best[], all_gather of the attaining edges, isocon_nn_finalize -- with the per-shard device work
replaced by an oracle-backed stand-in that emits the same (best, hits) contract as isocon_nn_partial."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

NN_INF = 0x3FFFFFFF


class FakeStore(object):
    """Same interface as isocon_amd.store.SeqStore.nn_partial; distances from the CPU oracle."""

    def __init__(self, seqs):
        self.seqs = seqs
        self.n = len(seqs)
        self.lens = np.array([len(s) for s in seqs], dtype=np.int64)

    def nn_partial(self, q_begin, q_end, phase, best, is_converged=None, is_target=None, depth=2 ** 32, q_stride=1, q_block=1, wide_queries=None):
        from oracle import oracle as O
        from isocon_amd.store import shard_entries
        owned = [int(x) for x in shard_entries(q_begin, min(q_end, self.n), q_stride, q_block)]
        hits = []
        n = self.n
        if is_target is not None:
            raise NotImplementedError
        conv = np.zeros(n, bool) if is_converged is None else np.asarray(is_converged, bool)
        if phase == 0:      # seed pass: nothing in this stand-in
            pass
        elif phase in (1, 3):    # pairs owned through their lower index, band limit 63
            for q in owned:
                for t in range(q + 1, n):
                    if self.lens[t] - self.lens[q] > 63:
                        break
                    d = O.ed_bounded(self.seqs[q], self.seqs[t], 63)
                    if d <= 0:
                        continue
                    for (e, o) in ((q, t), (t, q)):
                        if not conv[e] and d <= self.lens[e] and d <= best[e]:
                            best[e] = d
                            hits.append((e, o, d))
        else:               # owned queries still unresolved: unbounded distances inside |len diff| <= len(q)
            for q in owned:
                if conv[q] or best[q] != NN_INF:
                    continue
                for t in range(n):
                    if t == q or abs(self.lens[t] - self.lens[q]) > self.lens[q]:
                        continue
                    d = O.ed_bounded(self.seqs[q], self.seqs[t], int(self.lens[q]))
                    if 0 < d <= best[q]:
                        best[q] = d
                        hits.append((q, t, d))
        return np.array(hits, dtype=np.int32).reshape(-1, 3), {"phase": phase}


def _worker(rank, world, port, seqs, conv, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isocon_amd.dist import sharded_nn_graph
    best, row_ptr, cols = sharded_nn_graph(FakeStore(seqs), is_converged=conv, dist=dist, device=torch.device("cpu"))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), best=best, row_ptr=row_ptr, cols=cols)
    dist.barrier()
    dist.destroy_process_group()


def _worker_mismatch(rank, world, port, out_dir, sizes_differ=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isocon_amd.dist import sharded_nn_graph
    st = FakeStore(["ACGT", "ACGTA", "ACGTAC"] + (["ACGTACG", "ACGTACGT"] * rank if sizes_differ else []))
    st.fingerprint = 1234 + rank          # the ranks packed different orders / sets
    try:
        sharded_nn_graph(st, dist=dist, device=torch.device("cpu"))
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, "mismatch%d.txt" % rank), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("sizes_differ", [False, True])
def test_ranks_with_different_stores_are_rejected(tmp_path, sizes_differ):
    """The fingerprints are reduced while phase 0 runs and compared before the first collective whose size depends on n: ranks whose sets
    differ -- also in SIZE, where a reduction of best[] would pair tensors of different lengths -- all raise, none is left in a collective."""
    mp.spawn(_worker_mismatch, args=(2, _free_port(), str(tmp_path), sizes_differ), nprocs=2, join=True)
    for r in (0, 1):
        assert "fingerprint mismatch" in open(tmp_path / ("mismatch%d.txt" % r)).read()


def _worker_one_rank_fails(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isocon_amd.dist import sharded_nn_graph

    class Failing(FakeStore):
        def nn_partial(self, q_begin, q_end, phase, best, **kw):
            if rank == 1 and phase == 1:
                raise MemoryError("out of device memory (simulated)")
            return FakeStore.nn_partial(self, q_begin, q_end, phase, best, **kw)

    try:
        sharded_nn_graph(Failing(["ACGTACGT", "ACGTTCGT", "ACGTTCGTA", "ACGAACGTAA"]), dist=dist, device=torch.device("cpu"))
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, "fail%d.txt" % rank), "w").write(msg)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_a_failure_on_one_rank_raises_on_every_rank(tmp_path):
    """the failing rank's status rides on the min-reduction: nobody is left blocked in a collective"""
    mp.spawn(_worker_one_rank_fails, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert "phase 1 failed on another rank" in open(tmp_path / "fail0.txt").read()
    assert "phase 1 failed on this rank" in open(tmp_path / "fail1.txt").read()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_two_rank_sharded_graph_equals_serial_oracle(tmp_path):
    from isocon_amd import synth
    from oracle import oracle as O
    accs, seqs, _ = synth.make_reads(90, 150, 3, seed=9)
    seqs = sorted(dict.fromkeys(seqs), key=len)
    seqs.append("ACGT" * 60 + "TTGACCA")          # isolated entry: resolved in phase 1 only
    seqs = sorted(seqs, key=len)
    n = len(seqs)
    conv = np.zeros(n, np.uint8)
    conv[[3, 10]] = 1
    mp.spawn(_worker, args=(2, _free_port(), seqs, conv, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for k in ("best", "row_ptr", "cols"):
        assert r0[k].tolist() == r1[k].tolist()
    # serial oracle loop (the reference's semantics)
    row_ptr, cols, eds, _ = O.nn_1set(seqs, conv, 0, n)
    assert r0["row_ptr"].tolist() == row_ptr.tolist()
    assert r0["cols"].tolist() == cols.tolist()
    exp_best = [int(eds[row_ptr[i]]) if row_ptr[i + 1] > row_ptr[i] else -1 for i in range(n)]
    assert r0["best"].tolist() == exp_best


class FakePairStore(object):
    """SeqStore.ed_pairs / sg_trace contract with the CPU oracle behind it."""

    def __init__(self, seqs):
        self.seqs = seqs
        self.lens = np.array([len(s) for s in seqs], dtype=np.int64)

    def ed_pairs(self, a, b, k=None):
        from oracle import oracle as O
        return np.array([O.ed_bounded(self.seqs[int(x)], self.seqs[int(y)], -1 if k is None else int(k[i])) for i, (x, y) in enumerate(zip(a, b))], dtype=np.int32)

    def hw_pairs(self, q, t, k, **_unused):
        import sys
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import oracle as O
        from test_all_nn import hw_row
        return np.array([hw_row(O, self.seqs[int(x)], self.seqs[int(y)], int(kk)) for x, y, kk in zip(q, t, k)], dtype=np.int32).reshape(-1, 5)

    def sg_trace(self, a, b, mismatch, match=2, open_=2, ext=0, tie_policy=0, ed_upper=None):
        from oracle import oracle as O
        import re
        ops, ptr, res = [], [0], []
        code = {"=": 0, "X": 1, "I": 2, "D": 3}
        for i, (x, y) in enumerate(zip(a, b)):
            t = O.sg_trace(self.seqs[int(x)], self.seqs[int(y)], match, int(mismatch[i]), open_, ext, tie_policy)
            for ln, c in re.findall(r"(\d+)([=XID])", t["cigar"]):
                ops.append((int(ln) << 4) | code[c])
            ptr.append(len(ops))
            res.append([t["score"], t["end_query"], t["end_ref"], t["matches"], t["mismatches"], t["indels"]])
        return np.array(ops, dtype=np.uint32), np.array(ptr, dtype=np.int64), np.array(res, dtype=np.int32).reshape(-1, 6)


def _worker_pairs(rank, world, port, seqs, a, b, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from isocon_amd.dist import sharded_ed_pairs, sharded_hw_pairs, sharded_sg_trace
    st = FakePairStore(seqs)
    ed = sharded_ed_pairs(st, a, b, dist=dist, device=torch.device("cpu"))
    hw = sharded_hw_pairs(st, a, b, 12, dist=dist, device=torch.device("cpu"))
    ops, ptr, res = sharded_sg_trace(st, a, b, np.full(len(a), -2, np.int8), dist=dist, device=torch.device("cpu"))
    np.savez(os.path.join(out_dir, "pairs%d.npz" % rank), ed=ed, ops=ops, ptr=ptr, res=res, hw=hw)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_pair_lists_two_ranks(tmp_path):
    """Pair lists split round-robin by size over two ranks and gathered: same as one rank doing everything."""
    import random
    rng = random.Random(3)
    seqs = ["".join(rng.choice("ACGT") for _ in range(rng.randint(20, 90))) for _ in range(14)]
    seqs += [s[:10] + "A" + s[10:] for s in seqs[:6]]
    a = np.array([rng.randrange(len(seqs)) for _ in range(23)], dtype=np.uint32)
    b = np.array([rng.randrange(len(seqs)) for _ in range(23)], dtype=np.uint32)
    a = np.concatenate([a, np.arange(6, dtype=np.uint32)])              # each of the first six with its one-insertion variant
    b = np.concatenate([b, np.arange(14, 20, dtype=np.uint32)])
    port = _free_port()
    mp.spawn(_worker_pairs, args=(2, port, seqs, a, b, str(tmp_path)), nprocs=2, join=True)
    st = FakePairStore(seqs)
    ed = st.ed_pairs(a, b)
    ops, ptr, res = st.sg_trace(a, b, np.full(len(a), -2, np.int8))
    hw = st.hw_pairs(a, b, np.full(len(a), 12))
    assert (hw[:, 0] >= 0).sum() >= 3
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), "pairs%d.npz" % r))
        assert (z["hw"] == hw).all()
        assert (z["ed"] == ed).all() and (z["ptr"] == ptr).all() and (z["ops"] == ops).all() and (z["res"] == res).all()


def _worker_orders(rank, world, port, lst, out_dir):
    """every rank with its own order inside the groups of equal length: the drop-in module must still return, on each rank,
    the graph a single process would return for THAT rank's list (dict order included)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    import zlib
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import isocon_amd.nearest_neighbor_graph as NNG
    from oracle import oracle as O

    class OrderedFakeStore(FakeStore):
        def __init__(self, seqs):
            FakeStore.__init__(self, list(seqs))
            self.fingerprint = zlib.crc32("\n".join(seqs).encode())

        def close(self):
            pass

    NNG.SeqStore = OrderedFakeStore
    NNG.remember = lambda st, seqs: None
    if rank == 1:           # reverse every group of equal length
        groups = {}
        for item in lst:
            groups.setdefault(len(item[0]), []).append(item)
        lst = [item for L in sorted(groups) for item in reversed(groups[L])]

    class P(object):
        nr_cores = 1
        neighbor_search_depth = 2 ** 32
        verbose = False

    got = NNG.get_exact_nearest_neighbor_graph(lst, set(), P())
    exp = O.get_exact_nearest_neighbor_graph(lst, set(), P())
    same = [(k, list(v.items())) for k, v in got.items()] == [(k, list(v.items())) for k, v in exp.items()]
    open(os.path.join(out_dir, "orders%d.txt" % rank), "w").write("%s %d" % (same, sum(len(v) for v in got.values())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ranks_with_different_orders_of_equal_lengths(tmp_path):
    from isocon_amd import synth
    accs, seqs, _ = synth.make_reads(80, 110, 2, seed=21)
    lst = sorted({s: a for a, s in zip(accs, seqs)}.items(), key=lambda x: len(x[0]))
    assert len(set(len(s) for s, _ in lst)) < len(lst) - 20          # plenty of equal lengths
    mp.spawn(_worker_orders, args=(2, _free_port(), lst, str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        flag, edges = open(tmp_path / ("orders%d.txt" % r)).read().split()
        assert flag == "True" and int(edges) > 40
