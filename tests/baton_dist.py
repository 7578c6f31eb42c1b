"""A process group of N ranks inside ONE process, for exercising isocon_amd.dist's protocol code at world sizes the test box has no GPUs
for (BASELINE.json configs[3]: 8 ranks): every rank is a thread with its own SeqStore (its own scratch pool, bound matrix and held
edge list -- like a real rank's process), and a `torch.distributed`-shaped object per rank implements the collectives the protocol
uses (all_reduce MIN / MAX / SUM, all_gather_into_tensor, all_gather, barrier) over the ranks' own tensors.  A BATON serialises
everything between collectives: exactly one rank runs at a time, so the ranks' kernels never overlap on the shared GPU and a rank's
HIP-event times are what the rank alone costs.  The code under test is the production path (dist.sharded_nn_graph ->
_nn_graph_device_resident: isocon_nn_partial_dev -> reduce -> isocon_nn_hits_dev -> gather -> isocon_nn_finalize_dev); only the
transport of the collectives differs from RCCL."""
import threading
import types


class BatonGroup(object):
    def __init__(self, world, backend="nccl"):
        self.world, self.backend = world, backend
        self.barrier = threading.Barrier(world)
        self.baton = threading.Lock()
        self.slots = [None] * world
        self.collectives = 0
        self.bytes_moved = 0

    def rank_handle(self, rank):
        return BatonDist(self, rank)


class BatonDist(object):
    """what isocon_amd.dist needs of torch.distributed, for one rank of a BatonGroup"""
    ReduceOp = types.SimpleNamespace(MIN="min", MAX="max", SUM="sum")

    def __init__(self, group, rank):
        self.g, self.rank, self.holding = group, rank, False

    def take(self):
        self.g.baton.acquire()
        self.holding = True

    def give(self):
        self.holding = False
        self.g.baton.release()

    def get_world_size(self):
        return self.g.world

    def get_rank(self):
        return self.rank

    def get_backend(self):
        return self.g.backend

    # the calling thread holds the baton whenever it is outside a collective
    def _collective(self, mine, root_fn):
        g = self.g
        g.slots[self.rank] = mine
        self.give()
        g.barrier.wait()
        if self.rank == 0:
            with g.baton:
                root_fn(list(g.slots))
                g.collectives += 1
        g.barrier.wait()
        self.take()

    def all_reduce(self, t, op="sum", async_op=False):
        """async_op: the reduction is done at once all the same (the ranks meet here); the handle's wait() has nothing left to do"""
        import torch

        def root(slots):
            st = torch.stack([s.to(slots[0].device) for s in slots])
            red = st.min(dim=0).values if op == "min" else st.max(dim=0).values if op == "max" else st.sum(dim=0)
            for s in slots:
                s.copy_(red)
            if red.is_cuda:
                torch.cuda.synchronize()
            self.g.bytes_moved += red.numel() * red.element_size()
        self._collective(t, root)
        return types.SimpleNamespace(wait=lambda: None) if async_op else None

    def all_gather_into_tensor(self, out, t):
        import torch

        def root(slots):
            cat = torch.cat([s[1].reshape(-1) for s in slots])
            for s in slots:
                s[0].reshape(-1).copy_(cat)
            if cat.is_cuda:
                torch.cuda.synchronize()
            self.g.bytes_moved += cat.numel() * cat.element_size()
        self._collective((out, t), root)

    def all_gather(self, outs, t):
        def root(slots):
            for dst, _ in slots:
                for r, (_, src) in enumerate(slots):
                    dst[r].copy_(src)
        self._collective((outs, t), root)

    def barrier(self):
        self._collective(None, lambda slots: None)


def run_ranks(world, fn, backend="nccl"):
    """fn(dist, rank) on `world` rank threads, one at a time between collectives; returns [fn's result per rank] (re-raises the first error)"""
    g = BatonGroup(world, backend)
    out, err = [None] * world, [None] * world

    def body(r):
        d = g.rank_handle(r)
        d.take()
        try:
            out[r] = fn(d, r)
        except BaseException as e:          # noqa: BLE001 -- reported by the caller; the other ranks are released through the barrier
            err[r] = e
            g.barrier.abort()
        finally:
            if d.holding:
                d.give()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    real = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if real or any(e is not None for e in err):
        raise (real or [e for e in err if e is not None])[0]
    return out, g
