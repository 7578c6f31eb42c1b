"""One-process-per-GPU sharding of the nearest-neighbour search (torch.distributed; backend "nccl" = RCCL on ROCm).

The packed sequence set is replicated on every GPU (31 MB for 50 k x 2.5 kb).  The 1-set search evaluates each
unordered pair once, on the rank that owns the pair's LOWER index; ownership is BLOCK-CYCLIC: the length-sorted order is cut into
blocks of SHARD_BLOCK = 256 consecutive entries (one tile row of the bound matrix) dealt round-robin, rank r of N owns the blocks
r, r + N, ... (`shard_of`).  Every rank gets the same mix of dense and sparse length regions, so the very uneven windows balance
by themselves (195 blocks at 50 000 reads), and a rank's 256 x 256 bound tiles are as dense as on one GPU -- with entry-cyclic
ownership (r, r + N, ...: rounds 2-3) a tile's 256 rows spanned N times the columns and the bound phase hardly scaled.
`shard_ranges` is the contiguous alternative, balanced by estimated window sizes.
Phase 0 = q-gram bounds of the rank's rows and the seeds they give; phase 1 = survivor lists and alignments of those rows;
phase 2 = 128/256/512-row bands and the un-banded kernel, only if the reduced bounds leave a query unresolved.  Exchange steps
(the only data-path collectives):
    all_reduce(MIN) of best[n]   after phase 0 (global seeds: without it the ranks' thresholds stay loose and the summed work
                                 doubles) and after phase 1 (and 2); the same tensor carries every rank's status word and
                                 edge count                                                      (4 B x (n + 1 + N))
    all_gather of the candidate edges that attain best[] on their rank, in fixed-size blocks of the largest count
    known from the last reduction (no size exchange)                                           (12 B x edges)
The reference has no distributed path (its Pool chunking: /root/reference/modules/nearest_neighbor_graph.py:33-35).
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .store import nn_finalize


def shard_ranges(lens, world_size, kcap=63, two_set_targets=None):
    """Contiguous ranges of the length-sorted order with ~equal estimated work.

    1-set: entry q is charged the number of longer-or-equal entries within kcap of its length (its upward window).
    2-set: every query is charged the number of targets within kcap of its length."""
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    if world_size <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world_size, 1) - 1)
    if two_set_targets is None:
        hi = np.searchsorted(lens, lens + kcap, side="right")
        work = (hi - np.arange(n) - 1).astype(np.float64) + 1.0
    else:
        t_lens = np.sort(lens[np.asarray(two_set_targets, dtype=bool)])
        work = (np.searchsorted(t_lens, lens + kcap, "right") - np.searchsorted(t_lens, lens - kcap, "left")).astype(np.float64) + 1.0
        work[np.asarray(two_set_targets, dtype=bool)] = 0.0
    cum = np.cumsum(work)
    cuts = [0]
    for r in range(1, world_size):
        cuts.append(int(np.searchsorted(cum, cum[-1] * r / world_size)))
    cuts.append(n)
    cuts = np.maximum.accumulate(np.asarray(cuts))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world_size)]


SHARD_BLOCK = 256          # = QM_TILE of csrc/qgram_mm.hpp: one tile row of the bound matrix


def shard_of(rank, world, n):
    """(q_begin, q_end, q_stride, q_block) of include/isocon_hip.h for rank `rank` of `world`: block-cyclic ownership.  Blocks of
    SHARD_BLOCK entries; on small sets the block shrinks (a power of two) until every rank has at least four blocks, so that the
    round-robin still balances -- a function of (n, world) only: the same on every rank."""
    block = SHARD_BLOCK
    while block > 1 and n < 4 * world * block:
        block //= 2
    return rank * block, n, world * block, block


PHASE2_STEPS = 1          # sub-steps of the wide-band phase of a sharded search (a reduction of best[] after each); see protocol_steps


def protocol_steps(rank, world, n, phase2_steps=None):
    """[(phase, (q_begin, q_end, q_stride, q_block))] of one sharded search.  Phases 0 and 1 cover the rank's shard.  Phase 2 (the
    128..512-row bands, entered only while a query longer than 63 is still unresolved) has no seeds: a query's threshold tightens only
    through the pairs a rank itself evaluates, and with N ranks each sees 1 / N of a query's partners until the next reduction -- summed
    work x1.24 / x1.27 at 4 / 8 ranks on the 200 000-read C5 set.  The phase CAN be run in sub-steps (the rank's blocks dealt into
    phase2_steps sub-shards, an all_reduce(MIN) of best[] after each, the phase's query set fixed at its start and handed down as
    `wide_queries`; any order of evaluating the pairs gives the same graph: tests/test_gpu_nn_graph.py).  Measured on 50 000 reads of the
    C5 shape with four sub-steps (profiles/r04l_phase2_substeps.txt): the summed work falls (x1.28 -> 1.11 at 4 ranks, x1.38 -> 1.25 at 8)
    but the critical path does not (1.77 -> 2.04 s at 2 ranks, 0.71 -> 0.83 s at 8): every sub-step pays its own sample stage and launch
    tails, queries that already have a bound run the narrower bands again, and a quarter of a rank's blocks balances worse (max / mean
    1.2 at 8 ranks).  So the default is ONE step."""
    qb, qe, qs, qk = shard_of(rank, world, n)
    steps = [(0, (qb, qe, qs, qk)), (1, (qb, qe, qs, qk))]
    sub = (PHASE2_STEPS if phase2_steps is None else phase2_steps) if world > 1 else 1
    for k in range(sub):
        steps.append((2, (qb + k * qs, qe, qs * sub, qk)))
    return steps


class _Agreement(object):
    """ONE all_reduce(MIN) per search of (fingerprint, -fingerprint): every rank learns from the reduced pair alone whether all ranks packed
    the very same sequences in the very same order (min == max).  Done on every call (a decision cached per store would let a rank whose
    store was rebuilt enter a collective that the others skip) -- but NOT waited for: the reduction is issued first, the rank's phase 0 (no
    collective in it, the rank's own store only) runs meanwhile, and confirm() is called before the first collective whose size
    depends on n.  A mismatch therefore still raises on every rank before any exchange of best[]; what it no longer costs is a
    launch + host round trip in front of the first kernel (0.36 ms of an 8.5 ms search at C3, profiles/r06l_sharded_overhead.txt).

    Whether a rank keeps best[] and its edges in device memory (`device_resident`) is the rank's OWN business: the two ways of
    running a phase issue the same collectives on tensors of the same shape, dtype and device (sharded_nn_graph below), so nothing is agreed."""

    def __init__(self, store, dist, device):
        import torch
        fp = getattr(store, "fingerprint", None)
        fp = 0 if fp is None else int(fp)
        self.t = torch.tensor([fp, -fp], dtype=torch.int64).to(device, non_blocking=True)
        self.work = dist.all_reduce(self.t, op=dist.ReduceOp.MIN, async_op=True)
        self.same = None

    def confirm(self):
        if self.same is None:
            if self.work is not None:
                self.work.wait()
            lo, neg_hi = (int(x) for x in self.t.tolist())
            self.same = lo == -neg_hi
        if not self.same:
            raise RuntimeError("sharded_nn_graph: the ranks hold different sequence sets / orders (fingerprint mismatch); "
                               "build the store from a deterministic order (not from set())")


_ONE_HIP_RUNTIME = None          # this process maps exactly one libamdhip64 (read from /proc/self/maps ONCE: 0.3 ms in a torch process)


def device_resident(store, n, device):
    """this rank can keep best[] and its candidate edges in device memory between the exchange steps"""
    global _ONE_HIP_RUNTIME
    if not (device.type == "cuda" and hasattr(store, "nn_partial_dev") and n > 0):
        return False
    if _ONE_HIP_RUNTIME is None:          # (asked after the library and torch's runtime are both loaded: the store exists, the device is torch's)
        import torch
        _ONE_HIP_RUNTIME = len(_lib.hip_runtimes_loaded()) == 1 and torch.cuda.is_available()
    return _ONE_HIP_RUNTIME


def _all_gather_rows(dist, rows, device):
    """all_gather of a [k, 3] int32 array whose k differs per rank (padded to the max)."""
    import torch
    world = dist.get_world_size()
    k = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    ks = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(ks, k)
    kmax = int(max(int(x.item()) for x in ks))
    buf = torch.zeros((max(kmax, 1), 3), dtype=torch.int32, device=device)
    if rows.shape[0]:
        buf[:rows.shape[0]] = torch.from_numpy(np.ascontiguousarray(rows)).to(device)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    parts = [outs[r][:int(ks[r].item())].cpu().numpy() for r in range(world)]
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, 3), np.int32)


def same_everywhere(value, dist=None, device=None):
    """True iff the int64 `value` (a store fingerprint, a hash of a pair list ...) is the same on every rank; every rank gets
    the same answer from ONE all_reduce(MIN) of (value, -value): the reduced pair is (min, -max)."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    value = int(value)
    t = torch.tensor([value, -value], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)          # (min, -max): equal on every rank iff min == max, which every rank sees
    lo, neg_hi = (int(x) for x in t.tolist())
    return lo == -neg_hi


def _digest(*arrays):
    """63-bit digest of some numpy arrays (order-sensitive)"""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return int.from_bytes(h.digest(), "little") >> 1


def _staging(store, name, n_int32, device):
    """A reusable int32 host tensor of at least n_int32 elements (pinned when the collectives run on the GPU), kept on the store."""
    import torch
    cache = getattr(store, "_dist_staging", None)
    if cache is None:
        cache = {}
        try:
            store._dist_staging = cache
        except AttributeError:
            pass
    t = cache.get(name)
    if t is None or t.numel() < n_int32:
        t = torch.empty(int(n_int32 * 1.25) + 64, dtype=torch.int32, pin_memory=device.type == "cuda")
        cache[name] = t
    return t[:n_int32]


def _nn_graph_device_resident(store, is_converged, is_target, depth, dist, lap, agreement):
    """The protocol of sharded_nn_graph with best[] and the candidate edges in device memory from the first phase to the CSR
    (SeqStore.nn_partial_dev / nn_hits_dev / nn_finalize_dev): RCCL reduces and gathers the device buffers themselves, the host sees
    1 + world status words per reduction and the final graph.  The library works on the null stream, which is torch's current
    stream here, so its kernels and the collectives are ordered without host synchronisation."""
    import time
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", torch.cuda.current_device())
    if torch.cuda.current_stream(dev) != torch.cuda.default_stream(dev):
        raise RuntimeError("sharded_nn_graph: call it on the default stream (the library's kernels run there)")
    n = store.n
    tl = time.perf_counter()
    cache = store.__dict__.setdefault("_dist_device", {})
    t = cache.get("reduce")
    if t is None or t.numel() != n + 1 + world:
        t = cache["reduce"] = torch.empty(n + 1 + world, dtype=torch.int32, device=dev)
        cache["tail_host"] = torch.empty(1 + world, dtype=torch.int32).pin_memory()
    tail_host = cache["tail_host"]
    t[:n].fill_(_lib.NN_INF)
    # phase 2 runs only if a query longer than 63 is still without a neighbour after the reduction (the same on every rank)
    far = np.asarray(store.lens)[:n] > 63
    if is_converged is not None:
        far &= np.asarray(is_converged)[:n] == 0
    if is_target is not None:
        far &= np.asarray(is_target)[:n] == 0
    far_d = torch.from_numpy(far).to(dev)
    tl = lap("setup", tl)
    stats_all = []
    held = 0
    wide = None          # the queries of phase 2: unresolved when the phase begins (the same on every rank: best[] is reduced)
    for phase, (qb, qe, qs, qk) in protocol_steps(rank, world, n):
        if phase == 2 and wide is None:
            wide_d = (t[:n] == _lib.NN_INF) & far_d
            if not bool(wide_d.any().item()):
                stats_all.append({k: 0 for k in stats_all[-1]} if stats_all else {})
                break
            wide = wide_d.to(torch.uint8).cpu().numpy()
        err = None
        tl = time.perf_counter()
        try:
            held, stats = store.nn_partial_dev(qb, qe, phase, t.data_ptr(), phase > 0, is_converged=is_converged, is_target=is_target,
                                               depth=depth, q_stride=qs, q_block=qk, wide_queries=wide if phase == 2 else None)
        except Exception as e:          # noqa: BLE001 -- re-raised below, on every rank
            err, stats = e, {}
        tl = lap("nn_partial_phase%d" % phase, tl)
        stats_all.append(stats)
        # best[n] | this rank's status | -(edges this rank holds) in its own slot: one all_reduce(MIN) of the device tensor
        tail_host.zero_()
        tail_host[0] = -1 if err is not None else 0
        tail_host[1 + rank] = -held
        t[n:].copy_(tail_host, non_blocking=True)
        agreement.confirm()                                # (the fingerprints: reduced while phase 0 ran; raises on every rank alike)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)           # THE exchange of thresholds
        tail_host.copy_(t[n:])
        if int(tail_host[0]) != 0:
            raise RuntimeError("sharded_nn_graph: phase %d failed on %s" % (phase, "this rank: %r" % (err,) if err is not None else "another rank"))
        counts = (-tail_host[1:]).numpy().copy()
        tl = lap("reduce_min", tl)
    tl = time.perf_counter()
    # ONE all_gather of fixed-size blocks (the largest count of the last reduction; unused rows are -1), device to device
    kmax = max(int(counts.max()) if len(counts) else 0, 1)
    blk = cache.get("block")
    if blk is None or blk.shape[0] < kmax:
        blk = cache["block"] = torch.empty((int(kmax * 1.25) + 64, 3), dtype=torch.int32, device=dev)
    blk = blk[:kmax]
    store.nn_hits_dev(t.data_ptr(), blk.data_ptr(), kmax)
    if world == 1:
        gathered = blk
    else:
        gathered = cache.get("gathered")
        if gathered is None or gathered.shape[0] < world * kmax:
            gathered = cache["gathered"] = torch.empty((int(world * kmax * 1.25) + 64, 3), dtype=torch.int32, device=dev)
        gathered = gathered[:world * kmax]
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(gathered, blk)
        else:
            dist.all_gather(list(gathered.view(world, kmax, 3).unbind(0)), blk)
    tl = lap("gather_edges", tl)
    out = store.nn_finalize_dev(t.data_ptr(), gathered.data_ptr(), gathered.shape[0])
    lap("finalize", tl)
    return out + (stats_all,)


def sharded_nn_graph(store, is_converged=None, is_target=None, depth=2 ** 32, dist=None, device=None, return_stats=False, laps=None):
    """Exact NN graph of `store` (length-sorted) computed by all ranks of the default process group.

    `store` needs .n, .lens and .nn_partial(q_begin, q_end, phase, best, is_converged=, is_target=, depth=, q_stride=, q_block=).
    Every rank returns the full (best, row_ptr, cols).  laps: optional dict that receives the seconds spent per protocol part."""
    import time
    import torch

    def lap(name, t0):
        if laps is not None:
            laps[name] = laps.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = store.n
    tl = time.perf_counter()
    agreement = _Agreement(store, dist, device)
    tl = lap("fingerprint", tl)
    if device_resident(store, n, device):
        out = _nn_graph_device_resident(store, is_converged, is_target, depth, dist, lap, agreement)
        return out if return_stats else out[:3]
    hits_all, stats_all = [], []
    best = np.full(max(n, 1), _lib.NN_INF, dtype=np.int32)
    nb = len(best)
    lens_np = np.asarray(store.lens)[:n]
    is_query = np.ones(n, dtype=bool)
    if is_converged is not None:
        is_query &= np.asarray(is_converged)[:n] == 0
    if is_target is not None:
        is_query &= np.asarray(is_target)[:n] == 0
    # Phase 0 = q-gram bounds of the rank's rows + the seeds they give; phase 1 = survivor lists + alignments of those rows (finds the
    # bound matrix of phase 0 still in place); phase 2 = wide bands, only when the reduced bounds leave a query unresolved.  The
    # reduction between phases 0 and 1 is what keeps the work constant: a rank's seeds only tighten the entries its own rows reach,
    # and without the exchange every rank aligns against looser thresholds -- measured on C3 with the two phases fused into one call
    # (isocon_nn_partial phase 3): summed DP work x 1.98 / 2.68 / 3.94 at 2 / 4 / 8 ranks (profiles/r03d_emulate_sharding_fused.log).
    wide = None          # the queries of phase 2: unresolved when the phase begins
    for phase, (qb, qe, qs, qk) in protocol_steps(rank, world, n):          # block-cyclic ownership; phase 2 in sub-steps
        if phase == 2 and wide is None:
            # best[] is identical on all ranks after the reduction, so all ranks skip (or run) this phase together
            wide = (is_query & (best[:n] == _lib.NN_INF) & (lens_np > 63)).astype(np.uint8)
            if not wide.any():
                stats_all.append({k: 0 for k in stats_all[-1]} if stats_all else {})
                break
        # A rank that fails here (out of memory, a HIP error) must not leave the others blocked in the collective: its
        # status travels as one more word of the very reduction that follows (MIN: -1 wins), then every rank raises.
        err = None
        tl = time.perf_counter()
        try:
            hits, stats = store.nn_partial(qb, qe, phase, best, is_converged=is_converged, is_target=is_target, depth=depth,
                                           q_stride=qs, q_block=qk, **({"wide_queries": wide} if phase == 2 else {}))
        except Exception as e:          # noqa: BLE001 -- re-raised below, on every rank
            err, hits, stats = e, np.zeros((0, 3), np.int32), {}
        tl = lap("nn_partial_phase%d" % phase, tl)
        hits_all.append(hits)
        stats_all.append(stats)
        # one reduction carries: best[n] | this rank's status | -(edges this rank holds so far) in its own slot (0 in the
        # others' slots): after MIN every rank knows every rank's edge count, so the gather below needs no size exchange.
        # The tensor is assembled in pinned memory and reduced on the device RCCL runs on.
        t = _staging(store, "reduce", nb + 1 + world, device)          # pinned, reused from call to call
        t[:nb] = torch.from_numpy(best)
        t[nb:] = 0
        t[nb] = -1 if err is not None else 0
        t[nb + 1 + rank] = -sum(len(h) for h in hits_all)
        td = t.to(device, non_blocking=True) if device.type == "cuda" else t
        agreement.confirm()                                 # (the fingerprints: reduced while phase 0 ran; raises on every rank alike)
        dist.all_reduce(td, op=dist.ReduceOp.MIN)           # THE exchange of thresholds
        if device.type == "cuda":
            t.copy_(td, non_blocking=False)
        if int(t[nb]) != 0:
            raise RuntimeError("sharded_nn_graph: phase %d failed on %s" % (phase, "this rank: %r" % (err,) if err is not None else "another rank"))
        best[:] = t[:nb].numpy()
        counts = (-t[nb + 1:]).numpy().copy()
        tl = lap("reduce_min", tl)
    hits = np.concatenate(hits_all, axis=0) if hits_all else np.zeros((0, 3), np.int32)
    if len(hits):   # only edges that attain the global minimum of their endpoint travel
        keep = (hits[:, 2] >= 0) & (hits[:, 2] == best[np.clip(hits[:, 0], 0, max(n - 1, 0))])
        hits = hits[keep]
    tl = time.perf_counter()
    # exchange step 3: ONE all_gather of fixed-size blocks (the largest count of the reduction above; unused rows are -1)
    kmax = max(int(counts.max()) if len(counts) else 0, 1)
    buf = _staging(store, "edges", kmax * 3, device).view(kmax, 3)
    buf[len(hits):] = -1
    if len(hits):
        buf[:len(hits)] = torch.from_numpy(np.ascontiguousarray(hits, dtype=np.int32))
    if device.type == "cuda":
        bd = buf.to(device, non_blocking=True)
        out_d = torch.empty((world * kmax, 3), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(out_d, bd)
        out_h = _staging(store, "gathered", world * kmax * 3, device).view(world * kmax, 3)
        out_h.copy_(out_d, non_blocking=False)
        gathered = out_h.numpy()
    else:
        outs = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(outs, buf)
        gathered = torch.cat(outs, dim=0).numpy()
    gathered = gathered[gathered[:, 2] >= 0]
    tl = lap("gather_edges", tl)
    out = nn_finalize(n, best[:n], gathered)
    tl = lap("finalize", tl)
    return out + (stats_all,) if return_stats else out


# ---- explicit pair lists (edlib_align_sequences*, sw_align_sequences*): embarrassingly parallel ------------------------
def _pair_shards(lens_a, lens_b, world):
    """Round-robin over the pairs sorted by len(a) * len(b) (largest first): every rank gets the same mix of sizes."""
    cost = np.asarray(lens_a, dtype=np.int64) * np.asarray(lens_b, dtype=np.int64)
    order = np.argsort(-cost, kind="stable")
    return [order[r::world] for r in range(world)]


def _all_gather_ragged(dist, arr, device):
    """all_gather of 1-D int32 arrays of different lengths; returns the per-rank arrays."""
    import torch
    world = dist.get_world_size()
    k = torch.tensor([arr.shape[0]], dtype=torch.int64, device=device)
    ks = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(ks, k)
    kmax = max(int(x.item()) for x in ks)
    buf = torch.zeros(max(kmax, 1), dtype=torch.int32, device=device)
    if arr.shape[0]:
        buf[:arr.shape[0]] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(device)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [outs[r][:int(ks[r].item())].cpu().numpy() for r in range(world)]


def sharded_ed_pairs(store, a, b, k=None, dist=None, device=None):
    """Edit distances of the pairs (a[i], b[i]) computed by all ranks (each a round-robin share), gathered everywhere.
    `store` needs .lens and .ed_pairs(a, b, k).  The reference's counterpart is the Pool of EAM:25-47."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    a = np.asarray(a, dtype=np.uint32); b = np.asarray(b, dtype=np.uint32)
    lens = np.asarray(store.lens)
    shards = _pair_shards(lens[a], lens[b], world)
    mine = shards[rank]
    kk = None if k is None else np.asarray(k, dtype=np.int32)[mine]
    ed = np.asarray(store.ed_pairs(a[mine], b[mine], kk), dtype=np.int32) if len(mine) else np.zeros(0, np.int32)
    parts = _all_gather_ragged(dist, ed, device)
    out = np.empty(len(a), dtype=np.int32)
    for r in range(world):
        out[shards[r]] = parts[r]
    return out


def sharded_hw_pairs(store, q, t, k, dist=None, device=None):
    """Infix alignments (SeqStore.hw_pairs: distance, start, end, leading / trailing insertion run) of the pairs (q[i] inside
    t[i]) computed by all ranks, gathered everywhere -- the candidate-vs-candidate graph of the statistical test
    (end_invariant_functions.get_all_NN) is an explicit pair list like the others; the reference's counterpart is the Pool
    of end_invariant_functions.py:708-741."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    q = np.asarray(q, dtype=np.uint32); t = np.asarray(t, dtype=np.uint32)
    kk = np.ascontiguousarray(np.broadcast_to(np.asarray(k, dtype=np.int32), q.shape))
    lens = np.asarray(store.lens)
    if not same_everywhere(_digest(q, t, kk, np.asarray([getattr(store, "fingerprint", 0)], dtype=np.int64)), dist, device):
        raise RuntimeError("sharded_hw_pairs: the ranks hold different pair lists / sequence sets")
    shards = _pair_shards(lens[q], lens[t], world)
    mine = shards[rank]
    res = np.asarray(store.hw_pairs(q[mine], t[mine], kk[mine]), dtype=np.int32).reshape(-1) if len(mine) else np.zeros(0, np.int32)
    parts = _all_gather_ragged(dist, res, device)
    out = np.empty((len(q), 5), dtype=np.int32)
    for r in range(world):
        out[shards[r]] = parts[r].reshape(-1, 5)
    return out


def sharded_sg_trace(store, a, b, mismatch, match=2, open_=2, ext=0, tie_policy=0, ed_upper=None, dist=None, device=None):
    """Semi-global alignments of the pairs computed by all ranks, gathered everywhere: returns (ops, ops_ptr, res) in the
    caller's pair order like SeqStore.sg_trace.  CIGAR ops travel as one ragged int32 all_gather (a few hundred bytes per
    pair); the reference's counterpart is the Pool of SWM:121-162."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    world, rank = dist.get_world_size(), dist.get_rank()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    a = np.asarray(a, dtype=np.uint32); b = np.asarray(b, dtype=np.uint32)
    n = len(a)
    mm = np.broadcast_to(np.asarray(mismatch, dtype=np.int8), (n,))
    lens = np.asarray(store.lens)
    shards = _pair_shards(lens[a], lens[b], world)
    mine = shards[rank]
    if len(mine):
        hint = None if ed_upper is None else np.asarray(ed_upper, dtype=np.int32)[mine]
        ops, ptr, res = store.sg_trace(a[mine], b[mine], mm[mine], match=match, open_=open_, ext=ext, tie_policy=tie_policy, ed_upper=hint)
    else:
        ops, ptr, res = np.zeros(0, np.uint32), np.zeros(1, np.int64), np.zeros((0, 6), np.int32)
    cnt = np.diff(np.asarray(ptr, dtype=np.int64)).astype(np.int32)
    g_ops = _all_gather_ragged(dist, ops.view(np.int32) if ops.dtype == np.uint32 else ops.astype(np.int32), device)
    g_cnt = _all_gather_ragged(dist, cnt, device)
    g_res = _all_gather_ragged(dist, np.ascontiguousarray(res, dtype=np.int32).reshape(-1), device)
    counts = np.zeros(n, dtype=np.int64)
    for r in range(world):
        counts[shards[r]] = g_cnt[r]
    out_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=out_ptr[1:])
    out_ops = np.empty(int(out_ptr[n]), dtype=np.uint32)
    out_res = np.empty((n, 6), dtype=np.int32)
    for r in range(world):
        src_ptr = np.zeros(len(shards[r]) + 1, dtype=np.int64)
        np.cumsum(g_cnt[r], out=src_ptr[1:])
        rr = g_res[r].reshape(-1, 6)
        o = g_ops[r].view(np.uint32)
        for i, p in enumerate(shards[r].tolist()):
            out_ops[out_ptr[p]:out_ptr[p + 1]] = o[src_ptr[i]:src_ptr[i + 1]]
            out_res[p] = rr[i]
    return out_ops, out_ptr, out_res
