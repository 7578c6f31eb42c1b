#!/usr/bin/env python3
"""bench.py -- IsoCon hot path on MI355X: read x candidate alignments / s and NN-graph build wall-time.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1 under torch.distributed.run, one rank per GPU).
One "step" = one exact nearest-neighbour-graph build (compute_nearest_neighbor_graph semantics,
/root/reference/modules/nearest_neighbor_graph.py:237-296) over the workload, inputs already packed and resident in
HBM.  Workload = BASELINE.json configs[2]: 50 k synthetic CCS reads, ~2.5 kb, 10 isoforms (seed 30001).

value = alignments / s, where the numerator is the reference-defined pair set that ANY run of the reference loop must
evaluate: for every query the entries whose length is within its final NN distance (|len difference| <= best_ed is
the loop's own window rule, NNG:145,152).  It is a lower bound of the reference's edlib call count (whose exact
value depends on nr_cores through the chunk-local seed dictionary, NNG:112,125-129), is identical for the CPU
baseline and the GPU, and is computable from the result alone.

Extra objects on the JSON line: `roofline` (dominant kernel k_nn_scan_refill, HBM bound on ALGORITHMIC bytes,
SURVEY.md 8(d): len(q)+len(t)+8 bytes per aligned pair) and `cpu_baseline` (the C oracle -- a restatement of the
edlib-based loop -- under a multiprocessing Pool on this host's cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
VALU_CYCLES_PER_COLUMN = 65.1                                        # measured issue costs x instruction mix of the main-pass column
VALU_PEAK_LANE_COLS = 256 * 4 * 64 * 2.4e9 / VALU_CYCLES_PER_COLUMN    # 256 CUs x 4 SIMDs x 64 lanes at 2.4 GHz


def window_pairs(lens, best):
    """sum over queries of |{t != q : |len_t - len_q| <= best_q}| (rows with no neighbour contribute 0)."""
    lens = np.asarray(lens, dtype=np.int64)
    b = np.asarray(best, dtype=np.int64)
    has = b >= 0
    lo = np.searchsorted(lens, lens - np.where(has, b, 0), "left")
    hi = np.searchsorted(lens, lens + np.where(has, b, 0), "right")
    return int(((hi - lo - 1) * has).sum())


# ---- CPU baseline (oracle under a Pool; test infrastructure used as the reported baseline only) -------------------
_G = {}


def _cpu_query(i):
    from oracle import oracle as O
    row_ptr, cols, eds, calls = O.nn_1set(_G["seqs"], _G["conv"], int(i), 1, packed=_G["packed"])
    best = int(eds[0]) if len(eds) else -1
    return int(i), best, int(calls)


def usable_cores():
    """Threads this process may actually run concurrently: min(cpu_count, affinity, cgroup CPU quota)."""
    c = os.cpu_count() or 1
    try:
        c = min(c, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                c = min(c, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            c = min(c, max(1, q // p))
    except Exception:
        pass
    return c


def usable_cores():
    """Threads this process may actually run concurrently: min(cpu_count, affinity, cgroup CPU quota)."""
    c = os.cpu_count() or 1
    try:
        c = min(c, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                c = min(c, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            c = min(c, max(1, q // p))
    except Exception:
        pass
    return c


def cpu_baseline(seqs, lens, budget_s=15.0, max_queries=4096):
    """Times the oracle's restatement of get_nearest_neighbors (NNG:110-198, edlib-style banded Myers) on a bounded
    sample of queries spread over the sorted order, Pool(processes=usable cores) as the reference does with nr_cores (NNG:30)."""
    from multiprocessing import Pool
    from oracle import oracle as O
    O.build()
    cores = usable_cores()
    n = len(seqs)
    _G["seqs"] = seqs
    _G["conv"] = np.zeros(n, dtype=np.uint8)
    _G["packed"] = O.pack(seqs)            # one ASCII buffer + offsets, inherited by the workers
    order = np.random.Generator(np.random.PCG64(7)).permutation(n)[:max_queries]
    done, t0 = [], time.perf_counter()
    with Pool(processes=cores) as pool:    # fork: the sequence list is inherited, not pickled per task
        pos = 0
        t0 = time.perf_counter()
        while pos < len(order):
            chunk = order[pos:pos + 2 * cores]
            done.extend(pool.map(_cpu_query, chunk.tolist(), chunksize=1))
            pos += len(chunk)
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    idx = np.array([d[0] for d in done])
    best = np.array([d[1] for d in done])
    calls = int(sum(d[2] for d in done))
    has = best >= 0
    lo = np.searchsorted(lens, lens[idx] - np.where(has, best, 0), "left")
    hi = np.searchsorted(lens, lens[idx] + np.where(has, best, 0), "right")
    pairs = int(((hi - lo - 1) * has).sum())
    return {"value": pairs / dt, "unit": "alignments/s", "cores": cores, "kind": "port",
            "sample": "%d of %d queries (random, seed 7) against the full %d-sequence set, %.1f s wall, %d edlib-style "
                      "calls (%.0f calls/s); oracle/isocon_oracle.c orc_nn_1set under multiprocessing.Pool(%d)"
                      % (len(done), n, n, dt, calls, calls / dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50000)
    ap.add_argument("--length", type=int, default=2500)
    ap.add_argument("--isoforms", type=int, default=10)
    ap.add_argument("--seed", type=int, default=30001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("ISOCON_GPU_DEVICE", str(local_rank))

    from isocon_amd import synth

    accs, seqs, true_isoforms = synth.make_reads(args.reads, args.length, args.isoforms, args.seed)
    # unique strings in first-appearance order, then a STABLE sort by length (NNG:243-246).  Not set(): its iteration
    # order depends on the per-process string hash seed, and every rank must pack the very same order.
    seqs = sorted(dict.fromkeys(seqs), key=len)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))

    # CPU baseline first: its fork()ed workers must exist (and be gone) before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(seqs, lens, budget_s=args.cpu_budget)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("ISOCON_DIST_BACKEND", "nccl")   # "gloo": functional test of the N>1 path on one GPU
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    from isocon_amd.dist import sharded_nn_graph
    from isocon_amd.store import SeqStore

    store = SeqStore(seqs)                     # packed + uploaded: inputs resident in HBM before timing

    on_gpu_group = dist is not None and dist.get_backend() == "nccl"
    red_device = torch.device("cuda", torch.cuda.current_device()) if (dist is None or on_gpu_group) and torch.cuda.is_available() else torch.device("cpu")

    def sync():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    last = {}

    def step():
        if world == 1:
            best, row_ptr, cols, stats = store.nn_graph()
            last.update(best=best, edges=len(cols), stats=[stats], row_ptr=row_ptr, cols=cols)
        else:
            best, row_ptr, cols, stats = sharded_nn_graph(store, dist=dist, return_stats=True)
            last.update(best=best, edges=len(cols), stats=stats, row_ptr=row_ptr, cols=cols)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    scan_ms = []
    for _ in range(args.steps):
        step()
        scan_ms.append(sum(x["scan_kernel_ms"] + x["seed_kernel_ms"] for x in last["stats"]))
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    n_align = window_pairs(lens, last["best"])
    value = n_align / (ms_per_step / 1e3)

    # roofline of the dominant kernel on this rank (live HIP-event time of its launches inside the timed region)
    st0 = {k: sum(x[k] for x in last["stats"]) for k in ("pairs_evaluated", "cells_columns")}   # this rank, all phases
    pairs_eval = int(st0["pairs_evaluated"])
    mean_len = float(lens.mean())
    alg_bytes = pairs_eval * (2.0 * mean_len + 8.0)
    k_ms = float(np.mean(scan_ms)) if scan_ms else 0.0
    achieved = alg_bytes / (k_ms / 1e3) / 1e9 if k_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("nn_scan_main_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    is_default = (args.reads, args.length, args.isoforms, args.seed) == (50000, 2500, 10, 30001)
    if not is_default:
        traffic = None          # the committed PMC figure belongs to the default workload only
    lane_cols_s = float(st0["cells_columns"]) / (k_ms / 1e3) if k_ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": "k_nn_scan_refill (main pass) + k_nn_scan_up (seed pass) of one step", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel_ms": k_ms, "pairs_per_launch": pairs_eval, "alg_bytes_per_pair": 2.0 * mean_len + 8.0,
                "lane_columns_per_s": lane_cols_s,
                # The byte model above charges every pair both sequences (what the reference hands to edlib); the kernel
                # reads a query once per ~6 500 pairs (LDS table) and the neighbours from L2/MALL, so frac may exceed 1:
                # HBM is not what bounds it.  The real ceiling is VALU issue: 20.5 instructions =
                # 65.1 issue cycles per 64-lane DP column.
                "issue_bound": {"unit": "lane-columns/s", "achieved": lane_cols_s, "peak": VALU_PEAK_LANE_COLS,
                                "frac": lane_cols_s / VALU_PEAK_LANE_COLS, "cycles_per_wave_column_by_mix": VALU_CYCLES_PER_COLUMN}}

    result = {
        "metric": "read x candidate alignments/sec (NN-graph build, %dk x %.1fkb reads)" % (args.reads // 1000, args.length / 1000.0),
        "value": value, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": ("C3: " if is_default else "custom: ") + "%d synthetic CCS reads (%d unique), ~%d bp, %d isoforms, seed %d; 1-set NN graph"
                               % (args.reads, len(seqs), args.length, args.isoforms, args.seed),
                   "alignments_per_step": n_align, "nn_graph_wall_ms": ms_per_step, "edges": int(last["edges"]),
                   "median_nn_distance": float(np.median(last["best"][last["best"] >= 0])) if (last["best"] >= 0).any() else None,
                   "parallelism": "1 process/GPU; pairs sharded by lower index; all_reduce(MIN)+all_gather over RCCL" if world > 1 else "single GPU"},
        "roofline": roofline,
    }
    # not part of the timed region: the other two kernels of the path on (read, first NN) pairs of the same set
    try:
        if world == 1:
            row_ptr = last["row_ptr"]
            has = np.nonzero(row_ptr[1:] > row_ptr[:-1])[0]
            q = has[:: max(1, len(has) // 4096)][:4096]
            t = last["cols"][row_ptr[q]]
            t0 = time.perf_counter(); ed, ed_ms = store.ed_pairs(t, q, None, return_ms=True); ed_wall = time.perf_counter() - t0
            mm = np.full(len(q), -2, dtype=np.int8)
            store.sg_trace(t[:64], q[:64], mm[:64])
            t0 = time.perf_counter(); ops, ptr, res, sw_ms = store.sg_trace(t, q, mm, return_ms=True); sw_wall = time.perf_counter() - t0
            cells = float((lens[t] * lens[q]).sum())
            # the same batch with the pairs' edit distances as band hints (what sw_align_sequences passes down)
            store.sg_trace(t[:64], q[:64], mm[:64], ed_upper=ed[:64])
            t0 = time.perf_counter(); ops_b, ptr_b, res_b, swb_ms = store.sg_trace(t, q, mm, return_ms=True, ed_upper=ed); swb_wall = time.perf_counter() - t0
            same = bool((res_b == res).all() and len(ops_b) == len(ops) and (ops_b == ops).all())
            # infix alignments (edlib HW + path, the candidate graph of the statistical test) of the same pairs, k = 25 and 63
            store.hw_pairs(t[:64], q[:64], 25)
            hw25, hw25_ms = store.hw_pairs(t, q, 25, return_ms=True)
            hw63, hw63_ms = store.hw_pairs(t, q, 63, return_ms=True)
            # the read -> candidate (2-set) search of the pipeline's last steps: all reads against the true isoforms
            cands = [c for c in dict.fromkeys(true_isoforms) if c not in set(seqs)]
            merged = sorted([(s, 0) for s in seqs] + [(c, 1) for c in cands], key=lambda x: len(x[0]))
            st2 = SeqStore([s for s, _ in merged])
            is_t = np.array([f for _, f in merged], dtype=np.uint8)
            st2.nn_graph(is_target=is_t)
            t0 = time.perf_counter(); b2, rp2, c2, stats2 = st2.nn_graph(is_target=is_t); two_wall = time.perf_counter() - t0
            st2.close()
            result["other_kernels"] = {
                "nn_2set_reads": int((is_t == 0).sum()), "nn_2set_candidates": int(is_t.sum()), "nn_2set_wall_ms": two_wall * 1e3,
                "nn_2set_kernel_ms": float(stats2["kernel_ms"]), "nn_2set_pairs_evaluated": int(stats2["pairs_evaluated"]),
                "nn_2set_reads_with_a_candidate": int((np.diff(rp2)[is_t == 0] > 0).sum()),
                "hw_pairs_per_s_kernel_k25": len(q) / (hw25_ms / 1e3) if hw25_ms > 0 else None, "hw_hits_k25": int((hw25[:, 0] >= 0).sum()),
                "hw_pairs_per_s_kernel_k63": len(q) / (hw63_ms / 1e3) if hw63_ms > 0 else None, "hw_hits_k63": int((hw63[:, 0] >= 0).sum()),
                "hw_distance_le_global_distance": bool(((hw63[:, 0] <= ed) | (ed > 63))[hw63[:, 0] >= 0].all()),
                "pairs": int(len(q)),
                "ed_pairs_per_s_kernel": len(q) / (ed_ms / 1e3) if ed_ms > 0 else None, "ed_pairs_wall_ms": ed_wall * 1e3,
                "sw_pairs_per_s_kernel": len(q) / (sw_ms / 1e3) if sw_ms > 0 else None, "sw_wall_ms": sw_wall * 1e3,
                "sw_cell_updates_per_s": cells / (sw_ms / 1e3) if sw_ms > 0 else None,
                "sw_trace_hbm_write_GBps": (cells / 2) / (sw_ms / 1e3) / 1e9 if sw_ms > 0 else None,
                "sw_banded_pairs_per_s_kernel": len(q) / (swb_ms / 1e3) if swb_ms > 0 else None, "sw_banded_wall_ms": swb_wall * 1e3,
                "sw_banded_equals_full": same}
    except Exception as e:  # the headline line must still be printed
        result["other_kernels"] = {"error": repr(e)}
    if cpu is not None:
        result["cpu_baseline"] = cpu
        result["speedup_vs_cpu_baseline"] = value / result["cpu_baseline"]["value"] if result["cpu_baseline"]["value"] else None
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
