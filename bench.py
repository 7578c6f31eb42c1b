#!/usr/bin/env python3
"""bench.py -- IsoCon hot path on MI355X: read x candidate alignments / s and NN-graph build wall-time.

Contract: `python bench.py --gpus N --steps K --warmup W`.  N > 1: one rank per GPU over RCCL -- either launched by the
caller under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) or, when WORLD_SIZE is unset,
started by this script itself as a child `python -m torch.distributed.run --nproc-per-node N bench.py ...` (before
this process has imported torch or touched the GPU) whose rank-0 JSON line is relayed.
One "step" = one exact nearest-neighbour-graph build (compute_nearest_neighbor_graph semantics,
/root/reference/modules/nearest_neighbor_graph.py:237-296) over the workload, inputs already packed and resident in
HBM.  Workload = BASELINE.json configs[2]: 50 k synthetic CCS reads, ~2.5 kb, 10 isoforms (seed 30001).

value = alignments / s, where the numerator is the reference-defined pair set that ANY run of the reference loop must
evaluate: for every query the entries whose length is within its final NN distance (|len difference| <= best_ed is
the loop's own window rule, NNG:145,152).  It is a lower bound of the reference's edlib call count (whose exact
value depends on nr_cores through the chunk-local seed dictionary, NNG:112,125-129), is identical for the CPU
baseline and the GPU, and is computable from the result alone.

Extra objects on the JSON line:
  roofline      dominant kernel k_qgram_mm (q-gram lower bounds of every pair of the length window as a banded A B^T on the matrix cores,
                fp4 MFMA): `achieved` = algorithmic flops (tiles x 65 536 pairs x K x 2) / the kernel's own HIP-event time of THIS run, against
                the dense fp4 peak; `traffic` = PMC bytes (2 x FETCH_SIZE + WRITE_SIZE) of the dispatch from profiles/counters.json, which
                carries a digest of the kernel sources it was collected from (other sources: ignored, null).  Until round 6 the dominant
                kernel was the table kernel k_nn_scan_refill (VALU-bound, 8.5 of 18 ms); the block filter in front of it (`filter_pass`:
                k_nn_block_filter, LDS-bound) now rejects 93 % of its pairs and what is left runs one pair per lane (`lanes_pass`).
                `step_kernels_ms` = HIP events per phase; `table_pass` only when table launches ran.  `algorithmic` = SURVEY 8(d)'s byte
                model, for reference only (it charges bytes the kernels never move through HBM).
  cpu_baseline  the reference's loop on this host's cores: real edlib if the wheel imports ("edlib"), else the C
                restatement under a Pool ("port").  Reported, not the target.  `ed_pairs` / `sw_pairs`: the pair-list half of the
                metric -- the reference's Pool pattern of edlib_align_sequences / sw_align_sequences on a sample of the partition
                pairs the wrappers below were timed on.
  wrappers      wall time of the PUBLIC functions end to end (string handling, H2D, kernels, D2H, dict rebuild).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                       # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2  # 256 CUs x 4 SIMD-32 x 2.4 GHz, one wave64 VALU instruction per 2 cycles (same guide)
HALF_RATE_SHARE = 5.0 / 19.0      # of the table kernel's column (static, from its ISA)
HALF_RATE_COST = 1.72             # issue cost of a half-rate vector instruction relative to a full-rate one (measured)
STREAM_RATE_OF_NOMINAL = 2.0 / 2.43   # a pure stream of independent v_and_b32, 6 waves per SIMD, every CU busy: 2.43 cycles per instruction
                                      # at the nominal 2.4 GHz (clock under load + issue overhead): what "all issue slots used" measures as
MFMA_FP4_PEAK_MACS = 5.0e15                 # ~10 PFLOP/s dense FP4 (same guide, chip-level parameters) = 5e15 multiply-adds / s
# digest of the C3 graph every row of which was recomputed with the reference loop on the CPU (tests/golden/g17_c3_graph.npz, written by
# tests/golden/make_golden_g17.py; tests/test_oracle_golden.py checks that this constant is that fixture's digest): a step that
# produces another graph is not a measurement
EXPECTED_GRAPH_DIGEST_C3 = "65944c838d76a6c9"
# digest of the alignments of the C3 partition pair list (49 990 pairs) as the oracle computes them on the CPU (tests/golden/g18_c3_sw.npz, written by
# tests/golden/make_golden_g18.py: orc_sg_trace = SWM:64-86, tie policy 0; tests/test_oracle_golden.py checks that this constant is that fixture's)
EXPECTED_SW_DIGEST_C3 = "79dafdad72773d7f"
COUNTERS = os.path.join(ROOT, "profiles", "counters.json")   # written by scripts/summarize_profile.py from rocprofv3 --pmc passes


def window_pairs(lens, best):
    """sum over queries of |{t != q : |len_t - len_q| <= best_q}| (rows with no neighbour contribute 0)."""
    lens = np.asarray(lens, dtype=np.int64)
    b = np.asarray(best, dtype=np.int64)
    has = b >= 0
    lo = np.searchsorted(lens, lens - np.where(has, b, 0), "left")
    hi = np.searchsorted(lens, lens + np.where(has, b, 0), "right")
    return int(((hi - lo - 1) * has).sum())


def graph_digest(best, row_ptr, cols):
    """identity of a graph (bounds, row pointers, neighbour order): equal across N = 1, 2, 4, 8 runs of the same workload"""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for x, t in ((best, np.int32), (row_ptr, np.int64), (cols, np.uint32)):
        h.update(np.ascontiguousarray(x, dtype=t).tobytes())
    return h.hexdigest()


_H1, _H2, _H3 = np.uint64(0xD6E8FEB86659FD93), np.uint64(0x9E3779B97F4A7C15), np.uint64(0xC2B2AE3D27D4EB4F)


def sw_pair_hashes(ops, ops_ptr):
    """uint64 per pair: position-dependent hash of the pair's run-length CIGAR ops (len << 4 | code), vectorised so that 50 000 pairs cost
    milliseconds (what tests/golden/g18_*_sw.npz stores instead of ~15 MB of ops)"""
    ptr = np.asarray(ops_ptr, dtype=np.int64)
    x = np.asarray(ops, dtype=np.uint64)[:int(ptr[-1])]
    pos = (np.arange(len(x), dtype=np.int64) - np.repeat(ptr[:-1], np.diff(ptr))).astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x + np.uint64(1) + pos * _H1) * _H2
        x ^= x >> np.uint64(29)
        x *= _H3
        x ^= x >> np.uint64(32)
        cs = np.concatenate([np.zeros(1, np.uint64), np.cumsum(x, dtype=np.uint64)])
        return cs[ptr[1:]] - cs[ptr[:-1]]


def sw_digest(a, b, res, hashes):
    """identity of a batch of alignments: the pairs in (a, b) order with their result rows (score, end cell, matches, mismatches, indels)
    and op hashes"""
    import hashlib
    a, b = np.asarray(a, dtype=np.int64), np.asarray(b, dtype=np.int64)
    o = np.lexsort((b, a))
    h = hashlib.blake2b(digest_size=8)
    for x, t in ((a[o], np.uint32), (b[o], np.uint32), (np.asarray(res)[o], np.int32), (np.asarray(hashes)[o], np.uint64)):
        h.update(np.ascontiguousarray(x, dtype=t).tobytes())
    return h.hexdigest()


def usable_cores():
    """Threads this process may actually run concurrently: min(cpu_count, affinity, cgroup CPU quota)."""
    c = os.cpu_count() or 1
    try:
        c = min(c, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
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


# ---- CPU baseline: the reference's loop NNG:110-198 on this host's cores (bounded sample) ---------------------------
_G = {}
CHUNK = 20            # queries per task: the reference's minimum chunk (NNG:33 max(int(n / (10 nr_cores)), 20)); the
#                       chunk-local seed dictionary NNG:112,125-129 fires inside a chunk as it does there


def _cpu_chunk_port(start):
    from oracle import oracle as O
    cnt = min(CHUNK, len(_G["seqs"]) - start)
    row_ptr, cols, eds, calls = O.nn_1set(_G["seqs"], _G["conv"], int(start), cnt, packed=_G["packed"])
    best = [int(eds[row_ptr[i]]) if row_ptr[i + 1] > row_ptr[i] else -1 for i in range(cnt)]
    return int(start), best, int(calls)


def _cpu_chunk_edlib(start):
    """The reference loop (NNG:110-198) restated in Python around the real edlib.align -- Backend A of SURVEY 8(d)."""
    import edlib
    seqs = _G["seqs"]
    n = len(seqs)
    cnt = min(CHUNK, n - start)
    calls = 0
    out = []
    lower = {}
    for i in range(start, start + cnt):
        s1 = seqs[i]
        best_ed = lower[i] if i in lower else len(s1)
        stop_up = stop_down = False
        j = 1
        found = False
        while True:
            if i - j < 0:
                stop_down = True
            if i + j >= n:
                stop_up = True
            if not stop_down and len(s1) - len(seqs[i - j]) > best_ed:
                stop_down = True
            if not stop_up and len(seqs[i + j]) - len(s1) > best_ed:
                stop_up = True
            for stopped, p in ((stop_down, i - j), (stop_up, i + j)):
                if stopped:
                    continue
                calls += 1
                d = edlib.align(s1, seqs[p], mode="NW", task="distance", k=best_ed)["editDistance"]
                if 0 < d < best_ed:
                    best_ed = d
                    found = True
                elif d == best_ed:
                    found = True
                if d > 0 and (p not in lower or d < lower[p]):       # NNG:164-169,180-185
                    lower[p] = d
            if stop_down and stop_up:
                break
            j += 1
        out.append(best_ed if found else -1)
    return int(start), out, calls


def cpu_baseline(seqs, lens, budget_s=15.0, keep_pool=False):
    """Times get_nearest_neighbors (NNG:110-198) on random chunks of CHUNK consecutive queries against the full set,
    Pool(processes=usable cores) as the reference does with nr_cores (NNG:30).  keep_pool: the workers -- forked here, before this
    process touches the GPU -- stay for the pair-list legs (cpu_pair_legs), returned under "_pool"."""
    from multiprocessing import Pool
    cores = usable_cores()
    n = len(seqs)
    _G["seqs"] = seqs
    try:
        import edlib  # noqa: F401
        kind, fn, what = "edlib", _cpu_chunk_edlib, "real edlib %s, reference loop NNG:110-198 in Python" % getattr(edlib, "__version__", "?")
    except Exception:
        from oracle import oracle as O
        O.build()
        _G["conv"] = np.zeros(n, dtype=np.uint8)
        _G["packed"] = O.pack(seqs)            # one ASCII buffer + offsets, inherited by the workers
        kind, fn, what = "port", _cpu_chunk_port, "oracle/isocon_oracle.c orc_nn_1set (C restatement of the edlib-based loop; the edlib wheel does not import here)"
    starts = (np.random.Generator(np.random.PCG64(7)).permutation(max(n // CHUNK, 1)) * CHUNK).tolist()
    done = []
    pool = Pool(processes=cores)           # fork: the sequence list is inherited, not pickled per task
    try:
        pos = 0
        t0 = time.perf_counter()
        while pos < len(starts):
            part = starts[pos:pos + cores]
            done.extend(pool.map(fn, part, chunksize=1))
            pos += len(part)
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    finally:
        if not keep_pool:
            pool.close()
            pool.join()
    idx = np.concatenate([np.arange(s, s + len(b)) for s, b, _ in done])
    best = np.concatenate([np.asarray(b, dtype=np.int64) for _, b, _ in done])
    calls = int(sum(d[2] for d in done))
    has = best >= 0
    lo = np.searchsorted(lens, lens[idx] - np.where(has, best, 0), "left")
    hi = np.searchsorted(lens, lens[idx] + np.where(has, best, 0), "right")
    pairs = int(((hi - lo - 1) * has).sum())
    out = {"value": pairs / dt, "unit": "alignments/s", "cores": cores, "kind": kind,
           "sample": "%d of %d queries (%d random chunks of %d consecutive queries, seed 7) against the full %d-sequence set, %.1f s wall, "
                     "%d edlib-style calls (%.0f calls/s); %s; multiprocessing.Pool(%d)"
                     % (len(idx), n, len(done), CHUNK, n, dt, calls, calls / dt, what, cores)}
    if keep_pool:
        out["_pool"] = pool
    return out


def _cpu_ed_task(task):
    """one pair of edlib_align_sequences' Pool (EAM:32: the task carries both sequences, the result carries them back)"""
    (s1, s2, i, j), kw = task
    try:
        import edlib
        return s1, s2, edlib.align(s1, s2, "NW")["editDistance"]          # EAM:111
    except ImportError:
        from oracle import oracle as O
        return O._eam_task(task)


def _cpu_sw_task(task):
    """one pair of sw_align_sequences' Pool (SWM:144): parasail_alignment = full-matrix semi-global DP + traceback + cigar_to_seq +
    the two column scans (SWM:64-86)"""
    from oracle import oracle as O
    return O._swm_task(task)


def cpu_pair_legs(pool, cores, pair_ed, budget_s=10.0):
    """The pair-list half of the metric on this host's cores: the reference's Pool pattern of edlib_align_sequences (EAM:25-47) and
    sw_align_sequences (SWM:121-162) -- pool.map_async over one task per pair, each task pickling its two sequences in and the
    results (for SW: two gapped strings more) out -- on a sample of the SAME partition pairs the GPU wrappers were timed on.
    pair_ed: [(centre, member, edit distance)].  The arithmetic is real edlib if the wheel imports, else the oracle's C restatement;
    parasail does not import here (profiles/r02_probe_real_libs.json): the SW leg is the restatement (`kind: "port"`)."""
    from oracle import oracle as O
    try:
        import edlib  # noqa: F401
        ed_kind = "reference"
    except ImportError:
        ed_kind = "port"
    out = {}
    # edit distances: fast per pair -> a large sample
    n_ed = len(pair_ed)
    tasks = [((s1, s2, i, 0), {}) for i, (s1, s2, _) in enumerate(pair_ed[:n_ed])]
    t0 = time.perf_counter()
    res = pool.map_async(_cpu_ed_task, tasks).get(999999999)
    dt = time.perf_counter() - t0
    assert all(r[2] == p[2] for r, p in zip(res, pair_ed[:n_ed])), "CPU edit distances differ from the GPU's"
    out["ed_pairs"] = {"value": n_ed / dt, "unit": "pairs/s", "cores": cores, "kind": ed_kind,
                       "sample": "%d of %d partition pairs, unbounded global edit distance, Pool(%d).map_async with one task "
                                 "per pair as EAM:25-47, %.2f s wall; distances equal to the GPU's" % (n_ed, len(pair_ed), cores, dt)}
    # alignments with traceback: ~6 M cells per pair -> doubled batches of a seeded random order until the budget is used
    done = 0
    t_sw = 0.0
    batch = max(4 * cores, 64)
    order = np.random.Generator(np.random.PCG64(11)).permutation(len(pair_ed)).tolist()
    cells = 0
    while done < len(pair_ed) and t_sw < budget_s / 3:          # (batches double: the next one alone takes as long as all before it)
        part = [pair_ed[i] for i in order[done:done + batch]]
        cells += sum(len(s1) * len(s2) for s1, s2, _ in part)
        tasks = [((s1, s2, i, 0), {"mismatch_penalty": O.mismatch_penalty_for(ed, len(s1), len(s2))}) for i, (s1, s2, ed) in enumerate(part)]
        t0 = time.perf_counter()
        res = pool.map_async(_cpu_sw_task, tasks).get(999999999)
        t_sw += time.perf_counter() - t0
        done += len(part)
        batch *= 2
    out["sw_pairs"] = {"value": done / t_sw, "unit": "pairs/s", "cores": cores, "kind": "port",
                       "sample": "%d of %d partition pairs (random: the head of a seeded permutation, PCG64(11); mean len(q) x len(t) = %.3g cells), full-matrix semi-global "
                                 "alignment + traceback + gapped strings + counts (oracle orc_sg_trace, tie policy 0), Pool(%d).map_async with one task per pair as SWM:121-162, %.2f s wall"
                                 % (done, len(pair_ed), cells / max(done, 1), cores, t_sw)}
    return out


def _cpu_chunk_2set(start):
    """CHUNK consecutive entries of the merged (reads + candidates) order through the reference's 2-set loop (NNG:341-424; rows of candidates are empty)"""
    from oracle import oracle as O
    cnt = min(CHUNK, len(_G["seqs2"]) - start)
    row_ptr, cols, eds, calls = O.nn_2set(_G["seqs2"], _G["is_target2"], int(start), cnt, packed=_G["packed2"])
    return int(start), int(calls), int(len(cols))


def two_set_inputs():
    """reads and candidates of the two_set_graph leg (tests/golden/make_golden_g19.py): (X, C, merged list, is_target flags, fixture, module)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_g19", os.path.join(ROOT, "tests", "golden", "make_golden_g19.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    X, C = mod.candidates("c3")
    merged = mod.merged_list(X, C)
    z = np.load(os.path.join(ROOT, "tests", "golden", "g19_c3_graph_2set.npz"))
    return X, C, merged, z, mod


def cpu_two_set(pool, cores, budget_s=8.0):
    """get_nearest_neighbors_2set (NNG:341-424) under the reference's Pool pattern (NNG:300-334) on random chunks of CHUNK consecutive entries of the merged order
    against the full candidate set: the loop's own alignment count per second"""
    n = len(_G["seqs2"])
    starts = (np.random.Generator(np.random.PCG64(13)).permutation(max(n // CHUNK, 1)) * CHUNK).tolist()
    done = []
    pos = 0
    t0 = time.perf_counter()
    while pos < len(starts):
        part = starts[pos:pos + cores]
        done.extend(pool.map(_cpu_chunk_2set, part, chunksize=1))
        pos += len(part)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    calls = sum(d[1] for d in done)
    return {"value": calls / dt, "unit": "alignments/s", "cores": cores, "kind": "port",
            "sample": "%d of %d entries of the merged order (%d random chunks of %d consecutive entries, seed 13; the reads among them are the queries) against the %d candidates, "
                      "%.1f s wall, %d edlib-style calls of the loop; oracle/isocon_oracle.c orc_nn_2set (C restatement of NNG:341-424); multiprocessing.Pool(%d)"
                      % (len(done) * CHUNK, n, len(done), CHUNK, int(np.asarray(_G["is_target2"]).sum()), dt, calls, cores)}


def self_launch(args):
    """--gpus N > 1 without a launcher: start the N ranks as a child (this process has not imported torch)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if lines:
        print(lines[-1])
    else:
        sys.stdout.write(p.stdout)
    sys.exit(p.returncode)


def source_digest():
    """digest of the kernel sources: counters.json is only valid for the code it was collected from"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "isocon_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_counters():
    """(counters, note): the committed PMC figures, or {} when they were collected from other kernel sources"""
    try:
        c = json.load(open(COUNTERS))
    except Exception:
        return {}, "profiles/counters.json missing"
    if c.get("source_digest") != source_digest():
        return {}, "profiles/counters.json is stale: collected from kernel sources %s, this tree is %s (run scripts/profile_bench.sh + scripts/summarize_profile.py)" % (c.get("source_digest"), source_digest())
    return c, "profiles/counters.json (rocprofv3 --pmc, round %s, same kernel sources)" % c.get("round", "?")


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # the pool's driver supports dmabuf IPC only (RCCL reads it at init)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=50000)
    ap.add_argument("--length", type=int, default=2500)
    ap.add_argument("--isoforms", type=int, default=10)
    ap.add_argument("--seed", type=int, default=30001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (other kernels, wrappers)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from isocon_amd import synth

    accs, seqs_all, true_isoforms = synth.make_reads(args.reads, args.length, args.isoforms, args.seed)
    # unique strings in first-appearance order, then a STABLE sort by length (NNG:243-246).  Not set(): its iteration
    # order depends on the per-process string hash seed, and every rank must pack the very same order.
    seqs = sorted(dict.fromkeys(seqs_all), key=len)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))

    # CPU baseline first: its fork()ed workers must exist (and be gone) before this process touches the GPU
    cpu = None
    cpu_pool = None
    if rank == 0 and not args.no_cpu_baseline:          # (rank 0 of any world size; the others wait at the first barrier)
        if world == 1 and not args.no_extras and (args.reads, args.length, args.isoforms, args.seed) == (50000, 2500, 10, 30001):
            try:          # the 2-set leg's inputs, for the workers forked below (cpu_two_set)
                _X, _C, _merged, _z, _mod = two_set_inputs()
                _G["seqs2"] = [s for s, _ in _merged]
                _G["is_target2"] = np.ascontiguousarray(_z["is_target"], dtype=np.uint8)
                from oracle import oracle as _O
                _O.build()
                _G["packed2"] = _O.pack(_G["seqs2"])
            except Exception:
                _G.pop("seqs2", None)
        cpu = cpu_baseline(seqs, lens, budget_s=args.cpu_budget, keep_pool=world == 1 and not args.no_extras)
        cpu_pool = cpu.pop("_pool", None)

    import torch
    n_dev = max(torch.cuda.device_count(), 1)
    backend = os.environ.get("ISOCON_DIST_BACKEND", "nccl")   # "gloo": functional test of the N > 1 path on fewer GPUs than ranks
    if world > 1 and backend == "nccl" and world > n_dev:
        raise SystemExit("--gpus %d but only %d device(s) visible (set ISOCON_DIST_BACKEND=gloo to let ranks share a device)" % (world, n_dev))
    device_index = local_rank % n_dev
    os.environ.setdefault("ISOCON_GPU_DEVICE", str(device_index))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(device_index)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
    if torch.cuda.is_available():
        torch.cuda.set_device(device_index)

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
    phase_ms = {k: [] for k in ("scan_kernel_ms", "narrow_kernel_ms", "seed_kernel_ms", "bound_kernel_ms", "mm_kernel_ms", "list_kernel_ms", "filter_kernel_ms", "lanes_kernel_ms", "kernel_ms")}
    for _ in range(args.steps):
        step()
        for k in phase_ms:          # HIP events on the kernels' own stream (EventTimer, csrc/isocon_hip.hip)
            phase_ms[k].append(sum(x.get(k, 0.0) for x in last["stats"]))
    all_ms = phase_ms["kernel_ms"]
    sync()
    dt = time.perf_counter() - t0
    per_rank_kernel_ms = [float(np.mean(all_ms))]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        km = torch.tensor([float(np.mean(all_ms))], dtype=torch.float64, device=red_device)
        kms = [torch.zeros_like(km) for _ in range(world)]
        dist.all_gather(kms, km)
        per_rank_kernel_ms = [float(x.item()) for x in kms]
    ms_per_step = dt / args.steps * 1e3
    n_align = window_pairs(lens, last["best"])
    value = n_align / (ms_per_step / 1e3)

    # ---- roofline of the dominant kernel on this rank --------------------------------------------------------------
    st0 = {k: sum(x.get(k, 0) for x in last["stats"]) for k in ("pairs_evaluated", "cells_columns", "pairs_prefiltered", "pairs_lanes", "bound_tiles",
                                                                 "narrow_columns", "pairs_narrow", "pairs_block_rejected")}   # this rank, all phases
    pairs_eval = int(st0["pairs_evaluated"])
    mean_len = float(lens.mean())
    pm = {k: (float(np.mean(v)) if v else 0.0) for k, v in phase_ms.items()}
    is_default = (args.reads, args.length, args.isoforms, args.seed) == (50000, 2500, 10, 30001)
    ctr, ctr_note = load_counters()
    use_ctr = is_default and world == 1          # the committed PMC figures belong to the default workload on one GPU

    def valu_frac(key, ms):
        c = ctr.get(key, {})
        return float(c["SQ_INSTS_VALU"]) / (ms / 1e3) / VALU_PEAK_WAVE_INSTR if (use_ctr and c.get("SQ_INSTS_VALU") and ms > 0) else None

    def hbm(key, ms):
        c = ctr.get(key, {})
        if not (use_ctr and c.get("hbm_bytes") and ms > 0):
            return None, None
        return float(c["hbm_bytes"]), float(c["hbm_bytes"]) / (ms / 1e3) / 1e9 / HBM_PEAK_GBS

    # the dominant kernel: the bound matrix, a banded A B^T of thermometer-coded q-gram profiles on the matrix cores (k_qgram_mm)
    from isocon_amd import _lib as _l
    kk = int(_l.load().isocon_qgram_params(None))
    mm_ms = pm["mm_kernel_ms"]
    macs = float(st0["bound_tiles"]) * 65536.0 * kk
    mm_traffic, mm_hbm_frac = hbm("nn_bound", mm_ms)
    achieved = 2.0 * macs / (mm_ms / 1e3) / 1e12 if mm_ms > 0 else None
    # the second rejection test: one lane per surviving pair, the owner's 8-gram set in LDS (k_nn_block_filter)
    cf = ctr.get("nn_filter", {}) if use_ctr else {}
    filter_pass = None
    if pm["filter_kernel_ms"] > 0:
        f_ms = pm["filter_kernel_ms"]
        probes = float(st0["pairs_block_rejected"] + pairs_eval) * (mean_len / 4.0)          # upper estimate: every pair probed to its end
        filter_pass = {"kernel": "isocon::k_nn_block_filter (greedy count of disjoint 8-grams of the partner that the owner does not hold: bitmap of the owner's grams "
                                 "in LDS, one partner per lane, a probe every 4 bases; csrc/nn_filter.hpp)",
                       "bound": "lds", "kernel_ms": f_ms, "pairs_rejected": int(st0["pairs_block_rejected"]),
                       "share_of_the_survivors_rejected": float(st0["pairs_block_rejected"]) / max(1.0, float(st0["pairs_block_rejected"] + pairs_eval)),
                       "valu_frac": valu_frac("nn_filter", f_ms),
                       "lds_instructions": cf.get("SQ_INSTS_LDS"), "lds_bank_conflict_cycles": cf.get("SQ_LDS_BANK_CONFLICT"),
                       "lds_frac": ((2.0 * float(cf["SQ_INSTS_LDS"]) + float(cf.get("SQ_LDS_BANK_CONFLICT", 0.0))) / 256.0 / (f_ms / 1e3) / 2.4e9) if cf.get("SQ_INSTS_LDS") else None,
                       "lds_frac_note": "(2 cycles per conflict-free ds_read_b32 + conflict cycles) per CU against the kernel's time at 2.4 GHz (MI355X_MICROARCH.md, LDS)",
                       "probes_per_s_upper": probes / (f_ms / 1e3), "hbm_bytes": hbm("nn_filter", f_ms)[0], "hbm_frac": hbm("nn_filter", f_ms)[1]}
    # the table launches (what the filter leaves in chunks, when that is enough for a launch; every survivor without the filter)
    wave_cols_narrow = float(st0["narrow_columns"]) / 64.0
    wave_cols = float(st0["cells_columns"]) / 64.0 - wave_cols_narrow
    tables_ms = pm["scan_kernel_ms"]
    k_ms = tables_ms - pm["narrow_kernel_ms"]
    table_pass = None
    if wave_cols > 0 and int(st0["pairs_lanes"]) < pairs_eval:
        cm = ctr.get("nn_main", {})
        ipc = float(cm["SQ_INSTS_VALU"]) / float(cm["wave_columns"]) if cm.get("SQ_INSTS_VALU") and cm.get("wave_columns") else None
        table_pass = {"kernel": "isocon::k_nn_scan_refill<4, 1, false> / <8, 1, true> (64- and 32-row table kernels on the chunks the filter leaves)", "bound": "valu",
                      "kernel_ms": k_ms, "narrow_kernel_ms": pm["narrow_kernel_ms"], "wave_columns_this_run": wave_cols, "narrow_wave_columns_this_run": wave_cols_narrow,
                      "valu_insts_per_wave_column": ipc, "frac": ipc * wave_cols / (k_ms / 1e3) / VALU_PEAK_WAVE_INSTR if (ipc and k_ms > 0) else None}
    alg_bytes = pairs_eval * (2.0 * mean_len + 8.0)
    align_ms = tables_ms + pm["lanes_kernel_ms"]
    roofline = {"bound": "mfma",
                "kernel": "isocon::k_qgram_mm (q-gram bounds of every pair of the length window: banded (profiles) x (profiles)^T, fp4 MFMA 32x32x64, K = %d binary "
                          "elements per read; 256 x 256 tiles, 4-stage LDS ring filled by LDS-DMA)" % kk,
                "achieved": achieved, "peak": 2.0 * MFMA_FP4_PEAK_MACS / 1e12, "unit": "TFLOP/s",
                "frac": achieved / (2.0 * MFMA_FP4_PEAK_MACS / 1e12) if achieved else None, "traffic": mm_traffic,
                "kernel_ms": mm_ms, "tiles_256x256": int(st0["bound_tiles"]), "flops_per_launch": 2.0 * macs,
                "flops_note": "algorithmic: tiles x 65 536 pairs x K multiply-adds x 2 (a pair costs K / 2 x 2 / 256 = %.0f B of operand traffic and 2 B of output)" % (kk / 256.0),
                "hbm_frac": mm_hbm_frac, "counters": ctr_note,
                "share_of_step_kernels": mm_ms / pm["kernel_ms"] if pm["kernel_ms"] > 0 else None,
                "step_kernels_ms": {"profiles + row layout (k_qgram_profile4, k_lbt_rows)": pm["bound_kernel_ms"] - mm_ms, "bound matrix (k_qgram_mm)": mm_ms,
                                    "seeds (k_qgram_seed_pairs, k_ed_lanes)": pm["seed_kernel_ms"],
                                    "survivor lists (k_nn_entry_meta, k_nn_survivors)": pm["list_kernel_ms"] - pm["filter_kernel_ms"],
                                    "block filter (k_nn_block_filter)": pm["filter_kernel_ms"],
                                    "tables (k_nn_scan_refill; 0 = event overhead only: nothing launched)": tables_ms,
                                    "pair per lane (k_ed_lanes)": pm["lanes_kernel_ms"], "all kernels": pm["kernel_ms"]},
                "filter_pass": filter_pass, "table_pass": table_pass,
                "lanes_pass": {"kernel": "isocon::k_ed_lanes<true> (what the filter leaves + owners with few pairs: banded bit-vector edit distance, one pair per lane)", "bound": "valu",
                               "kernel_ms": pm["lanes_kernel_ms"], "pairs": int(st0["pairs_lanes"]), "valu_frac": valu_frac("nn_lanes", pm["lanes_kernel_ms"])},
                "seed_pass": {"kernel": "isocon::k_ed_lanes<true> (the smallest-bound partners of every entry, aligned first: best[] starts the pass at practically final values)",
                              "bound": "valu", "kernel_ms": pm["seed_kernel_ms"], "valu_frac": valu_frac("nn_seed", pm["seed_kernel_ms"])},
                "list_pass": {"kernel": "isocon::k_nn_survivors (one wave per entry over its row and its transposed row of the bound matrix)", "bound": "hbm",
                              "kernel_ms": pm["list_kernel_ms"] - pm["filter_kernel_ms"], "hbm_bytes": hbm("nn_lists", pm["list_kernel_ms"] - pm["filter_kernel_ms"])[0],
                              "hbm_frac": hbm("nn_lists", pm["list_kernel_ms"] - pm["filter_kernel_ms"])[1],
                              "valu_frac": valu_frac("nn_lists", pm["list_kernel_ms"] - pm["filter_kernel_ms"])},
                "pairs_inside_their_threshold_window": int(st0["pairs_prefiltered"]) + int(st0["pairs_block_rejected"]) + pairs_eval,
                "pairs_rejected_by_qgram_bound": int(st0["pairs_prefiltered"]), "pairs_rejected_by_block_filter": int(st0["pairs_block_rejected"]),
                "pairs_aligned": pairs_eval, "pairs_aligned_one_per_lane": int(st0["pairs_lanes"]),
                "algorithmic": {"bytes_per_pair": 2.0 * mean_len + 8.0, "pairs_per_launch": pairs_eval,
                                "GBps": alg_bytes / (align_ms / 1e3) / 1e9 if align_ms > 0 else None,
                                "note": "SURVEY 8(d) byte model x pairs aligned / alignment kernel time; NOT a physical rate: of the reference's alignments (config.alignments_per_step) "
                                        "all but these are decided by the two lower bounds and never aligned"}}

    result = {
        "metric": "read x candidate alignments/sec (NN-graph build, %dk x %.1fkb reads)" % (args.reads // 1000, args.length / 1000.0),
        "value": value, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": ("C3: " if is_default else "custom: ") + "%d synthetic CCS reads (%d unique), ~%d bp, %d isoforms, seed %d; 1-set NN graph"
                               % (args.reads, len(seqs), args.length, args.isoforms, args.seed),
                   "alignments_per_step": n_align, "nn_graph_wall_ms": ms_per_step, "edges": int(last["edges"]),
                   "median_nn_distance": float(np.median(last["best"][last["best"] >= 0])) if (last["best"] >= 0).any() else None,
                   "graph_digest": graph_digest(last["best"], last["row_ptr"], last["cols"]),
                   "parallelism": "1 process/GPU; pairs sharded by lower index; all_reduce(MIN)+all_gather over RCCL" if world > 1 else "single GPU"},
        "roofline": roofline,
        "rccl_ranks": dist.get_world_size() if dist is not None else 1,
        "dist_backend": dist.get_backend() if dist is not None else None,
        "per_rank_kernel_ms": per_rank_kernel_ms,
        # what a step costs beyond the slowest rank's kernels: host work of the phases, the reductions and the gather, waiting for the others
        "protocol_ms": ms_per_step - max(per_rank_kernel_ms),
    }
    if world == 1 and not args.no_extras:
        try:
            result["other_kernels"] = other_kernels(store, seqs, lens, last, true_isoforms, ctr if is_default else {})
        except Exception as e:  # the headline line must still be printed
            result["other_kernels"] = {"error": repr(e)}
        pair_ed = []
        sw_leg = {}
        try:
            result["wrappers"] = wrappers(accs, seqs_all, pair_ed, sw_leg)
        except Exception as e:
            result["wrappers"] = {"error": repr(e)}
        if is_default:
            try:
                result["two_set_graph"] = two_set_leg()
                result["roofline_2set"] = roofline_2set(result["two_set_graph"], kk, ctr if is_default else {})
            except Exception as e:
                result["two_set_graph"] = {"error": repr(e)}
        if sw_leg:
            result["roofline_sw"] = roofline_sw(sw_leg, ctr if is_default else {})
            if is_default:
                result["config"]["sw_digest"] = sw_leg["digest"]
                result["config"]["sw_digest_expected"] = EXPECTED_SW_DIGEST_C3
                result["config"]["alignments_equal_oracle_fixture"] = sw_leg["digest"] == EXPECTED_SW_DIGEST_C3
        if cpu_pool is not None and "seqs2" in _G and "error" not in result.get("two_set_graph", {"error": 1}):
            try:
                cpu["two_set"] = cpu_two_set(cpu_pool, cpu["cores"])
                result["two_set_graph"]["vs_cpu_baseline_wall"] = result["two_set_graph"]["alignments_per_s_wall"] / cpu["two_set"]["value"]
            except Exception as e:
                cpu["two_set_error"] = repr(e)
        if cpu_pool is not None and pair_ed:
            try:
                cpu.update(cpu_pair_legs(cpu_pool, cpu["cores"], pair_ed))
                w = result["wrappers"]
                w["edlib_align_sequences_vs_cpu_baseline"] = w["edlib_align_sequences_pairs_per_s"] / cpu["ed_pairs"]["value"]
                w["sw_align_sequences_vs_cpu_baseline"] = w["sw_align_sequences_pairs_per_s"] / cpu["sw_pairs"]["value"]
            except Exception as e:
                cpu["pair_legs_error"] = repr(e)
    if cpu_pool is not None:
        cpu_pool.close()
        cpu_pool.join()
    if cpu is not None:
        result["cpu_baseline"] = cpu
        result["speedup_vs_cpu_baseline"] = value / cpu["value"] if cpu["value"] else None
    if is_default:
        result["config"]["graph_digest_expected"] = EXPECTED_GRAPH_DIGEST_C3
        result["config"]["graph_equals_reference_loop_fixture"] = result["config"]["graph_digest"] == EXPECTED_GRAPH_DIGEST_C3
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()
    if is_default and result["config"]["graph_digest"] != EXPECTED_GRAPH_DIGEST_C3:
        raise SystemExit("bench.py: the graph of the default workload has digest %s, the reference-loop fixture %s: the result is WRONG, the line above is not a measurement"
                         % (result["config"]["graph_digest"], EXPECTED_GRAPH_DIGEST_C3))
    if is_default and result.get("two_set_graph", {}).get("every_row_equals_reference_loop_fixture") is False:
        raise SystemExit("bench.py: the 2-set graph of the default workload's reads against the g19 candidates differs from the reference-loop fixture: the two_set_graph leg is WRONG")
    if is_default and result["config"].get("alignments_equal_oracle_fixture") is False:
        raise SystemExit("bench.py: the alignments of the default workload's partition pairs have digest %s, the oracle fixture %s: the wrappers leg is WRONG"
                         % (result["config"]["sw_digest"], EXPECTED_SW_DIGEST_C3))


def _fracs(ctr, key, ms, own_time=False):
    """VALU-issue and HBM fractions of one profiled dispatch class.  own_time False: the class is (practically) the whole call, the
    denominators are this run's live kernel time.  own_time True: the class is ONE kernel of a call that runs several (k_sg_band of
    band + walk + compact + expand): its counters are divided by ITS OWN duration in the profiled run (`kernel_ms_profiled`), not by
    the call's kernel time."""
    c = ctr.get(key, {})
    out = {}
    t_valu = t_hbm = ms
    if own_time:
        t_valu = c["dur_ns_sq_pass"] / 1e6 if c.get("dur_ns_sq_pass") else None
        t_hbm = c["dur_ns_fetch_pass"] / 1e6 if c.get("dur_ns_fetch_pass") else t_valu
        if t_valu:
            out["kernel_ms_profiled"] = t_valu
    if c.get("SQ_INSTS_VALU") and t_valu and t_valu > 0:
        out["valu_frac"] = float(c["SQ_INSTS_VALU"]) / (t_valu / 1e3) / VALU_PEAK_WAVE_INSTR
    if c.get("hbm_bytes") and t_hbm and t_hbm > 0:
        out["hbm_frac"] = float(c["hbm_bytes"]) / (t_hbm / 1e3) / 1e9 / HBM_PEAK_GBS
        out["hbm_bytes"] = float(c["hbm_bytes"])
    return out


def other_kernels(store, seqs, lens, last, true_isoforms, ctr):
    """Not part of the timed region: the other kernels of the path on (read, first NN) pairs of the same set."""
    from isocon_amd.store import SeqStore
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
    # the shape isocon_hw_pairs is built for: the candidate-vs-candidate graph of the statistical test (get_all_NN,
    # end_invariant_functions.py:622-681) -- ~4 900 near-identical candidates of the 10 isoforms (a few residual errors, ragged
    # ends), every candidate against its length window 10 + 2 x 15, k = 25: millions of pairs, full tiles
    from isocon_amd import end_invariant_functions as END
    from isocon_amd import synth
    grng = np.random.Generator(np.random.PCG64(77))
    cset = set()
    for iso in true_isoforms:
        arr = np.frombuffer(iso.encode("ascii"), dtype=np.uint8)
        for _ in range(490):
            v = synth.mutate(grng, arr, dict(rate=0.0012, ins=0.4, dele=0.4, sub=0.2))
            a, b = int(grng.integers(0, 12)), int(grng.integers(0, 12))
            cset.add(v[a:len(v) - b].tobytes().decode())
    cseqs = sorted(cset, key=len)
    clens = np.fromiter((len(x) for x in cseqs), dtype=np.int64, count=len(cseqs))
    gq, gt = END._window_pairs(clens, 0, len(cseqs), 40, 2 ** 32)
    stg = SeqStore(cseqs)
    gk = np.full(len(gq), 25, dtype=np.int32)
    stg.hw_pairs(gq[:4096], gt[:4096], gk[:4096])
    stg.hw_pairs(gq, gt, gk, reuse_buffer=True)          # (as end_invariant_functions.get_all_NN calls it; the first call pins the result buffer)
    t0 = time.perf_counter(); gres, g_ms = stg.hw_pairs(gq, gt, gk, return_ms=True, reuse_buffer=True); g_wall = time.perf_counter() - t0
    stg.close()
    # the byte-wise kernel (sets with more than four distinct symbols, csrc/ed_bytes.hpp): the same pairs with an 'N' put into every first sequence
    sub = sorted(set(t.tolist()) | set(q.tolist()))
    pos = {v: i for i, v in enumerate(sub)}
    tset = set(t.tolist())
    stb = SeqStore([seqs[v][:len(seqs[v]) // 2] + "N" + seqs[v][len(seqs[v]) // 2 + 1:] if v in tset else seqs[v] for v in sub])
    bt = np.fromiter((pos[v] for v in t.tolist()), dtype=np.uint32, count=len(t))
    bq = np.fromiter((pos[v] for v in q.tolist()), dtype=np.uint32, count=len(q))
    k63 = np.full(len(q), 63, dtype=np.int32)
    stb.ed_pairs(bt[:64], bq[:64], k63[:64])
    edb, edb_ms = stb.ed_pairs(bt, bq, k63, return_ms=True)
    stb.close()
    # the read -> candidate (2-set) search of the pipeline's last steps: all reads against the true isoforms
    cands = [c for c in dict.fromkeys(true_isoforms) if c not in set(seqs)]
    merged = sorted([(s, 0) for s in seqs] + [(c, 1) for c in cands], key=lambda x: len(x[0]))
    st2 = SeqStore([s for s, _ in merged])
    is_t = np.array([f for _, f in merged], dtype=np.uint8)
    st2.nn_graph(is_target=is_t)
    t0 = time.perf_counter(); b2, rp2, c2, stats2 = st2.nn_graph(is_target=is_t); two_wall = time.perf_counter() - t0
    st2.close()
    out = {
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
        "sw_banded_pairs_per_s_kernel": len(q) / (swb_ms / 1e3) if swb_ms > 0 else None, "sw_banded_wall_ms": swb_wall * 1e3,
        "sw_banded_equals_full": same,
        # VALU-issue / HBM fractions of the profiled dispatches of THIS batch (profiles/counters.json), at the live kernel times
        # (call_kernel_ms = all kernels of the call: forward / band + walk + compact + expand; the fractions belong to the named kernel alone)
        "sw_full": dict(kernel="k_sg_forward (4096 pairs, full matrix)", call_kernel_ms=sw_ms, **_fracs(ctr, "sg_full", sw_ms, own_time=True)),
        "sw_banded": dict(kernel="k_sg_band (4096 pairs, edit-distance band hints: the band's diagonals on the lanes)", call_kernel_ms=swb_ms,
                          **_fracs(ctr, "sg_banded", swb_ms, own_time=True)),
        "hw_k25": dict(kernel="infix kernel, 4096 pairs, k = 25", kernel_ms=hw25_ms, **_fracs(ctr, "hw_k25", hw25_ms)),
        "hw_k63": dict(kernel="infix kernel, 4096 pairs, k = 63", kernel_ms=hw63_ms, **_fracs(ctr, "hw_k63", hw63_ms)),
        "ed_bytes": dict(kernel="k_ed_bytes (one wavefront per pair on the sequences' bytes; 4096 pairs with an 'N' in one sequence, k = 63; VALU-bound)", kernel_ms=edb_ms,
                         pairs_per_s_kernel=len(q) / (edb_ms / 1e3) if edb_ms > 0 else None, hits=int((edb >= 0).sum()),
                         within_two_of_the_acgt_distance=bool((np.abs(edb - ed)[(edb >= 0) & (ed <= 61)] <= 2).all())),          # (one 'N' per sequence, both sequences of a pair may carry one)
        "hw_graph": dict(kernel="k_hw_locate + k_hw_finish, candidate-vs-candidate graph (%d candidates, k = 25)" % len(cseqs), pairs=int(len(gq)),
                         hits=int((gres[:, 0] >= 0).sum()), wall_ms=g_wall * 1e3, kernel_ms=g_ms, pairs_per_s_kernel=len(gq) / (g_ms / 1e3) if g_ms > 0 else None,
                         pairs_per_s_wall=len(gq) / g_wall, **_fracs(ctr, "hw_graph", g_ms))}
    return out


def roofline_sw(leg, ctr):
    """The alignment dispatch of the wrappers leg -- isocon_sg_trace_batch on ALL partition pairs of the workload (C3: 49 990), with the pairs'
    distances as band hints, as get_partition_alignments / sw_align_sequences call it.  SURVEY 8(d): the trace kernel is the one kernel of the path
    whose stated bound is HBM (4 bit / cell trace stream); reported: the HBM fraction from the PMC bytes of THAT dispatch of k_sg_band
    (profiles/counters.json `sg_partition`) over its live event time, its VALU fraction, bytes per pair against the 10.2 kB of the byte model,
    cell updates / s (full-matrix equivalent, what parasail would compute) and band cells / s (what the kernel computes)."""
    c = ctr.get("sg_partition", {})
    fwd_ms, n = leg["stats"]["forward_ms"], leg["pairs"]
    own_ms = c["dur_ns_sq_pass"] / 1e6 if c.get("dur_ns_sq_pass") else None
    hbm = float(c["hbm_bytes"]) if c.get("hbm_bytes") else None
    insts = float(c["SQ_INSTS_VALU"]) if c.get("SQ_INSTS_VALU") else None
    alg = leg["algorithmic_bytes"]
    return {"bound": "hbm", "what_binds_it": "VALU issue (valu_frac); SURVEY 8(d) states HBM as the bound of the trace kernel, so frac is the HBM fraction",
            "kernel": "isocon::k_sg_band<true, true, 4 | 2> (%d partition pairs of the workload: the certified band's diagonals on the lanes, four per lane for bands of up to 256 diagonals, "
                      "two for up to 128 -- and for wider bounds whose middle 128 diagonals certify themselves; 4-bit trace per cell; the launches of the call, one per class "
                      "and one for the pairs that run again)" % n,
            "pairs": n, "kernel_ms": fwd_ms, "call_kernel_ms": leg["kernel_ms"], "call_wall_ms": leg["wall_ms"],
            "kernels_ms": {k: leg["stats"][k] for k in ("forward_ms", "walk_ms", "compact_ms", "expand_ms")},
            "pairs_band": int(leg["stats"]["pairs_band"]), "pairs_band_two_diagonals_per_lane": int(leg["stats"].get("pairs_band_narrow", 0)),
            "pairs_tried_on_128_diagonals_first": int(leg["stats"].get("pairs_tried_narrow", 0)), "pairs_run_again_on_256": int(leg["stats"].get("pairs_retried_wider", 0)),
            "pairs_strips": int(leg["stats"]["pairs_strips"]), "pairs_redone_in_full": int(leg["stats"]["pairs_redone"]),
            "achieved": hbm / (fwd_ms / 1e3) / 1e9 if hbm and fwd_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": hbm / (fwd_ms / 1e3) / 1e9 / HBM_PEAK_GBS if hbm and fwd_ms > 0 else None, "traffic": hbm,
            "traffic_per_pair": hbm / n if hbm else None, "trace_scratch_bytes_per_pair": leg["stats"]["trace_bytes"] / n,
            "algorithmic_bytes_per_pair": alg / n, "traffic_vs_algorithmic": hbm / alg if hbm else None,
            "algorithmic_frac": alg / (fwd_ms / 1e3) / 1e9 / HBM_PEAK_GBS if fwd_ms > 0 else None,
            "valu_frac": insts / (fwd_ms / 1e3) / VALU_PEAK_WAVE_INSTR if insts and fwd_ms > 0 else None,
            "valu_frac_profiled_pass": insts / (own_ms / 1e3) / VALU_PEAK_WAVE_INSTR if insts and own_ms else None,
            "cell_updates_per_s": leg["cells_full"] / (leg["kernel_ms"] / 1e3) if leg["kernel_ms"] > 0 else None,
            "band_cells_per_s": leg["cells_band"] / (fwd_ms / 1e3) if fwd_ms > 0 else None,
            "pairs_per_s_kernel": n / (leg["kernel_ms"] / 1e3) if leg["kernel_ms"] > 0 else None,
            "digest": leg["digest"],
            "note": "algorithmic bytes = SURVEY 8(d): len(q) + len(t) + 2 len(alignment) + 12 per pair; traffic = PMC 2 x FETCH_SIZE + WRITE_SIZE of the dispatch; "
                    "cell_updates_per_s counts len(q) x len(t) per pair over ALL kernels of the call, band_cells_per_s the 128 (four diagonals per lane) or 64 (two) cells per anti-diagonal the band kernels compute"}


def two_set_leg():
    """The search the metric is named after, at the size it is quoted on: compute_2set_nearest_neighbor_graph (NNG:201-234) of the workload's
    50 000 reads against the seeded candidate set of tests/golden/g19 (1 030 candidates: the isoforms, variants a few edits away, 20 reads),
    strings in, dict of dicts out -- EVERY read's row compared with the fixture the oracle's reference loop produced on the CPU
    (tests/golden/make_golden_g19.py; `reference_loop_alignments` is that loop's own count of edlib calls, NNG:387/403)."""
    from isocon_amd import nearest_neighbor_graph as NNG
    X, C, merged, z, mod = two_set_inputs()
    if str(z["inputs_sha1"]) != mod.inputs_sha1(merged):
        return {"error": "tests/golden/g19_c3_graph_2set.npz belongs to another read / candidate set"}

    class P(object):
        nr_cores = 1
        neighbor_search_depth = 2 ** 32
        verbose = False

    t0 = time.perf_counter(); graph = NNG.compute_2set_nearest_neighbor_graph(X, C, P()); t_first = time.perf_counter() - t0
    del graph
    t0 = time.perf_counter(); graph = NNG.compute_2set_nearest_neighbor_graph(X, C, P()); t = time.perf_counter() - t0
    kern = float(NNG.LAST_STATS.get("kernel_ms", 0.0))
    ls = dict(NNG.LAST_STATS)
    pos = {a: i for i, (_, a) in enumerate(merged)}
    best, row_ptr, cols = z["best"], z["row_ptr"], z["cols"]
    is_t = z["is_target"]
    same = list(graph) == [a for (_, a), tflag in zip(merged, is_t.tolist()) if not tflag]
    edges = 0
    for a, nbrs in graph.items():
        i = pos[a]
        edges += len(nbrs)
        if [pos[b] for b in nbrs] != cols[row_ptr[i]:row_ptr[i + 1]].tolist() or any(d != int(best[i]) for d in nbrs.values()):
            same = False
    calls = int(z["edlib_calls"])
    return {"reads": len(X), "candidates": len(C), "edges": edges, "reference_loop_alignments": calls,
            "wall_ms": t * 1e3, "first_call_wall_ms": t_first * 1e3, "kernel_ms": kern,
            "alignments_per_s_wall": calls / t if t > 0 else None, "alignments_per_s_kernel": calls / (kern / 1e3) if kern > 0 else None,
            "every_row_equals_reference_loop_fixture": bool(same and edges == len(cols)),
            "stats": {k: (float(v) if isinstance(v, float) else int(v)) for k, v in ls.items()},
            "note": "public function end to end (Python strings in, dict of dicts out); fixture tests/golden/g19_c3_graph_2set.npz"}


def roofline_2set(leg, kk, ctr=None):
    """Where the 2-set leg's kernel time goes (HIP events of its own call) and why it costs more per REFERENCE alignment than the 1-set step."""
    st = leg.get("stats") or {}
    if not st:
        return None
    phases = {"profiles + row layout": st.get("bound_kernel_ms", 0.0) - st.get("mm_kernel_ms", 0.0), "bound matrix (k_qgram_mm)": st.get("mm_kernel_ms", 0.0),
              "seeds (k_ed_lanes)": st.get("seed_kernel_ms", 0.0), "survivor lists (k_nn_survivors)": st.get("list_kernel_ms", 0.0) - st.get("filter_kernel_ms", 0.0),
              "block filter (k_nn_block_filter)": st.get("filter_kernel_ms", 0.0), "tables (k_nn_scan_refill)": st.get("scan_kernel_ms", 0.0), "pair per lane (k_ed_lanes)": st.get("lanes_kernel_ms", 0.0)}
    dom = max(phases, key=lambda k: phases[k])
    macs = float(st.get("bound_tiles", 0)) * 65536.0 * kk
    mm_ms = st.get("mm_kernel_ms", 0.0)
    window_pairs_all = int(st.get("pairs_prefiltered", 0)) + int(st.get("pairs_block_rejected", 0)) + int(st.get("pairs_evaluated", 0))
    return {"bound": "mfma", "kernel": "isocon::k_qgram_mm (the same bound kernel as the 1-set step)", "dominant_phase": dom, "kernels_ms": phases, "kernel_ms": mm_ms,
            "achieved": 2.0 * macs / (mm_ms / 1e3) / 1e12 if mm_ms > 0 else None, "peak": 2.0 * MFMA_FP4_PEAK_MACS / 1e12, "unit": "TFLOP/s",
            "frac": (2.0 * macs / (mm_ms / 1e3) / 1e12) / (2.0 * MFMA_FP4_PEAK_MACS / 1e12) if mm_ms > 0 else None,
            "traffic": float(ctr["nn_2set"]["hbm_bytes"]) if ctr and ctr.get("nn_2set", {}).get("hbm_bytes") else None,
            "hbm_frac": float(ctr["nn_2set"]["hbm_bytes"]) / (mm_ms / 1e3) / 1e9 / HBM_PEAK_GBS if ctr and ctr.get("nn_2set", {}).get("hbm_bytes") and mm_ms > 0 else None,
            "valu_frac_of_the_bound_kernel": float(ctr["nn_2set"]["SQ_INSTS_VALU"]) / (mm_ms / 1e3) / VALU_PEAK_WAVE_INSTR if ctr and ctr.get("nn_2set", {}).get("SQ_INSTS_VALU") and mm_ms > 0 else None,
            "tiles_256x256": int(st.get("bound_tiles", 0)),
            "read_x_candidate_pairs_with_a_role": window_pairs_all, "pairs_rejected_by_qgram_bound": int(st.get("pairs_prefiltered", 0)),
            "pairs_rejected_by_block_filter": int(st.get("pairs_block_rejected", 0)), "pairs_aligned": int(st.get("pairs_evaluated", 0)),
            "reference_loop_alignments": leg.get("reference_loop_alignments"),
            "why_more_per_reference_alignment": "the bound matrix is built over EVERY pair of the merged length-sorted order (reads x reads included: the tiles do not know roles), "
                                                "while the reference's 2-set loop only visits read x candidate pairs: the same %d tiles as the 1-set step for %.1f %% of its "
                                                "alignments; the candidates (about 100 near-identical variants per isoform) also sit at every read's threshold, so about %d of them per "
                                                "read survive both bounds and are aligned" % (int(st.get("bound_tiles", 0)), 100.0 * float(leg.get("reference_loop_alignments") or 0) / 5.93e8,
                                                                                            int(st.get("pairs_evaluated", 0)) // max(1, int(leg.get("reads") or 1)))}


def wrappers(accs, seqs_all, pair_ed_out=None, sw_leg=None):
    """SURVEY 8(d) "Timers": perf_counter() around the PUBLIC functions (string handling, packing, H2D, kernels, D2H and the
    dict rebuild included) -- NNG:237-296, EAM:10-49, SWM:89-164 -- on the workload and on its partition pair list."""
    from isocon_amd import SW_alignment_module as SWM
    from isocon_amd import edlib_alignment_module as EAM
    from isocon_amd import nearest_neighbor_graph as NNG
    from isocon_amd import partitions

    class P(object):
        nr_cores = 1
        neighbor_search_depth = 2 ** 32
        verbose = False

    S = dict(zip(accs, seqs_all))
    # every public function twice: a process' first call also grows scratch pools and pinned buffers (kept for the following calls --
    # the pipeline calls each of them once per correction step); both times are reported, the second is "the" wall time
    t0 = time.perf_counter(); G, isolated = NNG.compute_nearest_neighbor_graph(S, set(), P()); t_nn_first = time.perf_counter() - t0
    del G
    t0 = time.perf_counter(); G, isolated = NNG.compute_nearest_neighbor_graph(S, set(), P()); t_nn = time.perf_counter() - t0
    kern = float(NNG.LAST_STATS.get("kernel_ms", 0.0))
    n_edges = sum(len(v) for v in G.values())
    del G
    t0 = time.perf_counter(); G_star, partition, M, converged = partitions.partition_strings(S, P()); t_part = time.perf_counter() - t0
    n_pairs = sum(len(v) for v in partition.values())
    t0 = time.perf_counter(); ed = EAM.edlib_align_sequences(partition); t_ed_first = time.perf_counter() - t0
    del ed
    t0 = time.perf_counter(); ed = EAM.edlib_align_sequences(partition); t_ed = time.perf_counter() - t0
    # (sw_align_sequences' first call allocates the pinned output buffers: 2 x 146 MB, ~0.1 s)
    t0 = time.perf_counter(); sw = SWM.sw_align_sequences(ed); t_sw_first = time.perf_counter() - t0
    del sw
    t0 = time.perf_counter(); sw = SWM.sw_align_sequences(ed); t_sw = time.perf_counter() - t0
    n_sw = sum(len(v) for v in sw.values())
    if pair_ed_out is not None:          # the pair list with its distances, for the CPU legs (cpu_pair_legs)
        pair_ed_out.extend((s1, s2, d) for s1, row in ed.items() for s2, d in row.items())
    del sw
    # one correction iteration of the candidate phase as isocon_get_candidates.find_candidate_transcripts runs it:
    # partition_strings (NN graph + partition) -> get_partition_alignments (distances, CIGAR ops, exon filter) -> correct_strings
    from isocon_amd import correction_module as COR
    from isocon_amd import isocon_get_candidates as IGC

    class Q(P):
        min_exon_diff = 20
        ignore_ends_len = 15

    G_star, partition, M, converged = partitions.partition_strings(S, Q())          # (the store of this very set must be the remembered one)
    # twice, like sw_align_sequences above: the first call of a process also grows the alignment scratch (the trace buffer: gigabytes of hipMalloc)
    t0 = time.perf_counter(); pa = IGC.get_partition_alignments(partition, M, G_star, set(), Q()); t_pa_first = time.perf_counter() - t0
    del pa
    t0 = time.perf_counter(); pa = IGC.get_partition_alignments(partition, M, G_star, set(), Q()); t_pa = time.perf_counter() - t0
    seq_to_acc = IGC.get_unique_seq_accessions(S)
    t0 = time.perf_counter(); S_prime, _ = COR.correct_strings(pa, seq_to_acc, {}, 1); t_cor = time.perf_counter() - t0
    batch = getattr(pa, "batch", None)
    if sw_leg is not None and batch is not None and batch.alive():
        # the alignment half of the metric, pinned: digest of (pair, score, end cell, counts, ops) over the whole partition pair list (ids = positions
        # in the length-sorted unique entries), and the dispatch `roofline_sw` reports: the same pairs through the C ABI once more, alone
        entries = sorted(dict.fromkeys(seqs_all), key=len)
        idx = {x: i for i, x in enumerate(entries)}
        ia = np.fromiter((idx[m] for m, _ in batch.pairs), dtype=np.int64, count=len(batch.pairs))
        ib = np.fromiter((idx[x] for _, x in batch.pairs), dtype=np.int64, count=len(batch.pairs))
        sw_leg["digest"] = sw_digest(ia, ib, batch.res, sw_pair_hashes(batch.ops, batch.ops_ptr))
        from isocon_amd.store import sg_last_stats
        st, a, b = batch.store, batch.a, batch.b
        ed = st.ed_pairs(a, b, None)
        la, lb = st.lens[a].astype(np.int64), st.lens[b].astype(np.int64)
        rate = ed.astype(np.float64) / np.minimum(la, lb).astype(np.float64)
        mm = np.where(rate <= 0.01, -1, np.where(rate <= 0.09, -2, -4)).astype(np.int8)
        st.sg_trace(a, b, mm, ed_upper=ed)
        t0 = time.perf_counter(); ops, ptr, res, ms = st.sg_trace(a, b, mm, ed_upper=ed, return_ms=True); wall = time.perf_counter() - t0
        aln_len = res[:, 3].astype(np.int64) + res[:, 4] + res[:, 5]
        # the band class of every pair as csrc/sg_host.inc decides it: X = ceil((match + Q) ed / match) - |D| + 1, diagonals = |D| + 2 X + 1
        aD = np.abs(la - lb)
        Qp = np.maximum(-mm.astype(np.int64), 2)
        X = np.maximum(((2 + Qp) * ed.astype(np.int64) + 1) // 2 - aD + 1, 1)
        diags = aD + 2 * X + 1
        # (a bound of 129 .. 256 diagonals runs on 128 first when that keeps >= 85 % of X; the few that do not certify themselves run again on 256)
        tried = (diags > 128) & (diags <= 256) & ((127 - aD) // 2 >= 1) & (((127 - aD) // 2) * 100 >= 85 * X)
        narrow = (diags <= 128) | tried
        sw_leg.update(pairs=len(a), kernel_ms=float(ms), wall_ms=wall * 1e3, stats=sg_last_stats(), cells_full=float((la * lb).sum()),
                      cells_band=float(((la + lb) * np.where(narrow, 64, 128)).sum()), tried_model=int(tried.sum()), algorithmic_bytes=float((la + lb + 2 * aln_len + 12).sum()),
                      same_as_wrappers=bool((res == batch.res).all() and len(ops) == len(batch.ops) and (ops == batch.ops).all()))
    return {"compute_nearest_neighbor_graph_wall_ms": t_nn * 1e3, "compute_nearest_neighbor_graph_kernel_ms": kern, "nn_edges": n_edges,
            "partition_centres": len(partition), "partition_pairs": n_pairs,
            "compute_nearest_neighbor_graph_first_call_wall_ms": t_nn_first * 1e3,
            "edlib_align_sequences_wall_ms": t_ed * 1e3, "edlib_align_sequences_first_call_wall_ms": t_ed_first * 1e3,
            "edlib_align_sequences_pairs_per_s": n_pairs / t_ed if t_ed > 0 else None,
            "sw_align_sequences_wall_ms": t_sw * 1e3, "sw_align_sequences_first_call_wall_ms": t_sw_first * 1e3,
            "sw_align_sequences_pairs_per_s": n_sw / t_sw if t_sw > 0 else None,
            "partition_strings_wall_ms": t_part * 1e3, "get_partition_alignments_wall_ms": t_pa * 1e3,
            "get_partition_alignments_first_call_wall_ms": t_pa_first * 1e3, "correct_strings_wall_ms": t_cor * 1e3,
            "corrected_reads": len(S_prime),
            "note": "public functions end to end (dict of 2.5 kb strings in, dict out); the timed region above starts with the store resident"}


if __name__ == "__main__":
    main()
