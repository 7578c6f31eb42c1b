"""Emulates the N-rank sharded NN search on ONE GPU: the ranks' phases are run one after the other and the min-reductions
are done on the host.  Prints, per world size, the summed lane-columns (work inflation from the staler thresholds) and
the largest per-rank kernel time of each phase (what an N-GPU step would wait for)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from isocon_amd import synth, _lib
from isocon_amd.dist import protocol_steps, shard_of
from isocon_amd.store import SeqStore, nn_finalize
if len(sys.argv) > 1 and sys.argv[1] in ("c5", "c5s"):          # 200 000 mixed-length reads, 6 % errors (BASELINE.json configs[4]); c5s: 50 000 of that shape
    accs, seqs, _ = synth.make_reads(200000 if sys.argv[1] == "c5" else 50000, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
else:
    accs, seqs, _ = synth.make_reads(50000, 2500, 10, 30001)
seqs = sorted(dict.fromkeys(seqs), key=len)
st = SeqStore(seqs)
n = st.n
SUB = int(os.environ.get("PHASE2_STEPS", "1"))          # sub-steps of phase 2 (dist.protocol_steps); the product default is 1
b0, rp0, c0, s0 = st.nn_graph()
print("single: kernel %.1f ms lane-cols %.3g pairs %.4g" % (s0["kernel_ms"], s0["cells_columns"], s0["pairs_evaluated"]))
for world in (2, 4, 8):
    best = np.full(n, _lib.NN_INF, dtype=np.int32)
    tot_cols = 0; tot_pairs = 0; crit = 0.0; walls = []
    hits_all = []
    # dist.sharded_nn_graph's steps: phases 0, 1 on the rank's shard, phase 2 in sub-steps with a reduction after each (dist.protocol_steps);
    # FUSED=1: seeds + pass in one call, no exchange between them; CYCLIC=1: entry-cyclic ownership of rounds 2-3; PHASE2_STEPS=k: phase 2 in k sub-steps
    n_steps = len(protocol_steps(0, world, n, SUB))
    wide = None
    for k in range(n_steps):
        phase = protocol_steps(0, world, n, SUB)[k][0]
        if os.environ.get("FUSED") and phase == 0:
            continue
        if phase == 2 and wide is None:
            wide = ((best[:n] == _lib.NN_INF) & (np.asarray(st.lens)[:n] > 63)).astype(np.uint8)
        if phase == 2 and not wide.any():
            continue
        bests = []; kms = []
        for r in range(world):
            b = best.copy()
            ph, (qb, qe, qs, qk) = protocol_steps(r, world, n, SUB)[k]
            if os.environ.get("FUSED") and ph == 1:
                ph = 3
            if os.environ.get("CYCLIC"):
                qb, qe, qs, qk = r, n, world, 1
            if ph == 1 and not os.environ.get("NO_REUSE"):
                # A real rank is a process of its own: the bound matrix its phase 0 built is still in its scratch pool when phase 1 starts and is
                # reused (csrc/nn_bounds.inc: bound_tag).  Here all ranks share ONE pool, so the rank's phase 0 is repeated (untimed, results dropped)
                # to put its matrix back.  NO_REUSE=1: as before round 4's last measurements (every phase 1 rebuilds its matrix: + 1.2 ms at 8 ranks).
                st.nn_partial(qb, qe, 0, np.full(n, _lib.NN_INF, dtype=np.int32), q_stride=qs, q_block=qk)
            t = time.time(); hits, stats = st.nn_partial(qb, qe, ph, b, q_stride=qs, q_block=qk, wide_queries=wide if ph == 2 else None); walls.append(time.time() - t)
            bests.append(b); kms.append(stats["kernel_ms"]); tot_cols += stats["cells_columns"]; tot_pairs += stats["pairs_evaluated"]
            hits_all.append(hits)
        best = np.minimum.reduce(bests)
        crit += max(kms)
        print("  world %d step %d (phase %d): per-rank kernel ms max %.2f mean %.2f (max/mean %.3f); call wall max %.2f ms" % (world, k, phase, max(kms), np.mean(kms), max(kms) / max(np.mean(kms), 1e-9), 1e3 * max(walls[-world:])))
    hits = np.concatenate(hits_all)
    keep = (hits[:, 2] >= 0) & (hits[:, 2] == best[hits[:, 0]])
    out = nn_finalize(n, best, hits[keep])
    ok = (out[0] == b0).all() and (out[1] == rp0).all() and (out[2] == c0).all()
    print("world %d: graph identical %s; sum lane-cols %.3g (x%.3f of single), pairs %.4g; critical-path kernel time %.1f ms" %
          (world, ok, tot_cols, tot_cols / s0["cells_columns"], tot_pairs, crit))
