#!/usr/bin/env python3
"""Golden fixture tests/golden/g20_candidates_c5shape.json: the reference's own modules/isocon_get_candidates.py::find_candidate_transcripts
on reads of BASELINE configs[4]'s SHAPE above toy size -- ONT error profile (6 %), 5 gene families of 1-5 kb, 50 isoforms, seed 50001,
the first N_READS reads of that generator -- i.e. the whole candidate phase (partition / align / correct until convergence, naming,
end-invariant collapse with the default ignore_ends_len = 15, read-to-candidate alignment), imported from /root/reference with
tests/golden/shims standing in for the absent edlib / parasail wheels (they forward to the CPU oracle: arithmetic pinned by definition,
alignment tie-breaks "parity unpinned").  nr_cores = 8: the reference's own Pool fan-out (NNG:19-82, EAM:25-47, SWM:121-162).
Stored PER HASH SEED (the reference's partition step iterates over sets of sequences: above toy size its outcome depends on PYTHONHASHSEED --
3 000 reads: 607 candidates under seed 0, 618 under seed 1): the converged candidates (accession, digest, length), the number of correction
steps, the candidates written after every step and their digests, the reads left to realign.

The 200 000-read run of configs[4] (tests/test_gpu_c5_full.py) is checked by invariants only; this fixture is the largest read set of that
shape on which every output of the candidate phase is compared with the reference's.

Usage:  python tests/golden/make_golden_g20.py [N_READS]          (build container only; ~tens of CPU-minutes)
"""
import contextlib
import glob
import hashlib
import io
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
N_READS = 3000
SEEDS = (0, 1)          # PYTHONHASHSEED values the outcome must not depend on


def sha(s):
    return hashlib.sha1(s.encode()).hexdigest()[:16]


def reads(n):
    sys.path.insert(0, ROOT)
    from isocon_amd import synth
    accs, seqs, _ = synth.make_reads(n, 0, 50, 50001, profile=synth.ONT_PROFILE, families=5, length_range=(1000, 5000))
    return dict(zip(accs, seqs))


def child(n):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "shims"))
    sys.path.insert(0, REF)
    import networkx
    if not hasattr(networkx.Graph, "node"):
        networkx.Graph.node = property(lambda g: g.nodes)
    with contextlib.redirect_stdout(io.StringIO()):
        from modules import isocon_get_candidates as R_IGC
    S = reads(n)
    with tempfile.TemporaryDirectory() as tmp:
        read_file = os.path.join(tmp, "reads.fa")
        with open(read_file, "w") as fh:
            for acc, seq in S.items():
                fh.write(">%s\n%s\n" % (acc, seq))

        class Params(object):
            nr_cores = 8
            neighbor_search_depth = 2 ** 32
            verbose = False
            develop_logfile = None
            logfile = open(os.path.join(tmp, "log.txt"), "w")
            min_exon_diff = 20
            ignore_ends_len = 15
            min_candidate_support = 2
            is_fastq = False
            ccs = None
            outfolder = tmp

        with contextlib.redirect_stdout(io.StringIO()):
            cand_file, read_partition, to_realign = R_IGC.find_candidate_transcripts(read_file, Params())
        cands = []
        acc = None
        for line in open(cand_file):
            if line.startswith(">"):
                acc = line[1:].strip()
            else:
                cands.append([acc, sha(line.strip()), len(line.strip())])
        step_files = sorted(glob.glob(os.path.join(tmp, "candidates_step_*.fa")), key=lambda f: int(f.rsplit("_", 1)[1].split(".")[0]))
        per_step = [sum(1 for ln in open(f) if ln.startswith(">")) for f in step_files]
        step_sets = [sorted(sha(ln.strip()) for ln in open(f) if not ln.startswith(">")) for f in step_files]
        steps = 1 + len(step_files)
    rp = sorted([c, r, sha(t[0]), sha(t[1]), list(t[2])] for c in read_partition for r, t in read_partition[c].items())
    sys.stdout.write(json.dumps({"candidates": cands, "read_partition": rp, "to_realign": sorted(to_realign), "steps": steps, "candidates_per_step": per_step, "candidate_digests_per_step": step_sets}))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else N_READS
    outs = []
    for seed in SEEDS:
        t0 = time.time()
        env = dict(os.environ, PYTHONHASHSEED=str(seed))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n)], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            raise SystemExit("child failed")
        outs.append(r.stdout)
        e = json.loads(r.stdout)
        print("seed %d: %d reads -> %d candidates, %d steps %s, %d assigned, %d to realign (%.0f s)" % (
            seed, n, len(e["candidates"]), e["steps"], e["candidates_per_step"], len(e["read_partition"]), len(e["to_realign"]), time.time() - t0), flush=True)
    agree = all(o == outs[0] for o in outs)
    print("hash seeds agree:", agree)
    # The reference's partition step iterates over sets of sequences (modules/partitions.py:319-343): above toy size its candidates depend
    # on PYTHONHASHSEED.  Every seed's outcome is stored; the test compares exactly what all seeds agree on (the leading steps) and asks of the
    # rest that the build's (deterministic) outcome is as close to each seed's as the seeds are to each other.
    by_seed = {}
    for seed, o in zip(SEEDS, outs):
        e = json.loads(o)
        by_seed[str(seed)] = {"candidates": e["candidates"], "steps": e["steps"], "candidates_per_step": e["candidates_per_step"],
                              "candidate_digests_per_step": e["candidate_digests_per_step"], "assigned": len(e["read_partition"]),
                              "to_realign": e["to_realign"], "read_partition": e["read_partition"] if agree else None}
    S = reads(n)
    h = hashlib.sha1()
    for a, s in S.items():
        h.update(a.encode()); h.update(b"\t"); h.update(s.encode()); h.update(b"\n")
    json.dump({"generator": "tests/golden/make_golden_g20.py %d" % n, "reads": "isocon_amd.synth.make_reads(%d, 0, 50, 50001, profile=ONT_PROFILE, families=5, length_range=(1000, 5000))" % n,
               "n_reads": n, "inputs_sha1": h.hexdigest(), "hash_seeds": list(SEEDS),
               "params": {"ignore_ends_len": 15, "min_exon_diff": 20, "min_candidate_support": 2, "nr_cores": 8}, "hash_seeds_agree": agree, "by_seed": by_seed},
              open(os.path.join(HERE, "g20_candidates_c5shape.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
