#!/usr/bin/env python3
"""Golden fixture tests/golden/g12_candidates.json: the reference's own
modules/isocon_get_candidates.py::find_candidate_transcripts (the whole candidate-inference phase: partition / align /
correct until convergence, candidate naming, read-to-candidate alignment) on its public test FASTA (n = 200) and on a
synthetic read set, with ignore_ends_len = 0 and 15 (the default; adds the end-invariant collapse of candidates), under
PYTHONHASHSEED 0..2 (kept if all agree).  edlib / parasail are absent: tests/golden/shims forward to the CPU oracle
(alignment tie-breaks "parity unpinned").  Stored: the converged candidates (accession, digest, length), the
read -> candidate assignment with alignment digests, the reads left to realign, the number of correction steps.

Usage:  python tests/golden/make_golden_candidates.py          (build container only)
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

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def sha(s):
    return hashlib.sha1(s.encode()).hexdigest()[:16]


def inputs():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    from isocon_amd import synth
    from make_golden_inputs import read_fasta
    fa = read_fasta(os.path.join(REF, "test", "data", "simulated_pacbio_reads_n_200.fa"))
    accs, seqs, _ = synth.make_reads(150, 500, 3, seed=81)
    return [("test_data_n200", fa), ("synth_150x500_3iso", dict(zip(accs, seqs)))]


def child(ci, ends=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "shims"))
    sys.path.insert(0, REF)
    import networkx
    if not hasattr(networkx.Graph, "node"):
        networkx.Graph.node = property(lambda g: g.nodes)
    with contextlib.redirect_stdout(io.StringIO()):
        from modules import isocon_get_candidates as R_IGC
    name, S = inputs()[ci]
    with tempfile.TemporaryDirectory() as tmp:
        read_file = os.path.join(tmp, "reads.fa")
        with open(read_file, "w") as fh:
            for acc, seq in S.items():
                fh.write(">%s\n%s\n" % (acc, seq))

        class Params(object):
            nr_cores = 1
            neighbor_search_depth = 2 ** 32
            verbose = False
            develop_logfile = None
            logfile = open(os.path.join(tmp, "log.txt"), "w")
            min_exon_diff = 20
            ignore_ends_len = ends
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
        steps = 1 + len(glob.glob(os.path.join(tmp, "candidates_step_*.fa")))
    rp = sorted([c, r, sha(t[0]), sha(t[1]), list(t[2])] for c in read_partition for r, t in read_partition[c].items())
    sys.stdout.write(json.dumps({"candidates": cands, "read_partition": rp, "to_realign": sorted(to_realign), "steps": steps}))


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), int(sys.argv[3]))
    kept, dropped, stored_inputs = [], [], {}
    for ci, (name, S) in enumerate(inputs()):
      for ends in (0, 15):
        outs = []
        for seed in range(3):
            env = dict(os.environ, PYTHONHASHSEED=str(seed))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(ci), str(ends)], env=env, capture_output=True, text=True, check=True)
            outs.append(r.stdout)
        agree = all(o == outs[0] for o in outs)
        e = json.loads(outs[0])
        if agree:
            kept.append({"name": "%s_ends%d" % (name, ends), "ignore_ends_len": ends, "input": name, "expect": e})
            stored_inputs[name] = [[a, s] for a, s in S.items()]
        else:
            dropped.append("%s_ends%d" % (name, ends))
        print(name, "ends", ends, "agree" if agree else "HASH-ORDER DEPENDENT", len(S), "reads ->", len(e["candidates"]), "candidates,", e["steps"], "steps,",
              len(e["read_partition"]), "assigned,", len(e["to_realign"]), "to realign")
    json.dump({"generator": "tests/golden/make_golden_candidates.py", "params": {"ignore_ends_len": "0 and 15 (per case)", "min_exon_diff": 20, "min_candidate_support": 2},
               "hash_order_dependent_cases_dropped": dropped, "cases": kept, "inputs": stored_inputs}, open(os.path.join(HERE, "g12_candidates.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
