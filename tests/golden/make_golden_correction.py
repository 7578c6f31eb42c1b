#!/usr/bin/env python3
"""Golden fixture tests/golden/g11_correction.json: the reference's own modules/correction_module.py::correct_strings
(with modules/functions.py's multi-alignment code) run on the partition alignments of the g7/g8 cases, i.e. on
partition_strings -> get_partition_alignments of the reference itself, under PYTHONHASHSEED 0..3 (kept if all agree).
edlib / parasail are absent: tests/golden/shims forward to the CPU oracle; the insertion threading in functions.min_ed
uses the stand-in's NW path rule ("parity unpinned", see tests/golden/shims/edlib.py).  Corrected sequences are stored
as digests, plus the multi-alignment width and column digest of every corrected partition.

Usage:  python tests/golden/make_golden_correction.py          (build container only)
"""
import contextlib
import hashlib
import io
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
from make_golden_partitions import cases  # noqa: E402


def sha(s):
    return hashlib.sha1(s.encode()).hexdigest()[:16]


def child(ci):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(HERE, "shims"))
    sys.path.insert(0, REF)
    import networkx
    if not hasattr(networkx.Graph, "node"):
        networkx.Graph.node = property(lambda g: g.nodes)
    with contextlib.redirect_stdout(io.StringIO()):
        from modules import partitions as R_PART
        from modules import isocon_get_candidates as R_IGC
        from modules import correction_module as R_COR
        from modules import functions as R_FUN

    class Params(object):
        nr_cores = 1
        neighbor_search_depth = 2 ** 32
        verbose = False
        develop_logfile = None
        min_exon_diff = 20
        ignore_ends_len = 15

    name, S = cases()[ci]
    with contextlib.redirect_stdout(io.StringIO()):
        G, partition, M, converged = R_PART.partition_strings(S, Params())
        pa = R_IGC.get_partition_alignments(partition, M, G, set(), Params())
        seq_to_acc = R_IGC.get_unique_seq_accessions(S)
        S_prime, _ = R_COR.correct_strings(pa, seq_to_acc, {}, 1, nr_cores=1, verbose=False)
        msa = []
        uid = {}
        for seq in S.values():
            uid.setdefault(seq, len(uid))
        for m in sorted(pa, key=lambda x: uid[x]):
            if len(pa[m]) > 1:
                am = R_FUN.create_multialignment_matrix(m, pa[m])
                msa.append([uid[m], len(am[m]), sha("".join("".join(am[s]) for s in sorted(am, key=lambda x: uid[x])))])
    sys.stdout.write(json.dumps({"S_prime": sorted([acc, sha(s), len(s)] for acc, s in S_prime.items()), "msa": msa}))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]))
    kept, dropped = [], []
    for ci, (name, S) in enumerate(cases()):
        outs = []
        for seed in range(4):
            env = dict(os.environ, PYTHONHASHSEED=str(seed))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(ci)], env=env, capture_output=True, text=True, check=True)
            outs.append(r.stdout)
        agree = all(o == outs[0] for o in outs)
        e = json.loads(outs[0])
        (kept if agree else dropped).append({"name": name, "expect": e} if agree else name)
        print(name, "agree" if agree else "HASH-ORDER DEPENDENT", len(e["S_prime"]), "corrected accessions,", len(e["msa"]), "multi-alignments")
    json.dump({"generator": "tests/golden/make_golden_correction.py", "inputs": "cases of g7_partitions.json (same names)",
               "hash_order_dependent_cases_dropped": dropped, "cases": kept}, open(os.path.join(HERE, "g11_correction.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
