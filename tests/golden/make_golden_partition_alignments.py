#!/usr/bin/env python3
"""Golden fixture tests/golden/g8_partition_alignments.json: the reference's own
modules/isocon_get_candidates.py::get_partition_alignments run on the output of its partition_strings, for the inputs
of make_golden_partitions.py, under PYTHONHASHSEED 0..3 (kept only if all agree).  edlib / parasail are absent:
tests/golden/shims forward to the CPU oracle (distances pinned by definition; alignment tie-breaks = policy 0,
"parity unpinned", see make_golden.py).  Gapped strings are stored as sha1 digests to keep the file small.

Usage:  python tests/golden/make_golden_partition_alignments.py          (build container only)
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
        exon_filtered = set()
        pa = R_IGC.get_partition_alignments(partition, M, G, exon_filtered, Params())
    uid = {}
    for seq in S.values():
        uid.setdefault(seq, len(uid))
    rows = sorted([uid[m], uid[s], int(t[0]), sha(t[1]), sha(t[2]), int(t[3])] for m in pa for s, t in pa[m].items())
    sys.stdout.write(json.dumps({"rows": rows, "exon_filtered": sorted(uid[s] for s in exon_filtered)}))


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
        (kept if agree else dropped).append({"name": name, "expect": json.loads(outs[0])} if agree else name)
        print(name, "agree" if agree else "HASH-ORDER DEPENDENT", len(json.loads(outs[0])["rows"]), "rows", len(json.loads(outs[0])["exon_filtered"]), "filtered")
    json.dump({"generator": "tests/golden/make_golden_partition_alignments.py", "inputs": "cases of g7_partitions.json (same names)",
               "hash_order_dependent_cases_dropped": dropped, "cases": kept}, open(os.path.join(HERE, "g8_partition_alignments.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
